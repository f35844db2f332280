#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <round tag, e.g. r02>  - every profile artefact profiles/ holds for a round:
#   <tag>_block2d_kernel_stats.csv / <tag>_block3d_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the standalone blocks
#   <tag>_block2d_dispatches.csv / _block3d_ + block_rocprof.json      per-dispatch rows of the timed segments, span / kernel-sum per call
#   <tag>_block2d_pmc.txt / <tag>_block3d_pmc.txt                      SQ / TA counters of the block kernels (separate --pmc passes)
#   block_traffic.json + <tag>_block_traffic.txt                       FETCH_SIZE / WRITE_SIZE passes of the standalone blocks
#   <tag>_bench_kernel_stats.csv, hbm_traffic.json                     the same for bench.py's training step
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out profiles
for w in c2 c4; do
  n=$([ $w = c2 ] && echo 2d || echo 3d)
  # same warm protocol as bench.py's _timed (3 untimed groups, 5 x 20 timed calls each way); clocks sampled alongside
  ( while true; do rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk' | tr '\n' ' '; echo; sleep 0.5; done ) > $out/clocks_$w.txt & smi=$!
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -- python tools/block_prof.py $w 20 5 3 > $out/stats_$w.log 2>&1
  kill $smi 2>/dev/null
  cp $(ls $out/stats_$w/*/*kernel_stats.csv | head -1) profiles/${tag}_block${n}_kernel_stats.csv
  python tools/block_rocprof_summary.py $out/stats_$w $w profiles/block_rocprof.json profiles/${tag}_block${n}_dispatches.csv 100 > $out/rocprof_summary_$w.txt 2>&1
  { grep 'HIP-event' $out/stats_$w.log; sort $out/clocks_$w.txt | uniq -c | sort -rn | head -4; } > profiles/${tag}_block${n}_live_and_clocks.txt
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
             "GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/pmc_$w/p$i -- python tools/block_prof.py $w 5 2 1 > $out/pmc_$w.p$i.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_$w > profiles/${tag}_block${n}_pmc.txt
done
bash tools/block_traffic.sh $out $tag
# the training step
bash tools/pmc_traffic.sh $out/pmc_bench
python tools/traffic_json.py $out/pmc_bench profiles/hbm_traffic.json profiles/${tag}_bench_pmc_traffic.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python bench.py --no-cpu-baseline --no-extras > $out/bench_stats.log 2>&1
cp $(ls $out/bench_stats/*/*kernel_stats.csv | head -1) profiles/${tag}_bench_kernel_stats.csv
ls -la profiles/
