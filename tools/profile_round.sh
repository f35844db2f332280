#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <round tag, e.g. r02>  - every profile artefact profiles/ holds for a round:
#   <tag>_block2d_kernel_stats.csv / <tag>_block3d_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the standalone blocks
#   <tag>_block2d_pmc.txt / <tag>_block3d_pmc.txt                      SQ / TA counters of the block kernels (separate --pmc passes)
#   block_traffic.json + <tag>_block_traffic.txt                       FETCH_SIZE / WRITE_SIZE passes of the standalone blocks
#   <tag>_bench_kernel_stats.csv, hbm_traffic.json                     the same for bench.py's training step
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out profiles
for w in c2 c4; do
  n=$([ $w = c2 ] && echo 2d || echo 3d)
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -- python tools/block_prof.py $w 20 > $out/stats_$w.log 2>&1
  cp $(ls $out/stats_$w/*/*kernel_stats.csv | head -1) profiles/${tag}_block${n}_kernel_stats.csv
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
             "GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/pmc_$w/p$i -- python tools/block_prof.py $w 5 > $out/pmc_$w.p$i.log 2>&1
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
