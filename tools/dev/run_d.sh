cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3d; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/bench_stats -- python bench.py --no-cpu-baseline --no-extras > $o/bench_stats.log 2>&1
cp $(ls $o/bench_stats/*/*kernel_stats.csv | head -1) $o/bench_kernel_stats.csv
rm -rf $o/bench_stats
head -50 $o/bench_kernel_stats.csv | cut -c1-200
