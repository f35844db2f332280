cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3a; mkdir -p $o
timeout 2400 python -m pytest tests/test_hip_bench_shapes.py tests/test_hip_resample.py tests/test_hip_mixed.py tests/test_hip_adam.py -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
UNO_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $o/bench2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_c2 -- python tools/block_prof.py c2 > $o/stats_c2.log 2>&1
python tools/block_rocprof_summary.py $o/stats_c2 c2 $o/block_rocprof.json $o/c2_dispatches.csv > $o/summary_c2.txt 2>&1
rm -rf $o/stats_c2/*/*.db
timeout 900 python bench.py --no-cpu-baseline > $o/bench.log 2>&1
tail -3 $o/tests.log; tail -c 600 $o/bench2.log; tail -20 $o/summary_c2.txt
