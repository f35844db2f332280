cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3j; mkdir -p $o
timeout 2400 python -m pytest tests/test_hip_spectral2d.py tests/test_hip_random_shapes.py tests/test_hip_bench_shapes.py tests/test_hip_c5.py tests/test_hip_mixed.py -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
python tools/block_prof.py c2 > $o/block.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-extras > $o/bench.log 2>&1
tail -4 $o/tests.log; grep HIP-event $o/block.log; grep '^{' $o/bench.log | head -c 300
