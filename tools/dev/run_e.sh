cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3e; mkdir -p $o
timeout 2400 python -m pytest tests/test_hip_blocks.py tests/test_hip_channel_mix.py tests/test_hip_spectral2d.py tests/test_hip_spectral3d.py tests/test_hip_c5.py tests/test_hip_mixed.py tests/test_hip_bf16_block.py tests/test_hip_zz_dist.py tests/test_harness_ns.py tests/test_hip_random_shapes.py -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
timeout 900 python bench.py --no-cpu-baseline > $o/bench.log 2>&1
tail -5 $o/tests.log; grep '^{' $o/bench.log | head -c 300
