cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3b; mkdir -p $o
timeout 2400 python -m pytest tests/test_hip_channel_mix.py tests/test_hip_bf16_block.py tests/test_hip_blocks.py tests/test_hip_c5.py tests/test_hip_mixed.py tests/test_harness_ns.py tests/test_hip_zz_dist.py "tests/test_hip_bench_shapes.py::test_c2_darcy_conv5_two_source_block_full_size" -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
UNO_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $o/bench2.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-extras > $o/bench.log 2>&1
tail -5 $o/tests.log; grep '^{' $o/bench2.log | tail -c 1500; grep '^{' $o/bench.log | head -c 400
