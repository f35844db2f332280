cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3n; mkdir -p $o
timeout 2400 python -m pytest tests/test_hip_channel_mix.py tests/test_hip_blocks.py tests/test_hip_bf16_block.py tests/test_hip_random_shapes.py tests/test_harness_ns.py -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
python tools/dev/steplaunches.py > $o/launches.txt 2>&1
tail -3 $o/tests.log; grep -E "^ 7[0-9]|sum" $o/launches.txt
