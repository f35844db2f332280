cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3k; mkdir -p $o
timeout 1200 python -m pytest tests/test_hip_instnorm.py tests/test_hip_blocks.py tests/test_hip_bf16_block.py -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
timeout 900 python bench.py --no-cpu-baseline --no-extras > $o/bench.log 2>&1
tail -4 $o/tests.log; grep '^{' $o/bench.log | head -c 300
