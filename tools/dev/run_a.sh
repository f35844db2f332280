cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/tb; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_resample.py tests/test_hip_blocks.py tests/test_hip_c5.py tests/test_hip_bf16_block.py -x -q -m gpu > $o/tests.log 2>&1; echo "rc $?" >> $o/tests.log
grep -E "passed|failed|Error|error|assert" $o/tests.log | head -20
python tools/dev/steptime.py - 2>&1 | tail -1
