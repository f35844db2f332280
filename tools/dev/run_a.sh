cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/tb; mkdir -p $o
timeout 1200 python -m pytest tests/test_harness_ns.py tests/test_hip_blocks.py -x -q -m gpu > $o/tests.log 2>&1; echo "rc $?" >> $o/tests.log
grep -E "passed|failed|Error|error|assert" $o/tests.log | head -20
python - <<'PY'
import bench, torch, json
dev = torch.device("cuda:0")
PY
