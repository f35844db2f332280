cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/tb; mkdir -p $o
timeout 900 python -m pytest tests/test_harness_ns.py -x -q -m gpu > $o/tests.log 2>&1; echo "rc $?" >> $o/tests.log
grep -E "passed|failed|Error|error|assert" $o/tests.log | head -20
for i in 1 2; do
timeout 300 python tools/bench_ns.py --graph --c3 2>&1 | grep "C[34]"
done
