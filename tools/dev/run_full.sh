cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3full; mkdir -p $o
timeout 1500 python -m pytest tests/ -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc $?" >> $o/smoke.log
tail -6 $o/tests.log; tail -2 $o/smoke.log
