cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/full; mkdir -p $o
timeout 3000 python -m pytest tests/ -x -q -m gpu > $o/tests.log 2>&1
echo "tests rc $?" >> $o/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc $?" >> $o/smoke.log
tail -6 $o/tests.log; tail -2 $o/smoke.log
