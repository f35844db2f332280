cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6fuse; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_fused_upsample.py -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -2 $o/tests.log
python tools/dev/k3a_time.py 2>&1 | grep stagger
timeout 600 python tools/dev/fusetime.py 10 2>&1 | grep -E "round|fused|adjoint=True"
