cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6fuse; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_fused_upsample.py -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -3 $o/tests.log
for s in 0 1 3 4 24 27; do UNO_K3A_STAGGER=$s python tools/dev/k3a_time.py 2>&1 | grep "stagger"; done
timeout 600 python tools/dev/fusetime.py 10 > $o/fusetime.txt 2>&1; cat $o/fusetime.txt | grep -v amdgpu.ids
