cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6fuse; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_fused_upsample.py -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -15 $o/tests.log
timeout 600 python tools/dev/fusetime.py 20 > $o/fusetime.txt 2>&1; cat $o/fusetime.txt | grep -v amdgpu.ids
