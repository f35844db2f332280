cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6t6; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_resample.py tests/test_hip_blocks.py tests/test_hip_fused_upsample.py -x -q > $o/t_a.log 2>&1; echo "a rc $?"; tail -2 $o/t_a.log
python tools/dev/fusetime.py 20 2>&1 | grep -E "reverse|paired"
python tools/dev/steplaunches.py > $o/step_launches.txt 2>&1; grep -E "resample|dft2d_fwd|sum" $o/step_launches.txt | cut -c1-110
