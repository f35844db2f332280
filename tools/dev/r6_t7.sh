cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6t7; mkdir -p $o
python tools/dev/fusetime.py 20 2>&1 | grep -E "sweep|alternating|reverse sweep"
timeout 2400 python -m pytest tests/test_hip_channel_mix.py tests/test_hip_window.py tests/test_hip_blocks.py tests/test_hip_spectral2d.py tests/test_hip_resample.py tests/test_hip_fused_upsample.py tests/test_hip_adam.py tests/test_harness_ns.py tests/test_hip_redzone.py -x -q > $o/t_a.log 2>&1; echo "a rc $?"; tail -3 $o/t_a.log
python tools/dev/steplaunches.py > $o/step_launches.txt 2>&1; tail -1 $o/step_launches.txt
