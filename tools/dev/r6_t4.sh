cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6t4; mkdir -p $o
timeout 1500 python -m pytest tests/test_hip_window.py tests/test_hip_channel_mix.py tests/test_hip_spectral2d.py -x -q > $o/t_a.log 2>&1; echo "a rc $?"; tail -4 $o/t_a.log
timeout 2400 python -m pytest tests/test_hip_headline_parity.py -x -q > $o/t_b.log 2>&1; echo "b rc $?"; tail -3 $o/t_b.log
python tools/dev/steplaunches.py > $o/step_launches.txt 2>&1; sed -n 27,34p $o/step_launches.txt | cut -c1-110; tail -1 $o/step_launches.txt
