cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6t1; mkdir -p $o
timeout 1500 python -m pytest tests/test_hip_window.py tests/test_hip_fused_upsample.py tests/test_hip_adam.py -x -q > $o/t_a.log 2>&1; echo "a rc $?"; tail -3 $o/t_a.log
timeout 2400 python -m pytest tests/test_hip_bench_shapes.py -x -q -k "c2_ or zz" > $o/t_b.log 2>&1; echo "b rc $?"; tail -5 $o/t_b.log
python tools/dev/steplaunches.py > $o/step_launches.txt 2>&1; tail -3 $o/step_launches.txt
