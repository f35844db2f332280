cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pb; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_project_backward.py -x -q > $o/t1.log 2>&1; tail -4 $o/t1.log
python tools/dev/pbdebug.py - 2>&1 | grep -c "bad rows \[\]"
timeout 1500 python -m pytest tests/test_hip_window.py tests/test_hip_channel_mix.py tests/test_hip_headline_parity.py tests/test_hip_blocks.py tests/test_hip_zz_dist.py tests/test_hip_bench_shapes.py -x -q > $o/t2.log 2>&1; tail -3 $o/t2.log
python tools/dev/steplaunches.py - > $o/step_launches.txt 2>&1; sed -n 27,32p $o/step_launches.txt; tail -1 $o/step_launches.txt
python tools/dev/fusetime.py 20 2>&1 | grep -E "project_backward|fc1" > $o/ab.txt; cat $o/ab.txt
