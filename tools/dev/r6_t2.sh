cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6t2; mkdir -p $o
timeout 1500 python -m pytest tests/test_hip_window.py tests/test_hip_blocks.py tests/test_hip_spectral3d.py tests/test_harness_ns.py -x -q > $o/t_a.log 2>&1; echo "a rc $?"; tail -4 $o/t_a.log
timeout 2400 python -m pytest tests/test_hip_headline_parity.py -x -q > $o/t_b.log 2>&1; echo "b rc $?"; tail -4 $o/t_b.log
python tools/dev/steplaunches.py > $o/step_launches.txt 2>&1; tail -12 $o/step_launches.txt | cut -c1-110
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; python -c "
import json; d=json.load(open('$o/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['fwd_us'],v['frac']) for k,v in d['roofline']['operator_block'].items()})"
