cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6t3; mkdir -p $o
timeout 1500 python -m pytest tests/test_hip_spectral2d.py tests/test_hip_spectral3d.py tests/test_hip_blocks.py tests/test_hip_random_shapes.py tests/test_harness_ns.py -x -q > $o/t_a.log 2>&1; echo "a rc $?"; tail -4 $o/t_a.log
python tools/dev/steplaunches.py > $o/step_launches.txt 2>&1; grep -E "mode_gemm|sum" $o/step_launches.txt | cut -c1-110
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; python -c "
import json; d=json.load(open('$o/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['backward']['frac']); b=d['spectral_block_3d']; print('3d', b['fwd_us'], b['fwd_frac_of_8TBs'], b['bwd_us'], b['bwd_frac_of_8TBs'], {k:v['avg_us'] for k,v in b['bwd_kernels'].items()})"
