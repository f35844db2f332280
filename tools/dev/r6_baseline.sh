cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6base; mkdir -p $o
python bench.py --no-extras --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc $?"
python tools/dev/steplaunches.py > $o/step_launches.txt 2>&1
python tools/cpu_thread_sweep.py r06 > $o/sweep.log 2>&1; cp profiles/r06_cpu_threads.txt $o/
tail -c 1500 $o/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6base/bench.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r.get('frac_of_copy'), r.get('copy_ceiling'))
print(json.dumps({k:{kk:vv for kk,vv in v.items() if kk!='kernels'} for k,v in r.get('operator_block',{}).items()}, indent=1) if 'error' not in r.get('operator_block',{}) else r['operator_block'])
print(d['cpu_baseline'])
PY
tail -3 $o/step_launches.txt; cat $o/r06_cpu_threads.txt
