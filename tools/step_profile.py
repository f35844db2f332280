"""Per-kernel time of ONE steady-state training step from a rocprofv3 kernel-trace sqlite db
(rocprofv3 --kernel-trace --stats -d DIR -- python bench.py --steps K --warmup W --no-cpu-baseline)."""
import glob, re, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + '/*/*.db'))[-1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5       # warmup + steps
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
# the optimiser's first launch delimits steps: count launches per step
per = len(idx) // (nsteps) if nsteps else 1
marks = idx[::per][:nsteps + 1]
t0, t1 = rows[marks[-2]][1], rows[marks[-1]][1]
print("step window ms", (t1 - t0) / 1e6)
agg = {}
for name, s, e in rows:
    if t0 <= s < t1:
        if name.startswith('void at::native'):
            m = re.search(r'at::native::(?:\(anonymous namespace\)::)?(\w+)', name)
            keys = [k for k in ['upsample', 'direct_copy', 'CUDAFunctor_add', 'GeluBackward', 'Gelu', 'MulFunctor', 'FillFunctor', 'reduce_kernel', 'sqrt', 'addcdiv', 'CatArray', 'multi_tensor', 'batch_norm', 'div', 'neg', 'pad', 'Pad'] if k in name]
            short = 'aten:' + (m.group(1) if m else '') + ':' + ','.join(keys)
        else:
            short = re.sub(r'\(.*', '', name)[:70]
        a = agg.setdefault(short, [0, 0]); a[0] += 1; a[1] += e - s
print("busy ms", sum(v[1] for v in agg.values()) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{v[1]/1e6:8.2f} ms {v[0]:4d}  {k}")
