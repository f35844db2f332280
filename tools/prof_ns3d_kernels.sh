#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_ns3d_kernels.sh [width]  - rocprofv3 kernel statistics of 4 NS-3D training steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ns3d_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ns3d_prof -- python tools/prof_ns3d.py ${1:-32} > gpurun_out/ns3d_prof.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ns3d_prof/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step (ms):", tot / 4e6, " launches per step:", sum(int(r["Calls"]) for r in rows) / 4)
for r in rows[:30]:
    print(f'{r["Name"][:110]:110s} {int(r["Calls"])//4:>5d} {float(r["TotalDurationNs"])/4e6:8.2f} ms {float(r["AverageNs"])/1e3:8.1f} us')
PY
