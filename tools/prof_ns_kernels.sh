#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_ns_kernels.sh 2d | 3d [width]  - rocprofv3 kernel statistics of 4 NS-2D / NS-3D training steps
kind=${1:-2d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ns${kind}_prof
rm -rf $out
if [ "$kind" = "2d" ]; then target="python tools/prof_ns2d.py"; else target="python tools/prof_ns3d.py ${2:-32}"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- $target > $out.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
print("total kernel ms / step:", sum(float(r["TotalDurationNs"]) for r in rows) / 4e6)
for r in rows[:32]:
    print(f'{r["Name"][:110]:110s} {int(r["Calls"])//4:>6d} {float(r["TotalDurationNs"])/4e6:8.2f} ms {float(r["AverageNs"])/1e3:8.1f} us')
PY
