cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LIB=${1:--}; tag=$(basename $LIB .so)
for w in fwd bwd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cold_${tag}_$w -- python tools/dev/vol3dcold.py $LIB $w 60 > gpurun_out/cold_${tag}_$w.log 2>&1
  grep "us per call" gpurun_out/cold_${tag}_$w.log
  f=$(ls gpurun_out/cold_${tag}_$w/*/*kernel_stats.csv | head -1); python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'uno::' in r['Name']: print(f"   {r['Name'].split('(')[0][-62:]:62s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:6.1f} us  min {float(r['MinNs'])/1e3:6.1f}")
PY
done
