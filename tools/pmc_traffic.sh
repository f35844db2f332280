#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh <outdir>  - HBM traffic counters over a short bench run
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out; mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/p$i.log 2>&1
done
