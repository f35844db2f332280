#!/bin/bash
# usage: tools/block_traffic.sh <profile_round output dir> <tag>  - profiles/block_traffic.json = {"c2": {kernel: bytes ...}, "c4": {...}}
# from the FETCH_SIZE / WRITE_SIZE passes (p4, p5) of the standalone blocks; the two blocks share kernel instantiations (the
# weight-gradient form of K2), so the table is kept per block
out=$1; tag=$2
: > profiles/${tag}_block_traffic.txt
for w in c2 c4; do
  rm -rf $out/pmc_blocks_$w; mkdir -p $out/pmc_blocks_$w
  cp -r $out/pmc_$w/p4 $out/pmc_blocks_$w/p4; cp -r $out/pmc_$w/p5 $out/pmc_blocks_$w/p5
  python tools/traffic_json.py $out/pmc_blocks_$w $out/traffic_$w.json $out/traffic_$w.txt > /dev/null
  echo "## standalone block $w (tools/block_prof.py $w 5 2 1)" >> profiles/${tag}_block_traffic.txt
  cat $out/traffic_$w.txt >> profiles/${tag}_block_traffic.txt
done
python -c "import json; json.dump({w: json.load(open('$out/traffic_%s.json' % w)) for w in ('c2', 'c4')}, open('profiles/block_traffic.json', 'w'), indent=1)"
