for n in 1 2 3 6 7; do python tools/exp/blocktime.py tools/exp/lib/libk1exp$n.so 2>&1 | grep "fwd_ht\|^tools" | head -2; done
