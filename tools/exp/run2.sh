for n in 0 1 2 3 4 6 7; do python tools/exp/k3time.py tools/exp/lib/libk1exp$n.so 2>&1 | grep -v amdgpu.ids; done
