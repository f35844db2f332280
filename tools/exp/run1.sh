for n in 0 1 2 3 4 8 11 16; do python tools/exp/k3time.py tools/exp/lib/libexp$n.so 2>&1 | grep -v amdgpu.ids; done
EXP_NW=3 EXP_G=2 python tools/exp/k3time.py tools/exp/lib/libexp0.so 2>&1 | grep -v amdgpu.ids
EXP_NW=3 EXP_G=1 python tools/exp/k3time.py tools/exp/lib/libexp0.so 2>&1 | grep -v amdgpu.ids
EXP_NW=3 EXP_G=1 EXP_NOTAB=1 python tools/exp/k3time.py tools/exp/lib/libexp0.so 2>&1 | grep -v amdgpu.ids
EXP_NW=4 EXP_G=3 python tools/exp/k3time.py tools/exp/lib/libexp0.so 2>&1 | grep -v amdgpu.ids
EXP_NW=2 EXP_G=4 python tools/exp/k3time.py tools/exp/lib/libexp0.so 2>&1 | grep -v amdgpu.ids
EXP_NW=3 EXP_G=2 python tools/exp/k3time.py tools/exp/lib/libexp2.so 2>&1 | grep -v amdgpu.ids
python tools/exp/k3time.py tools/exp/lib/libexp0.so 432 2>&1 | grep -v amdgpu.ids
python tools/exp/k3time.py tools/exp/lib/libexp0.so 448 2>&1 | grep -v amdgpu.ids
