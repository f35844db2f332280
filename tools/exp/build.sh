#!/bin/bash
# experiment builds of K3: tools/exp/build.sh 0 1 2 3 ... -> tools/exp/lib/libexp<N>.so  (product objects + an UNO_EXP=N copy of dft2d_inv_b)
cd /root/repo
mkdir -p tools/exp/lib
OBJS=$(ls uno_amd/lib/obj/*.o | grep -v dft2d_inv_b.o)
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DUNO_EXP=$n -Iuno_amd/csrc -c tools/exp/dft2d_inv_b.hip -o tools/exp/lib/inv_b_$n.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $OBJS tools/exp/lib/inv_b_$n.o -o tools/exp/lib/libexp$n.so ) &
done
wait; ls -la tools/exp/lib/*.so
