#!/bin/bash
# experiment builds of K1-FT: tools/exp/build_k1.sh 0 1 2 ... -> tools/exp/lib/libk1exp<N>.so
cd /root/repo
mkdir -p tools/exp/lib
OBJS=$(ls uno_amd/lib/obj/*.o | grep -v dft2d_fwd_r4.o)
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DUNO_EXP=$n -Iuno_amd/csrc -c tools/exp/dft2d_fwd_r4.hip -o tools/exp/lib/fwd_r4_$n.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $OBJS tools/exp/lib/fwd_r4_$n.o -o tools/exp/lib/libk1exp$n.so ) &
done
wait; ls tools/exp/lib/libk1exp*.so
