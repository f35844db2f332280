#!/bin/bash
# build ablation variants of the library: tools/ablate.sh 1 2 4 7 ...  -> uno_amd/lib/libuno_ablate<N>.so
cd /root/repo/uno_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -shared -DUNO_ABLATE=$n capi.hip dft2d_fwd.hip dft2d_fwd_r4.hip dft2d_inv.hip dft2d_plane.hip mode_gemm.hip cdft_axis.hip resample2d.hip channel_mix.hip adam.hip pointwise_fused.hip instnorm.hip -o ../lib/libuno_ablate$n.so &
done
wait; ls -la ../lib/*.so
