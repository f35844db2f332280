"""Build libuno_spectral.so for gfx950 with hipcc (no GPU needed: hipcc cross-compiles).

    python -m uno_amd.build [--force] [--verbose]

The library is built IN-TREE (uno_amd/lib/libuno_spectral.so) so it travels with the
repository snapshot to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libuno_spectral.so")
STAMP = os.path.join(LIBDIR, "libuno_spectral.stamp")
SOURCES = ["capi.hip", "dft2d_fwd.hip", "dft2d_fwd_r4.hip", "dft2d_inv.hip", "dft2d_inv_b.hip", "dft2d_inv_c.hip", "dft2d_inv_add.hip", "dft2d_plane.hip", "dft2d_b16.hip", "mode_gemm.hip", "cdft_axis.hip", "dft3d_volume.hip", "dft_generic.hip", "resample2d.hip", "channel_mix.hip", "adam.hip", "pointwise_fused.hip", "instnorm.hip", "lift_bwd.hip"]
HEADERS = ["uno_common.h", "dft2d_fwd_kernel.h", "dft2d_fwd_ft_kernel.h", "dft2d_fwd_ht_kernel.h", "dft2d_inv_kernel.h", "dft2d_inv_add_kernel.h", os.path.join("..", "..", "include", "uno_spectral.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# per-source flags.  mode_gemm.hip: its 4x4x1 kernel keeps 16 accumulator tiles live across a 4-step unrolled loop; with the
# default AGPR form the register allocator permutes the tiles at every back edge (268 v_accvgpr moves per iteration)
# dft2d_inv_add.hip: its kernels park MFMA results (stage 1' of the addend) across a tile to use them as MFMA A operands; in the AGPR form
# the compiler copies all of them into VGPRs at the head of every tile (112 v_accvgpr_read + twice the registers: scratch reloads on a
# kernel with one wave per SIMD, 554 us against 187 for K3 alone)
EXTRA_FLAGS = {"mode_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "dft2d_inv_add.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _digest() -> str:
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == digest:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(STAMP, "w") as fh:
        fh.write(digest)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose))
