"""Build a variant of the library with extra -D flags on ONE source file (the other objects come from the normal build):
    python tools/dev/mkvariant.py <tag> <source.hip> -DUNO_X=1 ...   ->  uno_amd/lib/variants/libuno_<tag>.so
The dev timers (tools/dev/*.py <lib.so>) take the path; the product never loads a variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uno_amd import build as B
tag, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
vdir = os.path.join(B.LIBDIR, "variants")
os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, f"{os.path.splitext(src)[0]}_{tag}.o")
subprocess.run([B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *defs, "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
objs = [obj if s == src else os.path.join(B.LIBDIR, "obj", os.path.splitext(s)[0] + ".o") for s in B.SOURCES]
out = os.path.join(vdir, f"libuno_{tag}.so")
subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-fno-gpu-rdc", *objs, "-o", out], check=True)
print(out)
