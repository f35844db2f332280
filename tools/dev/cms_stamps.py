"""Phase cycle stamps of K8-S (UNO_CMS_EXP=64): python tools/dev/cms_stamps.py Ci Co P [B] [t]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["UNO_CMS_EXP"] = str(64 | int(os.environ.get("UNO_CMS_EXP", "0")))
import torch
from uno_amd import _native
if os.environ.get('UNO_LIB'): _native.LIB_PATH = os.path.abspath(os.environ['UNO_LIB'])
dev = torch.device("cuda:0")
Ci, Co, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
tr = len(sys.argv) > 5 and sys.argv[5] == "t"
x = torch.randn(B, Ci, P, device=dev); w = torch.randn(Ci, Co, device=dev) if tr else torch.randn(Co, Ci, device=dev)
npt = (P + 127) // 128; ntile = npt * (Co // 128); per_xcd = (ntile + 7) // 8
nwg = 8 * per_xcd * B
st = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
os.environ["UNO_CMS_STAMPS"] = hex(st.data_ptr())
for _ in range(3):
    _native.channel_mix(x, w, None, transpose_w=tr)
torch.cuda.synchronize()
s = st.view(nwg, 4, 8).cpu().double()
ok = s[:, 0, 5] > 0
s = s[ok]
nch = Ci // 32
names = ["store (wait loads + split + LDS writes)", "barrier 1", "compute", "barrier 2", "epilogue", "main loop"]
print(f"{Ci}->{Co} P={P} B={B} tr={tr}: {int(ok.sum())} workgroups, {nch} chunks; mean cycles per wave (per chunk for the first four)")
for i, n in enumerate(names):
    v = s[:, :, i].mean().item()
    print(f"  {n:42s} {v:10.0f}" + (f"   {v / nch:8.0f} per chunk" if i < 4 else ""))
t0 = s[:, 0, 6]; rt = s[:, 0, 7]
life = (s[:, 0, 5] + s[:, 0, 4])
print(f"  workgroup lifetime mean {life.mean().item():.0f} cycles; kernel span {(rt.max() - rt.min()).item() / 100:.1f} us (100 MHz clock, first start to last end-of-loop)")
