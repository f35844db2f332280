"""f32 MFMA floor of the Darcy training step: every K8 / K9 call of one step with its shape, flops and bytes.
python tools/dev/flopcount.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(16, 421, 1234, dev)
for _ in range(2): tr.step(a, u)
log = []
cm, cw = _native.channel_mix, _native.channel_wgrad
def cm_log(x, w, *args, **kw):
    B, Ci, P = x.shape[0], x.shape[1], x.shape[2:].numel()
    Co = w.shape[1] if kw.get("transpose_w") else w.shape[0]
    Ci_eff = w.shape[0] if kw.get("transpose_w") else w.shape[1]
    log.append(("K8", B, Ci_eff, Co, P))
    return cm(x, w, *args, **kw)
def cw_log(gy, x, *args, **kw):
    B, Co, P = gy.shape[0], gy.shape[1], gy.shape[2:].numel()
    log.append(("K9", B, x.shape[1], Co, P))
    return cw(gy, x, *args, **kw)
_native.channel_mix, _native.channel_wgrad = cm_log, cw_log
import uno_amd.integral_operators as io
tr.step(a, u)
torch.cuda.synchronize()
PEAK, HBM = 157e12, 5.0e12
tot_f = tot_b = tot_max = 0.0
for kind, B, Ci, Co, P in log:
    fl = 2.0 * B * Ci * Co * P
    by = 4.0 * B * P * (Ci + Co)
    tf, tb = fl / PEAK * 1e6, by / HBM * 1e6
    tot_f += tf; tot_b += tb; tot_max += max(tf, tb)
    print(f"{kind} B={B} {Ci:4d}->{Co:4d} P={P:7d}: {fl/1e9:7.1f} GFLOP = {tf:6.0f} us at 157 TF/s | {by/1e6:7.0f} MB = {tb:6.0f} us at 5 TB/s")
print(f"calls {len(log)}; sum of MFMA times {tot_f/1e3:.2f} ms, of byte times {tot_b/1e3:.2f} ms, of max(MFMA, bytes) {tot_max/1e3:.2f} ms")
