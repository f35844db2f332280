"""Every 1x1-layer call (K8) of one Darcy training step with its shape and flags, next to the kernel the library chose and its time."""
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
for _ in range(3): tr.step(a, u)
torch.cuda.synchronize()
log = []
orig2, orig1 = _native.channel_mix2, _native.channel_mix
def wrap(fn, two):
    def f(*args, **kw):
        x1 = args[0]; x2 = args[1] if two else None; w = args[2] if two else args[1]
        _native.profile_begin(8)
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        rec = _native.profile_end()
        flags = {k: (v is not None and v is not False) for k, v in kw.items() if k in ("transpose_w", "accumulate", "act_in", "dgelu_of", "y_act", "project", "out", "out2", "split_out", "dgelu_total")}
        log.append((tuple(x1.shape), None if x2 is None else tuple(x2.shape), tuple(w.shape), {k: v for k, v in flags.items() if v}, [(n.replace("uno::", ""), round(ms * 1e3, 1)) for n, ms, _ in rec]))
        return out
    return f
_native.channel_mix2, _native.channel_mix = wrap(orig2, True), wrap(orig1, False)
tr.step(a, u)
torch.cuda.synchronize()
for r in log: print(r)
