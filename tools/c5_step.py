"""C5 model step, float32 vs mixed precision (bench.py extras c5_model_f32 / c5_model_mixed) + per-kernel table of the mixed step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uno_amd import _native
from uno_amd.harness.mixed import c5_mixed_model_bench
dev = torch.device("cuda:0")
r = c5_mixed_model_bench(dev)
print("mixed:", {k: v for k, v in r.items() if k != "config"})
from uno_amd.harness import DarcyTrainer, MixedDarcyTrainer, UNO_9, synthetic_darcy_batch
for cls in (DarcyTrainer, MixedDarcyTrainer):
    torch.manual_seed(0)
    model = UNO_9(3, 64, pad=5).to(dev)
    tr = cls(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(4, 1024, 1234, dev)
    ms = bench._train_ms(lambda: tr.step(a, u), dev, steps=4, warmup=2)
    _native.profile_begin(100000)
    tr.step(a, u)
    torch.cuda.synchronize()
    agg = {}
    for name, t, by in _native.profile_end():
        e = agg.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += t
    print(cls.__name__, f"{ms:.1f} ms/step; library kernels: {sum(v[1] for v in agg.values()):.1f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {k:48s} {v[0]:4d} x  {v[1]:7.2f} ms")
    del tr, model
    torch.cuda.empty_cache()
