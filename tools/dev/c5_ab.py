"""C5 model step, f32 and mixed precision (UNO_9(3,64,pad=5) at 1024^2, batch 4): ms per step.  A/B by environment (UNO_CW_SPLIT_OFF=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
from uno_amd.harness.mixed import c5_mixed_model_bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(4, 1024, 1234, dev)
for _ in range(2): tr.step(a, u)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): loss = tr.step(a, u)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"C5 f32: {ms:.2f} ms/step loss {float(loss):.4f}", flush=True)
del tr, model; torch.cuda.empty_cache()
r = c5_mixed_model_bench(dev)
print(f"C5 mixed: {r['ms_per_step']:.2f} ms/step loss {r['final_loss']:.4f}  ratio f32/mixed {ms / r['ms_per_step']:.3f}", flush=True)
