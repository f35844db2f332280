import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spectral_oracle as so
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
nt = int(sys.argv[1]); B = int(sys.argv[2])
torch.set_num_threads(nt)
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5, block_cls=so.OracleOperatorBlock2d)
tr = DarcyTrainer(model)
a, u = synthetic_darcy_batch(B, 421, 1, "cpu")
t0 = time.perf_counter(); tr.step(a[:1], u[:1]); t1 = time.perf_counter()
tr.step(a, u); t2 = time.perf_counter()
print(f"threads={nt} B={B} warm(1 sample)={t1-t0:.1f}s step={t2-t1:.1f}s -> {B/(t2-t1):.3f} samples/s", flush=True)
