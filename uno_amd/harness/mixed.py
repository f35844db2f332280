"""Mixed-precision training step of the Darcy model (BASELINE.json configs[4]: bf16 activations + half-precision spectral
weights, f32 accumulation).

Recipe (the usual master-weight scheme): parameters, gradients and Adam state stay float32 / complex64; every activation
tensor between kernels - and every activation gradient - is bfloat16; each spectral layer reads its complex weights through a
float16 (re, im) copy made per call; all kernels accumulate in float32 (exact f32 MFMA / FMA on the widened values) and round
once on the way out.  The reference has no such mode (integral_operators.py:187 raises on bfloat16); parity is defined
against the float32 reference on pre-rounded inputs, tests/test_hip_c5.py / tests/test_hip_mixed.py."""
from __future__ import annotations

import time

import torch

from ..integral_operators import enable_mixed_precision
from .losses import lp_loss_rel_sum
from .models import UNO_9
from .train import DarcyTrainer, synthetic_darcy_batch


class MixedDarcyTrainer(DarcyTrainer):
    """DarcyTrainer whose forward / backward run on bfloat16 activations (same flat-gradient data parallelism, same Adam)."""

    def __init__(self, model, **kw):
        enable_mixed_precision(model)
        super().__init__(model, **kw)

    def step(self, a, u):
        B = a.shape[0]
        ab = a.to(torch.bfloat16)
        return self.step_with(lambda: lp_loss_rel_sum(self.model(ab).reshape(B, -1).float(), u.reshape(B, -1)))


def c5_mixed_model_bench(dev, B: int = 4, S: int = 1024, steps: int = 4, warmup: int = 2):
    """UNO_9(3, 64, pad=5) at S x S, batch B: ms / step of the mixed-precision step (bench.py extra key)."""
    torch.manual_seed(0)
    torch.cuda.synchronize(dev)                      # (also initialises the device context when this is the first CUDA call)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)          # the figure below is this run's peak, not an earlier workload's
    model = UNO_9(3, 64, pad=5).to(dev)
    tr = MixedDarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(B, S, 1234, dev)
    for _ in range(warmup):
        tr.step(a, u)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(a, u)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    lv = float(loss)
    assert lv == lv, "mixed-precision training produced NaN"
    return {"config": f"C5 model: UNO_9(3,64,pad=5) at {S}^2, batch {B}, bf16 activations + fp16 spectral weights, f32 accumulation "
                      "and f32 master weights / Adam", "ms_per_step": ms, "samples_per_s": B / ms * 1e3,
            "peak_mem_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30, "final_loss": lv}
