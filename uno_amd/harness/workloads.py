"""The three data-parallel training workloads of BASELINE.json as (trainer, step, shard) bundles - what `bench.py --workload`
times and what the world-size-2 tests drive:

    c2   Darcy 2-D 421^2, UNO_9(3, 64, pad=5), batch 16 per GPU                         (train_darcy.py:47-56)
    c4   Navier-Stokes 3-D 64 x 64 x 20, Uno3D_T20(6, 32, pad=3), batch 8 per GPU        (ns_train_3d.py:48-70, navier_stokes_uno3d.py:301-384)
    c5   Darcy 2-D 1024^2, UNO_9(3, 64, pad=5), bf16 activations + fp16 spectral weights, batch 4 per GPU

All three shard the minibatch over the ranks (one process per GPU), keep a full model replica per rank and exchange gradients
with the bucketed SUM all-reduce of FlatGradients (the reference losses are sums over samples).  `small=True` gives a shape of
the same structure that a CPU / a shared test GPU steps through in seconds."""
from __future__ import annotations

import torch

from .losses import lp_loss_rel_sum
from .mixed import MixedDarcyTrainer
from .models import UNO_9, Uno3D_T20
from .train import DarcyTrainer, ns3d_loss, synthetic_darcy_batch

WORKLOADS = ("c2", "c4", "c5")


class Workload:
    """trainer: DarcyTrainer (flat gradients + ComplexAdam); step(lo, hi) runs one training step on samples [lo, hi) of the batch
    this object holds and returns the device loss; `batch` = samples held; `describe` = the config string of the bench line."""

    def __init__(self, name, trainer, step, batch, describe, metric, dtype):
        self.name, self.trainer, self._step, self.batch, self.describe, self.metric, self.dtype = name, trainer, step, batch, describe, metric, dtype

    def step(self, lo=0, hi=None):
        return self._step(lo, self.batch if hi is None else hi)


def build(name: str, device, batch=None, seed: int = 1234, small: bool = False, model_seed: int = 0, block_cls=None, **trainer_kw) -> Workload:
    """`batch` samples of synthetic data on `device` (seeded: the same call gives the same tensors on every rank), the model
    initialised from `model_seed` (rank 0's parameters are broadcast by the trainer), trainer keywords passed through
    (lr, weight_decay, bucket_mb, group, force_collectives).  block_cls: operator-block class of the model (tests: the oracle's)."""
    if name not in WORKLOADS:
        raise ValueError(f"unknown workload {name!r} (choose from {WORKLOADS})")
    kw = dict(lr=1e-3, weight_decay=1e-3)
    kw.update(trainer_kw)
    mk = {} if block_cls is None else {"block_cls": block_cls}
    torch.manual_seed(model_seed)
    if name == "c4":
        S, width = (32, 4) if small else (64, 32)
        B = batch or (2 if small else 8)
        model = Uno3D_T20(6, width, pad=3, **mk).to(device)
        trainer = DarcyTrainer(model, **kw)
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = torch.randn(B, S, S, 10, 1, generator=g).to(device)
        y = torch.randn(B, S, S, 20, generator=g).to(device)
        step = lambda lo, hi: trainer.step_with(lambda: ns3d_loss(model, x[lo:hi], y[lo:hi]))
        return Workload(name, trainer, step, B, f"Navier-Stokes 3D {S}x{S}x20 (10 -> 20 steps), Uno3D_T20(6,{width},pad=3), train step "
                        "(fwd+loss+bwd+allreduce+Adam)", "UNO training samples/s (NS-3D 64x64x20)", "f32")
    S, width = (72, 8) if small else ((421, 64) if name == "c2" else (1024, 64))
    B = batch or (2 if small else (16 if name == "c2" else 4))
    model = UNO_9(3, width, pad=5, **mk).to(device)
    trainer = (MixedDarcyTrainer if name == "c5" else DarcyTrainer)(model, **kw)
    a, u = synthetic_darcy_batch(B, S, seed, device)
    step = lambda lo, hi: trainer.step(a[lo:hi], u[lo:hi])
    if name == "c5":
        return Workload(name, trainer, step, B, f"Darcy 2D {S}x{S}, UNO_9(3,{width},pad=5), bf16 activations + fp16 spectral weights "
                        "(f32 accumulation, f32 master weights / Adam), train step (fwd+loss+bwd+allreduce+Adam)",
                        "UNO training samples/s (1024^2 Darcy, mixed precision)", "bf16")
    return Workload(name, trainer, step, B, f"Darcy 2D {S}x{S}, UNO_9(3,{width},pad=5) 64ch, train step (fwd+loss+bwd+allreduce+Adam)",
                    "UNO training samples/s (421^2 Darcy)", "f32")


def flat_params(model):
    """every parameter as one float32 vector (complex ones as interleaved re / im) - what the data-parallel checks compare"""
    return torch.cat([(torch.view_as_real(p.detach()) if p.is_complex() else p.detach()).reshape(-1).float().cpu() for p in model.parameters()])
