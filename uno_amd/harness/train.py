"""Training step of the Darcy workload (reference train_darcy.py:47-56) with data parallelism over
one process per GPU.

Every sample is independent through the network (batch is a free index of the mode contraction,
InstanceNorm is per sample, the loss is a per-sample sum), so the minibatch is sharded across ranks
and the only exchange is ONE gradient all-reduce per step (RCCL over xGMI on MI355X, gloo in the CPU
tests).  All gradients live in a single flat float32 buffer (complex grads as interleaved re/im), so
that exchange is one large collective on a few hundred MB instead of a per-parameter stream:
xGMI all-reduce is per-link bandwidth bound, large messages are what it wants.  The reference loss
is a SUM over the batch, hence gradients are SUMMED over ranks: the update equals the single-process
update on the concatenated global batch."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .losses import lp_loss_rel_sum
from .optim import ComplexAdam


def synthetic_darcy_batch(batch, S, seed, device, dtype=torch.float32):
    """Synthetic Darcy pair of the benchmark shape: coefficient field a ~ U[0,1) (B,S,S,1) and target
    u ~ U[0,1) (B,S,S); generated on the host from a seeded generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.rand(batch, S, S, 1, generator=g, dtype=dtype)
    u = torch.rand(batch, S, S, generator=g, dtype=dtype)
    return a.to(device), u.to(device)


class FlatGradients:
    """Backs every parameter's .grad with a view into one flat float32 buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        sizes = [p.numel() * (2 if p.is_complex() else 1) for p in self.params]
        dev = self.params[0].device
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(self.params, sizes):
            seg = self.flat[off:off + n]
            if p.is_complex():
                p.grad = torch.view_as_complex(seg.view(*p.shape, 2))
            else:
                assert p.dtype == torch.float32
                p.grad = seg.view(p.shape)
            off += n

    def zero_(self):
        self.flat.zero_()

    def all_reduce_sum(self, group=None, force=False):
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)


def ns2d_rollout_loss(model, xx, yy, T_f, step=1):
    """Autoregressive loss of the NS-2D training step (reference ns_train_2d.py:46-62): the model predicts
    `step` frames from the last T_in, the prediction is appended to the input window, the per-step relative
    L2 losses are summed; ONE backward runs through the whole unrolled chain."""
    loss = 0
    B = yy.shape[0]
    for t in range(0, T_f, step):
        im = model(xx)
        loss = loss + lp_loss_rel_sum(im.reshape(B, -1), yy[..., t:t + step].reshape(B, -1))
        xx = torch.cat((xx[..., step:], im), dim=-1)
    return loss


def ns3d_loss(model, x, y):
    """Space-time loss of the NS-3D training step (reference ns_train_3d.py:53,64): one forward, global relative L2."""
    B, S, T_f = x.shape[0], x.shape[1], y.shape[-1]
    out = model(x).view(B, S, S, T_f)
    return lp_loss_rel_sum(out.reshape(B, -1), y.reshape(B, -1))


class DarcyTrainer:
    """model + ComplexAdam + flat-gradient data parallelism.  step(a, u) runs forward, relative-L2 loss,
    backward, gradient all-reduce and the optimiser update; it returns the (device) loss tensor and
    never synchronises with the host."""

    def __init__(self, model, lr=1e-3, weight_decay=1e-3, group=None, force_collectives=False):
        self.model = model
        self.group = group
        self.force_collectives = force_collectives      # tests: run the collectives even in a 1-rank group
        self.grads = FlatGradients(model.parameters())
        self.opt = ComplexAdam(model.parameters(), lr=lr, weight_decay=weight_decay)
        self.broadcast_parameters()

    def broadcast_parameters(self):
        if dist.is_available() and dist.is_initialized() and (self.force_collectives or dist.get_world_size(self.group) > 1):
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(torch.view_as_real(t.data) if t.is_complex() else t.data, src=0, group=self.group)

    def step_with(self, loss_closure):
        """zero grads -> loss_closure() -> backward -> gradient all-reduce -> optimiser update."""
        self.grads.zero_()
        loss = loss_closure()
        loss.backward()
        self.grads.all_reduce_sum(self.group, self.force_collectives)
        self.opt.step()
        return loss.detach()

    def step(self, a, u):
        B, S = a.shape[0], a.shape[1]
        return self.step_with(lambda: lp_loss_rel_sum(self.model(a).reshape(B, -1), u.reshape(B, -1)))
