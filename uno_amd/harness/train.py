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

import time

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
    """Backs every parameter's .grad with a view into one flat float32 buffer, and sums that buffer across the
    data-parallel group in a few large buckets while the backward pass is still running.

    Buckets are contiguous slices of the flat buffer (~`bucket_mb` each), numbered in the order the backward pass
    completes them (last-registered parameters first).  A post-accumulate hook on every parameter counts its
    bucket down; a complete bucket is all-reduced asynchronously (RCCL runs it on its own stream, overlapping the
    rest of the backward), always in bucket order so that every rank issues the same sequence of collectives.
    finish() issues whatever is left (parameters that took no part in this backward) and waits."""

    def __init__(self, params, bucket_mb: float = 32.0, comm_dtype=None):
        """comm_dtype: None - buckets travel as float32 (the default: bit-equal to a single process on the global batch up to summation
        order); torch.bfloat16 - a bucket is rounded to bfloat16 for the exchange and widened into the float32 buffer afterwards: half
        the bytes on the links (the 3.8 GB gradient of Uno3D_T20 at width 32 is link-bound at 8 ranks), gradient
        entries good to ~2^-8 relative.  The optimiser state and the update stay float32."""
        self.comm_dtype = comm_dtype
        self.params = [p for p in params if p.requires_grad]
        sizes = [p.numel() * (2 if p.is_complex() else 1) for p in self.params]
        dev = self.params[0].device
        n_cplx = sum(1 for p in self.params if p.is_complex())
        self.flat = torch.zeros(sum(sizes) + n_cplx, dtype=torch.float32, device=dev)      # + room for the alignment gaps
        offsets = []
        self.views = []
        off = 0
        for p, n in zip(self.params, sizes):
            if p.is_complex() and off % 2:
                off += 1                    # view_as_complex needs an even element offset (odd count of real entries in front)
            seg = self.flat[off:off + n]
            if p.is_complex():
                view = torch.view_as_complex(seg.view(*p.shape, 2))
            else:
                assert p.dtype == torch.float32
                view = seg.view(p.shape)
            p.grad = None
            p._uno_grad_buffer = view       # where the weight-gradient kernels write this parameter's gradient (integral_operators._grad_target)
            self.views.append(view)
            offsets.append(off)
            off += n
        # buckets: walk the parameters backwards, close a bucket once it holds bucket_mb
        limit = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []               # (start, end) element ranges, in issue order
        self._bucket_of = {}
        end = off
        count = 0
        members = []
        for i in range(len(self.params) - 1, -1, -1):
            members.append(i)
            count += sizes[i]
            if count >= limit or i == 0:
                for j in members:
                    self._bucket_of[j] = len(self.buckets)
                self.buckets.append((offsets[i], end))
                end = offsets[i]
                count = 0
                members = []
        self._bucket_params = [sum(1 for b in self._bucket_of.values() if b == k) for k in range(len(self.buckets))]
        self._pending = list(self._bucket_params)
        self._next = 0                  # next bucket to issue
        self._works = []
        self._group = None
        self._armed = False
        self._hooks = []
        self.trace = None               # set to [] to record, per issued bucket, the host time since arm() (bench.py `comm`)
        self._t_arm = 0.0

    def zero_(self):
        """Start of a step: every .grad is released (None).  `.flat` is NOT cleared here: it is valid only after finish() / collect()
        of the pass that follows (which also zero the segments of parameters that received no gradient).  The backward pass then writes each gradient ONCE into its view of the
        flat buffer - the library's weight-gradient kernels write there directly and autograd adopts an alias of the view as
        .grad (no zero fill, no `.grad +=` pass); gradients that arrive as ordinary tensors (a few small ones: normalisation
        affines, the final projection) are copied in by collect()."""
        for p in self.params:
            p.grad = None

    def collect(self, only=None):
        """Make every existing .grad live in the flat buffer: a gradient that autograd produced outside it is copied into its view
        and .grad re-pointed (called per parameter by the bucket hooks, and for all parameters at the end of a backward pass)."""
        pairs = zip(self.params, self.views) if only is None else ((self.params[only], self.views[only]),)
        for p, view in pairs:
            g = p.grad
            if g is None:
                if only is None:            # end of a pass: a parameter that took no part in it contributes ZERO to the sum over ranks
                    view.zero_()            # (zero_() does not clear the buffer, so its segment still holds an earlier step's values)
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
                p.grad = view.view(view.shape)

    # ------------------------------------------------------------------ overlapped all-reduce
    def arm(self, group=None, force=False):
        """Call before backward(): enables the bucket hooks for this backward if the group has more than one rank."""
        if self._works:                 # the previous pass neither finished nor aborted (it raised and the caller went on)
            self.abort()
        self._armed = bool(dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1))
        if not self._armed:
            return
        self._group = group
        self._t_arm = time.perf_counter()
        self._pending = list(self._bucket_params)
        self._next = 0
        self._works = []
        if not self._hooks:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        k = self._bucket_of[i]

        def hook(_param):
            if self._armed:
                self.collect(i)             # the gradient must be in the flat buffer before its bucket is sent
                self._pending[k] -= 1
                self._issue_ready()
        return hook

    def _issue(self, k):
        a, b = self.buckets[k]
        if self.trace is not None:
            self.trace.append(time.perf_counter() - self._t_arm)
        if self.comm_dtype is None:
            self._works.append((dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self._group, async_op=True), None, a, b))
        else:
            low = self.flat[a:b].to(self.comm_dtype)        # rounded copy (stream-ordered behind the kernels that wrote the bucket)
            self._works.append((dist.all_reduce(low, op=dist.ReduceOp.SUM, group=self._group, async_op=True), low, a, b))

    def _issue_ready(self):
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            self._issue(self._next)
            self._next += 1

    def finish(self):
        """Call after backward(): issues the buckets that are still open and waits for all of them."""
        self.collect()
        if not self._armed:
            return
        while self._next < len(self.buckets):
            self._issue(self._next)
            self._next += 1
        for w, low, a, b in self._works:
            w.wait()
            if low is not None:
                self.flat[a:b].copy_(low)                   # widened sum back into the float32 buffer the optimiser reads
        self._works = []
        self._armed = False

    def abort(self):
        """A backward pass that raised (out of memory, a user interrupt): wait for the collectives already issued - every rank issued the
        same ones up to its failure or will hang with us, which is the caller's to handle - and return to the un-armed state, so that the
        next arm() starts from bucket 0 with full counters.  The flat buffer holds a partial gradient: zero_() + a new pass overwrite it."""
        for w, low, a, b in self._works:
            try:
                w.wait()
            except Exception:
                pass
        self._works = []
        self._armed = False
        self._next = 0
        self._pending = list(self._bucket_params)

    def all_reduce_sum(self, group=None, force=False):
        """One blocking SUM over the whole buffer (no overlap)."""
        self.collect()
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)


def ns2d_rollout_loss(model, xx, yy, T_f, step=1):
    """Autoregressive loss of the NS-2D training step (reference ns_train_2d.py:46-62): the model predicts
    `step` frames from the last T_in, the prediction is appended to the input window, the per-step relative
    L2 losses are summed; ONE backward runs through the whole unrolled chain."""
    loss = 0
    B = yy.shape[0]
    if step == 1 and hasattr(model, "forward_cf") and hasattr(model, "get_grid"):
        # the window lives channels-first next to the model's positional features: per step ONE concatenation
        # [frames 1.., new frame, features] instead of the channels-last window, its layout change and the feature concatenation
        T_in = xx.shape[-1]
        z = torch.cat((xx.permute(0, 3, 1, 2), model.get_grid(xx.shape, xx.device).permute(0, 3, 1, 2)), dim=1)
        for t in range(T_f):
            im = model.forward_cf(z)                                # (B, 1, S, S)
            loss = loss + lp_loss_rel_sum(im.reshape(B, -1), yy[..., t:t + 1].reshape(B, -1))
            if t + 1 < T_f:
                z = torch.cat((z[:, 1:T_in], im, z[:, T_in:]), dim=1)
        return loss
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


class GraphedStep:
    """Forward + loss + backward of one training step captured ONCE into a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm)
    and replayed per step.  With a capturable optimiser (ComplexAdam(capturable=True): step count on the device, bias corrections
    evaluated there - reference Adam.py:27-52 takes them from state['step'] on the host) the update is part of the graph too; with
    any other optimiser it runs eagerly after the replay.
    Single rank only: no gradient all-reduce is issued between the replay and the update (DarcyTrainer is the data-parallel step).

    For launch-bound steps: the NS-2D roll-out (reference ns_train_2d.py:46-68) issues ~7400 kernels of 5-40 us per step and
    the host needs ~95 ms to enqueue them - as long as the device needs to run them.  A replay has no per-launch host work.
    Every kernel of this package launches on torch's current stream and allocates through torch's caching allocator, so
    the capture sees all of them (including the event fork / join onto the spectral backward's side stream); one-time set-up
    (twiddle tables, resampling tables, side streams) happens in the eager warm-up steps that precede the capture.

        gs = GraphedStep(model, opt, lambda xx, yy: ns2d_rollout_loss(model, xx, yy, 40), (xx0, yy0))
        loss = gs.step(xx, yy)        # device tensor, no host synchronisation
    """

    def __init__(self, model, opt, loss_fn, example_inputs, warmup: int = 2):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs the GPU (HIP graph capture)")
        self.model, self.opt, self.loss_fn = model, opt, loss_fn
        self.opt_in_graph = bool(getattr(opt, "capturable", False))
        if self.opt_in_graph:
            opt.init_state()                            # moments and step counters exist before the capture (nothing to re-zero on replay)
        self.static_in = tuple(t.clone() for t in example_inputs)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                   # eager warm-up off the default stream, as capture requires
            for _ in range(warmup):
                opt.zero_grad(set_to_none=True)
                loss_fn(*self.static_in).backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)                 # .grad tensors are created inside the capture: static across replays
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_loss = loss_fn(*self.static_in)
            self.static_loss.backward()
            if self.opt_in_graph:
                self.opt.step()
        self._params = [p for g in opt.param_groups for p in g["params"] if p.requires_grad]

    def step(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src, non_blocking=True)
        if self.opt_in_graph:
            self.opt.sync_hyper()                       # lr / eps / weight decay live on the device: a scheduler's edit reaches the replay
        self.graph.replay()                             # gradients are overwritten by the captured backward
        if self.opt_in_graph:
            torch.autograd.graph.increment_version(self._params)       # the replayed update wrote the parameters through raw pointers
        else:
            self.opt.step()
        return self.static_loss.detach().clone()        # the graph-owned scalar is overwritten by the next replay


class DarcyTrainer:
    """model + ComplexAdam + flat-gradient data parallelism.  step(a, u) runs forward, relative-L2 loss,
    backward, gradient all-reduce and the optimiser update; it returns the (device) loss tensor and
    never synchronises with the host."""

    def __init__(self, model, lr=1e-3, weight_decay=1e-3, group=None, force_collectives=False, bucket_mb=32.0, comm_dtype=None,
                 comm_cus=None):
        self.model = model
        self.group = group
        self.force_collectives = force_collectives      # tests: run the collectives even in a 1-rank group
        self.grads = FlatGradients(model.parameters(), bucket_mb=bucket_mb, comm_dtype=comm_dtype)
        # more than one rank: RCCL's all-reduce kernels run beside the backward pass - the library's device-sized launch geometries leave
        # them `comm_cus` compute units (default 16, UNO_COMM_CUS overrides; a single rank reserves none)
        self.comm_cus = 0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 and next(model.parameters()).is_cuda:
            import os
            from .. import _native
            self.comm_cus = int(os.environ.get("UNO_COMM_CUS", 16 if comm_cus is None else comm_cus))
            _native.reserve_cus(self.comm_cus)
        self.opt = ComplexAdam(model.parameters(), lr=lr, weight_decay=weight_decay)
        self.broadcast_parameters()

    def broadcast_parameters(self):
        if dist.is_available() and dist.is_initialized() and (self.force_collectives or dist.get_world_size(self.group) > 1):
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(torch.view_as_real(t.data) if t.is_complex() else t.data, src=0, group=self.group)

    def step_with(self, loss_closure):
        """zero grads -> loss_closure() -> backward with bucketed gradient all-reduce overlapped -> optimiser update."""
        self.grads.zero_()
        loss = loss_closure()
        self.grads.arm(self.group, self.force_collectives)
        try:
            loss.backward()
        except BaseException:
            self.grads.abort()          # collectives in flight are waited for, the bucket state is reset: the next step starts clean
            from ..integral_operators import release_pass_state
            release_pass_state()        # (autograd skips a failed pass's final callbacks: its in-place gradient map would linger)
            raise
        self.grads.finish()
        self.opt.step()
        return loss.detach()

    def step(self, a, u):
        B, S = a.shape[0], a.shape[1]
        return self.step_with(lambda: lp_loss_rel_sum(self.model(a).reshape(B, -1), u.reshape(B, -1)))

    def step_blocking(self, a, u):
        """The same step with the exchange NOT overlapped: backward first, then one blocking all-reduce of the whole flat
        buffer (the comparison figure of bench.py's `comm` object)."""
        B = a.shape[0]
        self.grads.zero_()
        loss = lp_loss_rel_sum(self.model(a).reshape(B, -1), u.reshape(B, -1))
        loss.backward()
        self.grads.all_reduce_sum(self.group, self.force_collectives)
        self.opt.step()
        return loss.detach()
