"""MI355X-native drop-in for the operator blocks of U-NO (module name kept: model code does
``from integral_operators import *`` - reference darcy_flow_uno2d.py:10, navier_stokes_uno2d.py:9,
navier_stokes_uno3d.py:6).

Same classes, constructor / forward signatures, attribute names, parameter names, shapes,
dtypes and registration order as the reference's integral_operators.py, so a reference
``state_dict`` loads with ``strict=True`` and seed-for-seed initialisation matches.  What differs is
what runs underneath the spectral convolutions:

    rFFT -> truncated-mode complex channel mixing -> zero-padded iRFFT

is executed by hand-written gfx950 kernels behind the C ABI in include/uno_spectral.h (pruned
forward DFT, per-mode complex MFMA GEMM, pruned inverse DFT; custom autograd with the same kernel
family).  The full spectrum is never materialised.  There is no CPU fallback for these layers: a
tensor that is not on a HIP device raises ``RuntimeError``.

The rest of an operator block runs on the same library for float32 device tensors: the point-wise branch (1x1
convolution = channel-mix kernels, bicubic anti-aliased resampling = banded separable kernels) accumulates into the
spectral branch's output buffer, InstanceNorm (+ GELU) is one kernel; only the GELU of non-normalised blocks and the
skip concatenations are stock PyTorch-ROCm ops (SpectralConv1d_Uno runs on the 2-D kernels, one row; pointwise_op_3D's FFT
resampling on the pruned-DFT kernels with explicit frequency tables).  CPU tensors take stock torch ops in these helper
layers (they are not part of the spectral path and the CPU-side harness tests use them with the oracle blocks).
"""
from __future__ import annotations

import threading
import time

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _native

__all__ = [
    "enable_mixed_precision", "GradJoin", "channel_mix_cat_project", "release_pass_state",
    "SpectralConv1d_Uno", "pointwise_op_1D", "OperatorBlock_1D",
    "SpectralConv2d_Uno", "pointwise_op_2D", "OperatorBlock_2D",
    "SpectralConv3d_Uno", "pointwise_op_3D", "OperatorBlock_3D",
]


# activation dtypes the device kernels take: float32 (the reference's contract) and bfloat16 (mixed precision, BASELINE.json
# configs[4]: opt-in per spectral layer via enable_mixed_precision - weights, statistics and accumulations stay float32)
_ACT = (torch.float32, torch.bfloat16)


def _dev_act(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype in _ACT


def enable_mixed_precision(module: nn.Module, enabled: bool = True) -> nn.Module:
    """Let the 2-D spectral layers under `module` take bfloat16 activations (the reference raises on them, integral_operators.py:187,
    and so do these layers unless enabled here).  In that mode a layer reads its complex weights through a half-precision
    (re, im) copy made per call (the parameters themselves - and their gradients - stay complex64: the master copy the
    optimiser updates), transforms bf16 images directly and accumulates in f32 / c64; outputs and input gradients are bf16."""
    for m in module.modules():
        if isinstance(m, SpectralConv2d_Uno):
            m.mixed_precision = bool(enabled)
    return module


def _half_weights(w1, w2):
    """(Ci, Co, m1, m2, 2) float16 copies of two complex64 weight tensors (storage format of the mixed-precision kernels).  The copy
    of a parameter is kept on it and re-made only when the parameter changed (its version counter moves with every in-place
    update - the optimiser step): repeated forward passes between updates (evaluation, roll-outs) convert nothing."""
    out = []
    # while a HIP graph is being captured the conversion must be PART of the graph: the optimiser updates the master weights between
    # replays (harness.GraphedStep runs it eagerly), and a copy made at warm-up and found in the cache would never be re-made - the
    # replays would read frozen weights.  The captured conversion re-reads the parameter on every replay.
    capturing = w1.is_cuda and torch.cuda.is_current_stream_capturing()
    with torch.no_grad():
        for w in (w1, w2):
            cached = None if capturing else getattr(w, "_uno_half", None)
            if cached is None or cached[0] != w._version or cached[1].device != w.device or cached[2] != w.data_ptr():
                cached = (w._version, torch.view_as_real(w.detach()).half().contiguous(), w.data_ptr())
                if not capturing:
                    try:
                        w._uno_half = cached
                    except (AttributeError, RuntimeError):
                        pass
            out.append(cached[1])
    return out[0], out[1]


def _plain(t: torch.Tensor) -> torch.Tensor:
    """Materialise lazy conj/neg views and non-contiguous layouts (the C ABI takes dense buffers;
    the reference accepts any strides - integral_operators.py:187 goes through torch.fft)."""
    if t.is_complex() and t.is_conj():
        t = t.resolve_conj()
    if t.is_neg():
        t = t.resolve_neg()
    if t.is_contiguous():
        return t
    # channels-last activations and gradients (what the reference's model files hand over: darcy_flow_uno2d.py:104-107, :126) go
    # through the tiled transposing copy; every other layout through torch's strided copy
    if t.is_cuda and t.dtype == torch.float32 and _native.channels_last_pitch(t) is not None:
        return _native.to_channels_first(t)
    return t.contiguous()


def _check_input(x: torch.Tensor, ndim: int, channels: int, who: str):
    if x.dim() != ndim:
        raise RuntimeError(f"{who}: expected a {ndim}-D tensor (batch, channels, *grid), got shape {tuple(x.shape)}")
    if x.shape[1] != channels:
        raise RuntimeError(f"{who}: expected {channels} input channels, got {x.shape[1]}")
    if x.dtype != torch.float32:
        # the reference raises as well: its out_ft is hard-coded cfloat, so a float64 input dies in the
        # einsum (integral_operators.py:179) and half/bfloat16 die in rfft2 (:187)
        raise RuntimeError(f"{who}: input must be float32 (got {x.dtype})")


class _SpectralConv2dFn(torch.autograd.Function):
    """y = irfft2(corner-mix(rfft2(x)));  saves only the truncated input spectrum."""

    @staticmethod
    def forward(ctx, x, w1, w2, Ho, Wo, half_weights=False):
        ctx.params = (w1, w2)
        x = _plain(x)
        ctx.stack = _stack_take(w1, (x.shape[0], x.shape[1], 2 * w1.shape[2], w1.shape[3]), x.device, _stack_wanted(ctx, 1, x, half_weights))
        w1, w2 = _plain(w1), _plain(w2)
        if half_weights:                    # complex64 master weights, read through float16 (re, im) copies
            w1, w2 = _half_weights(w1, w2)
        y, xt = _native.spectral_conv2d_forward(x, w1, w2, int(Ho), int(Wo), xt_out=None if ctx.stack is None else ctx.stack[0].X[ctx.stack[1]])
        ctx.save_for_backward(xt, w1, w2)
        ctx.in_hw = (x.shape[-2], x.shape[-1])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xt, w1, w2 = ctx.saved_tensors
        gx, gw1, gw2, _ = _spectral_backward(_plain(gy), xt, w1, w2, ctx.in_hw[0], ctx.in_hw[1], ctx.needs_input_grad[0],
                                          ctx.needs_input_grad[1] or ctx.needs_input_grad[2],
                                          ctx.needs_input_grad[1] and ctx.needs_input_grad[2], ctx.params, ctx.stack)
        return gx, gw1, gw2, None, None, None


class _ChannelMixFn(torch.autograd.Function):
    """y[b] = W . x[b] + bias on (B, C, pixels) views with the K8 / K9 kernels (csrc/channel_mix.hip)."""

    @staticmethod
    def forward(ctx, x, w, bias, leaves=None):
        x, w = _plain(x), _plain(w)
        y = _native.channel_mix(x, w, None if bias is None else _plain(bias))
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.leaves = leaves
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _plain(gy)
        gx = _native.channel_mix(gy, w, None, transpose_w=True) if ctx.needs_input_grad[0] else None
        gw, gb = _wgrad_into(ctx.leaves, gy, x, None, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return gx, gw, gb, None


def channel_mix(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """1x1 convolution of a channels-first tensor, y[b] = W . x[b] (+ bias) on the (B, C, pixels) view - no
    layout change, no im2col.  `weight` is a Conv (Co, Ci, 1, ...) or Linear (Co, Ci) weight.  float32 tensors on
    a HIP device run the K8 / K9 kernels; anything else (the CPU-side harness tests, other dtypes) is a stock
    batched matmul - this helper is not part of the spectral path and keeps torch semantics there."""
    B, Ci = x.shape[0], x.shape[1]
    w = weight.reshape(weight.shape[0], Ci)
    xv = x.reshape(B, Ci, -1)
    if _dev_act(x) and w.dtype == torch.float32:
        y = _ChannelMixFn.apply(xv, w, bias, (weight, bias))
    elif bias is not None:
        y = torch.baddbmm(bias.view(1, -1, 1), w.unsqueeze(0).expand(B, -1, -1), xv)
    else:
        y = torch.matmul(w, xv)
    return y.view(B, w.shape[0], *x.shape[2:])


# Weight-gradient kernels write a parameter's gradient where it will live instead of handing autograd fresh tensors to sum:
#   * FIRST contribution to a parameter in a backward pass: the kernel writes (beta = 0) into the parameter's registered buffer
#     (`_uno_grad_buffer`, set by harness.FlatGradients: a view into the flat all-reduce buffer) or into a fresh tensor, and the
#     backward returns an ALIAS of it - autograd adopts the alias as .grad when the pass ends (no zero fill, no `.grad +=` pass;
#     post-accumulate hooks - the bucketed all-reduce - fire as usual);
#   * LATER contributions in the same pass (a layer used several times in one graph: the 40-step roll-out of ns_train_2d.py:46-68
#     sums 40 gradients per weight; autograd would add them one by one in the input buffer of the parameter's AccumulateGrad
#     node): the kernel adds (beta = 1) into that same tensor and the backward returns None for the parameter.
# A parameter that already HAS a .grad when the pass starts (accumulation across passes) takes the ordinary path.
INPLACE_PARAM_GRADS = True
# State of the backward passes in flight, keyed by autograd's graph-task id (a nested pass - re-entrant activation checkpointing,
# torch.autograd.grad inside a hook - is its own task with its own state; the outer pass finds its state untouched when it resumes):
#   acc:    id(parameter) -> [tensor its gradient is being summed in, parameter, contributions so far, touched by a nested pass]
#   stacks: id(stack) -> (stack, weight leaves, weight shape, [slots whose gradient spectrum arrived in this pass])
#   uses:   id(weights1 leaf) -> [leaf, [spectral-layer backward calls of this pass that did NOT go through a stack]]
_PASSES = {}
_PASSES_LOCK = threading.Lock()
_STALE_PASS_SECONDS = 3600.0
_SWEPT = {}                 # ids of swept passes (bounded): a pass that shows up again after its state was released must not go on silently


def release_pass_state():
    """Drop the state of every backward pass on record.  For a training loop that caught an exception out of loss.backward(): autograd
    skips the final callbacks of a pass that raised, so its entry would otherwise wait for the time-based sweep.  Only call while no
    backward pass is running on any thread of this process (harness.DarcyTrainer does, from its except path)."""
    with _PASSES_LOCK:
        _PASSES.clear()


def _sweep_stale_passes():
    """Autograd skips a pass's final callbacks when the pass raises (an out-of-memory error the training loop catches and retries):
    its entry would stay in _PASSES for ever - graph-task ids are never reused - keep its gradient buffers alive and push every
    parameter it recorded off the in-place path.  Called from a thread that is NOT inside a backward pass; an entry older than
    _STALE_PASS_SECONDS seen from there belongs to no pass that could still be running (a live pass of another thread is younger
    than that by orders of magnitude: the longest step of this package is a fraction of a second).  Should a live pass be swept after
    all (an hour in a debugger), it raises at its next contribution instead of training on a partial gradient (_SWEPT)."""
    now = time.monotonic()
    with _PASSES_LOCK:
        for tid in [t for t, ps in _PASSES.items() if now - ps["born"] > _STALE_PASS_SECONDS]:
            _PASSES.pop(tid, None)
            _SWEPT[tid] = now
        while len(_SWEPT) > 256:
            _SWEPT.pop(next(iter(_SWEPT)))


# The pass state hangs on two private entry points of autograd (the id of the running graph task, the engine's final-callback
# queue).  Should a torch release drop either, the library falls back to the ordinary path - every weight-gradient kernel returns
# its gradient as a fresh tensor and autograd accumulates - instead of failing: slower (one zero fill + one add per parameter
# and use), same results.
_current_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_queue_callback = getattr(getattr(torch.autograd.Variable, "_execution_engine", None), "queue_callback", None)
_PASS_STATE_AVAILABLE = _current_graph_task_id is not None and _queue_callback is not None


def _graph_task_id() -> int:
    return _current_graph_task_id() if _PASS_STATE_AVAILABLE else -1


def _pass_state():
    """The dictionaries of the running backward pass (registered with the engine on first use), or None outside a pass."""
    tid = _graph_task_id()
    if tid < 0:
        if _PASSES:
            _sweep_stale_passes()
        return None
    ps = _PASSES.get(tid)
    if ps is None:
        if tid in _SWEPT:
            # wall-clock age is only a heuristic for "this pass raised": a LIVE pass that was paused for longer than the limit (debugger,
            # contended device) lost its in-place accumulation map when it was swept - later contributions would overwrite earlier ones
            raise RuntimeError(f"uno_amd: backward pass {tid} was idle for more than {_STALE_PASS_SECONDS:.0f} s and its in-place gradient "
                               "state was released; raise uno_amd.integral_operators._STALE_PASS_SECONDS or set INPLACE_PARAM_GRADS = False")
        if _PASSES:
            _sweep_stale_passes()              # (a pass that raised never ran its callback; see there)
        with _PASSES_LOCK:
            ps = _PASSES.get(tid)
            if ps is None:
                ps = _PASSES[tid] = {"id": tid, "acc": {}, "stacks": {}, "uses": {}, "born": time.monotonic()}
                # final callbacks belong to the graph task that is current when they are queued: this one runs when THIS pass completes
                _queue_callback(lambda: _end_of_pass(tid))
    return ps


def _end_of_pass(tid):
    """End of a backward pass (engine callback: every node, AccumulateGrad included, has run).  A parameter that received SEVERAL
    contributions in place must now have a .grad that aliases the tensor they were summed in; if it does not, autograd replaced
    that tensor on the way (a gradient for the same parameter from a path outside this library was added out of place) and the
    later in-place contributions would be missing - fail loudly instead of training on a wrong gradient.
    Spectral layers: remember how often each was used in this pass (the next forward passes stack that many spectra, see
    _SpectrumStack), and finish the stacks of which only a part of the uses was back-propagated."""
    with _PASSES_LOCK:
        ps = _PASSES.pop(tid, None)
    if ps is None:
        return
    acc, stacks, uses = ps["acc"], ps["stacks"], ps["uses"]
    for leaf, count in uses.values():
        if not getattr(leaf, "_uno_nostack", False):
            leaf._uno_uses = count[0]
    for st, leaves, wshape, slots in stacks.values():
        _stack_flush_partial(st, leaves, wshape, slots)
    for t, param, count, nested in acc.values():
        # nested: a pass that ran INSIDE this one gave the parameter a .grad of its own before this pass's AccumulateGrad ran; the
        # tensor summed here was then added to that .grad as a whole (complete: AccumulateGrad runs after every contribution)
        if count[0] > 1 and not nested[0] and param.grad is not None and param.grad.data_ptr() != t.data_ptr():
            raise RuntimeError("uno_amd: a parameter's gradient was accumulated in place by the library's kernels, but autograd also "
                               "received gradients for it from other operations and replaced the buffer; set "
                               "uno_amd.integral_operators.INPLACE_PARAM_GRADS = False for this model")


def _grad_plan(p, ps):
    """('acc', tensor): later contribution of this pass | ('new', registered buffer or None): first contribution | None: ordinary path"""
    if not (INPLACE_PARAM_GRADS and _PASS_STATE_AVAILABLE) or not isinstance(p, torch.Tensor) or not p.is_leaf or not p.requires_grad or not p.is_cuda:
        return None
    if ps is None:
        return None
    acc = ps["acc"].get(id(p))
    if acc is not None:
        return "acc", acc[0]
    if p.grad is not None:
        return None
    if len(_PASSES) > 1:
        # another pass is in flight (this one is nested in it, or the other way round): if it is summing this parameter's gradient
        # in place, its tensor - possibly the registered buffer - must not be overwritten by a beta = 0 write from here
        busy = False
        for other in list(_PASSES.values()):
            rec = other["acc"].get(id(p)) if other is not ps else None
            if rec is not None:
                rec[3][0] = True
                busy = True
        if busy:
            return None
    buf = getattr(p, "_uno_grad_buffer", None)
    if buf is not None and (buf.shape != p.shape or buf.dtype != p.dtype or buf.device != p.device or not buf.is_contiguous()):
        buf = None
    return "new", buf


def _grad_targets(params):
    """Targets of the parameters ONE kernel call writes together: all or nothing, one accumulate flag.
    -> list of (destination tensor, accumulate flag, value to return to autograd) or None"""
    ps = _pass_state()
    plans = [_grad_plan(p, ps) for p in params]
    if any(pl is None for pl in plans) or len({pl[0] for pl in plans}) != 1:
        return None
    if plans[0][0] == "acc":
        for p in params:
            ps["acc"][id(p)][2][0] += 1
        return [(pl[1], True, None) for pl in plans]
    out = []
    for p, pl in zip(params, plans):
        buf = pl[1] if pl[1] is not None else torch.empty(p.shape, dtype=p.dtype, device=p.device)
        ps["acc"][id(p)] = (buf, p, [1], [False])       # (tensor the gradient is summed in, parameter, contributions so far, nested)
        out.append((buf, False, buf.view(buf.shape)))
    return out


def _wgrad_into(leaves, gy, x1, x2, need_w, need_b, act_x=False, stack=None, window=None):
    """Weight / bias gradient of a channel-mix layer (K9), written in place where the layer's leaf parameters allow it.
    leaves = (weight leaf, bias leaf or None) or None.  -> (gw or None shaped (Co, Ci), gb or None) to return to autograd.
    stack = (stack, slot) of the block's spectral layer when that is batching its weight gradient over the uses of the pass: the
    second stage of this gradient is deferred to the last use as well (_stack_pointwise).  window: see _native.channel_mix2."""
    if not (need_w or need_b):
        return None, None
    has_bias = need_b
    tg = None
    fused = x2 is None or (x1.shape[1] % 64 == 0 and gy.shape[2] >= 64)
    if window is not None:
        if not fused:
            raise RuntimeError("uno_amd: a windowed two-source layer splits its sources at a multiple of 64 channels")
        stack = None
    if stack is not None and fused and leaves is not None and need_w and (leaves[1] is not None) == has_bias and gy.dtype == torch.float32 \
            and all(isinstance(t, torch.Tensor) and t.is_leaf for t in leaves if t is not None):
        out = _stack_pointwise(stack, leaves, gy, x1, x2, has_bias, act_x)
        if out is not NotImplemented:
            return out
    if fused and leaves is not None and need_w and (leaves[1] is not None) == has_bias:
        tg = _grad_targets([leaves[0]] + ([leaves[1]] if has_bias else []))        # committed: the call below writes them
    if tg is not None:
        _native.channel_wgrad2(gy, x1, x2, need_bias=has_bias, act_x=act_x, out_w=tg[0][0], out_b=tg[1][0] if has_bias else None,
                               accumulate=tg[0][1], window=window)
        Co, Ci = gy.shape[1], x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
        gw = None if tg[0][2] is None else tg[0][2].view(Co, Ci)
        return gw, (tg[1][2] if has_bias else None)
    if window is not None:
        return _native.channel_wgrad2(gy, x1, x2, need_bias=has_bias, act_x=act_x, window=window)
    if x2 is None:
        return _native.channel_wgrad(gy, x1, need_bias=has_bias, act_x=act_x)
    return _mix2_wgrad(gy, x1, x2, has_bias, act_x=act_x)


# ---- weight gradient of a spectral layer that is used SEVERAL times in one graph (the 40-step roll-out of ns_train_2d.py:46-68
# calls every layer 40 times before one backward), batched over the uses.
# gW[i, o, mode] = sum_t sum_b conj(X_t[b, i, mode]) gO_t[b, o, mode]: executed per use that is 40 per-mode GEMMs with K = batch
# (32) that each read and re-write the whole weight gradient (2 x 16-26 MB for 8-16 MB of operands: 68 us per call, 15 ms of the
# 88 ms NS-2D step).  Instead the layer keeps the truncated spectra of its uses in ONE tensor (T, B, Ci, 2 m1, m2) - K1 of use t
# writes slot t in the forward pass, K1 of the output gradient writes slot t of a second tensor in the backward pass - and the use
# whose backward comes LAST runs one GEMM with K = T B over both and hands the complete gradient to autograd (the other uses
# return None for the weights).  Nothing is copied; the spectra were saved for the backward pass anyway.
# How many slots to provide is the number of uses the layer saw in the previous backward pass (`_uno_uses` on the weights1 parameter,
# stacked or not; the first pass runs use by use, a pass with more uses than slots fills several stacks and the next one is sized
# for all of them).  A stack is closed for new uses once a backward pass touched it or the weights changed; a pass that
# back-propagates only some of a stack's uses finishes it at the end of the pass (gradient added to .grad directly) and turns the
# stacking off for that layer.
TIME_BATCHED_WGRAD = True


class _SpectrumStack:
    __slots__ = ("X", "G", "n", "sealed", "done", "version", "P", "Pinfo")

    def __init__(self, cap, shape, device, version):
        self.X = torch.empty((cap, *shape), dtype=torch.complex64, device=device)     # truncated input spectra, slot per use
        self.G = None               # truncated output-gradient spectra (allocated for the slots in use when the first one arrives)
        self.P = None               # (n, floats) split-K partial sums of the block's 1x1 convolution weight gradient, row per use
        self.Pinfo = None           # (Ci, Co, has_bias, (weight leaf, bias leaf)) of those
        self.n = 0                  # slots handed out
        self.sealed = False         # a backward pass has started on it: no new uses
        self.done = False           # its gradient has been produced: late backward calls (retain_graph) run on their own
        self.version = version


def _stack_take(leaf, shape, device, wanted):
    """Forward pass of a spectral layer: (stack, slot) for this use's truncated input spectrum, or None (layer used once per pass,
    no gradient wanted, stacking off)."""
    if not (TIME_BATCHED_WGRAD and wanted and INPLACE_PARAM_GRADS and _PASS_STATE_AVAILABLE) or not isinstance(leaf, torch.Tensor) or not leaf.is_leaf:
        return None
    cap = getattr(leaf, "_uno_uses", 0)
    # the per-mode GEMM addresses an operand with 32-bit byte offsets: a stack (and the stack of output-gradient spectra) stays under 2 GiB
    per_slot = 8 * shape[0] * max(shape[1], leaf.shape[1]) * shape[2] * shape[3]
    cap = min(cap, (2 ** 31 - 4096) // max(per_slot, 1))
    if cap < 2 or getattr(leaf, "_uno_nostack", False):
        return None
    st = getattr(leaf, "_uno_stack", None)
    if st is None or st.sealed or st.n >= st.X.shape[0] or tuple(st.X.shape[1:]) != tuple(shape) or st.X.device != device \
            or st.version != leaf._version:
        st = _SpectrumStack(cap, shape, device, leaf._version)
        try:
            leaf._uno_stack = st
        except (AttributeError, RuntimeError):
            return None
    st.n += 1
    return st, st.n - 1


def _stack_grad_slot(st, slot, Co):
    """Backward pass: where K1 writes the truncated spectrum of this use's output gradient, or None when the stack is finished."""
    if st.done:
        return None
    st.sealed = True
    if st.G is None:
        T, B, _, r2, m2 = st.X.shape
        st.G = torch.empty((st.n, B, Co, r2, m2), dtype=torch.complex64, device=st.X.device)
    return st.G[slot]


def _stack_wgrad(st, lo, hi, leaves, wshape, in_place):
    xt, go = st.X[lo:hi].flatten(0, 1), st.G[lo:hi].flatten(0, 1)
    tg = _grad_targets(leaves) if in_place else None
    gw1, gw2 = _native.mode_wgrad(xt, go, tuple(wshape[:4]), 2, out=[tg[0][0], tg[1][0]] if tg else None,
                                  accumulate=bool(tg and tg[0][1]))
    return (tg[0][2], tg[1][2]) if tg else (gw1, gw2)


def _stack_arrived(st, slot, leaves, wshape, in_place):
    """This use's gradient spectrum is in its slot.  -> (gw1, gw2) when it was the last of the stack's uses, else (None, None)."""
    ps = _pass_state()
    ps["uses"].setdefault(id(leaves[0]), [leaves[0], [0]])[1][0] += 1      # every use of the pass counts: the next stacks hold them all
    rec = ps["stacks"].setdefault(id(st), (st, leaves, wshape, []))
    rec[3].append(slot)
    if len(rec[3]) < st.n:
        return None, None
    del ps["stacks"][id(st)]
    out = _stack_wgrad(st, 0, st.n, leaves, wshape, in_place)
    st.done, st.G = True, None
    return out


def _stack_pointwise(stack, leaves, gy, x1, x2, has_bias, act_x):
    """The 1x1 convolution's weight gradient of a block whose spectral layer is stacked: K9's first stage leaves this use's split-K
    partial sums in row `slot` of the stack's (n, floats) buffer, and the use that completes the stack runs ONE second stage over
    all rows (the roll-out ran 40 second stages of ~5 us per layer; their read-modify-write of the gradient goes with them).
    -> (gw (Co, Ci), gb) for autograd ((None, None) until the last use), or NotImplemented: take the ordinary path for this call."""
    st, slot = stack
    B, Co, P = gy.shape
    Ci = x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
    nf = _native.channel_wgrad_partial_floats(B, Ci, Co, P)
    if st.P is None:
        if st.Pinfo is not None:
            return NotImplemented               # the stack's buffer has been consumed (late call on a retained graph)
        st.P = torch.empty((st.n, nf), dtype=torch.float32, device=gy.device)
        st.Pinfo = (Ci, Co, has_bias, leaves)
    fits = st.P.shape[1] == nf and st.Pinfo[:3] == (Ci, Co, has_bias)
    if fits:
        _native.channel_wgrad2(gy, x1, x2, need_bias=has_bias, act_x=act_x, partials_out=st.P[slot])
    else:
        st.P[slot].zero_()                      # another grid than the stack's other uses: this use is computed on its own
    if not st.done:
        return (None, None) if fits else NotImplemented
    # the spectral half of this backward call completed the stack: every row is written
    Ci0, Co0, hb0, lv = st.Pinfo
    tg = _grad_targets([lv[0]] + ([lv[1]] if hb0 else []))
    gw, gb = _native.channel_wgrad_finish(st.P, Ci0, Co0, hb0, out_w=tg[0][0] if tg else None,
                                          out_b=tg[1][0] if (tg and hb0) else None, accumulate=bool(tg and tg[0][1]))
    st.P = None
    if not fits:
        _native.channel_wgrad2(gy, x1, x2, need_bias=hb0, act_x=act_x, out_w=gw, out_b=gb, accumulate=True)
    if tg:
        gw, gb = (None if tg[0][2] is None else tg[0][2].view(Co0, Ci0)), (tg[1][2] if hb0 else None)
    return gw, gb


def _stack_flush_partial(st, leaves, wshape, slots):
    """End of a pass that back-propagated only `slots` of the stack's uses: their weight gradient goes to .grad directly (the
    parameters' AccumulateGrad nodes have run), the remaining uses - if a later pass reaches them - run one by one."""
    import warnings
    slots = sorted(slots)
    with torch.no_grad():
        tot = None
        k = 0
        while k < len(slots):
            e = k
            while e + 1 < len(slots) and slots[e + 1] == slots[e] + 1:
                e += 1
            g = _stack_wgrad(st, slots[k], slots[e] + 1, leaves, wshape, False)
            tot = g if tot is None else (tot[0] + g[0], tot[1] + g[1])
            k = e + 1
        for p, g in zip(leaves, tot):
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        if st.P is not None:                    # the block's 1x1 convolution: second stage over the rows that were written
            Ci0, Co0, hb0, lv = st.Pinfo
            ptot = None
            k = 0
            while k < len(slots):
                e = k
                while e + 1 < len(slots) and slots[e + 1] == slots[e] + 1:
                    e += 1
                g = _native.channel_wgrad_finish(st.P[slots[k]:slots[e] + 1], Ci0, Co0, hb0)
                ptot = g if ptot is None else (ptot[0] + g[0], (ptot[1] + g[1]) if hb0 else None)
                k = e + 1
            for p, g in ((lv[0], ptot[0]), (lv[1] if hb0 else None, ptot[1])):
                if p is not None:
                    g = g.view(p.shape)
                    if p.grad is None:
                        p.grad = g
                    else:
                        p.grad.add_(g)
            st.P = None
    st.done, st.G = True, None
    leaves[0]._uno_uses, leaves[0]._uno_nostack = 0, True
    warnings.warn("uno_amd: a backward pass covered only some of the uses of a spectral layer whose weight gradient is batched "
                  "over its uses (TIME_BATCHED_WGRAD); the gradient of this pass was added to .grad after the pass (gradient hooks "
                  "did not see it) and the batching is now off for this layer", RuntimeWarning, stacklevel=2)


def _note_use(leaf):
    """A spectral layer's backward ran outside a stack: count it (what the next forward passes size their stack by)."""
    ps = _pass_state()
    if ps is not None and isinstance(leaf, torch.Tensor) and leaf.is_leaf:
        ps["uses"].setdefault(id(leaf), [leaf, [0]])[1][0] += 1


def _stack_wanted(ctx, iw, x, half_weights):
    return bool(ctx.needs_input_grad[iw] and ctx.needs_input_grad[iw + 1] and x.dtype == torch.float32 and not half_weights)


# Up-sampling blocks (and the input gradient of down-sampling blocks): the inverse transform adds the resampled low-resolution result of
# the point-wise branch in the registers it holds its own result in, before the output tile is written (uno_dft2d_inverse_add), instead
# of K3 writing the block output and the accumulating resampling kernel reading and re-writing it.  False: the two-kernel form (A/B).
FUSE_UPSAMPLE_ADD = True
# Layers whose backward could take the composite entry point (no join, no stack, no addend): True - stage by stage with both per-mode
# GEMMs in ONE launch (uno_mode_backward); False - uno_spectral_conv2d_backward, which runs the weight-gradient GEMM on a side stream
# beside the input-gradient GEMM and the inverse transform (A/B switch; tools/dev/fusetime.py)
PAIR_BACKWARD_GEMMS = True
# A down-sampling resampling kernel reads the tensor that the block's forward transform (K1) reads as well.  True: it runs right BEFORE
# that K1 and walks the images in descending order, so that what it read last - the part of the tensor still in the 256 MB Infinity
# Cache - is what K1, walking up, reads first.  False: after the spectral branch, ascending (A/B switch).
REVERSE_SWEEP_RESAMPLE = True
# The backward pass of `fc2(F.gelu(fc1(cat)))` (reference darcy_flow_uno2d.py:125-131).  True: uno_project_backward where it applies - the
# gradient at fc1's output is formed inside the input-gradient and weight-gradient kernels from fc1's saved output; False: written by
# uno_gelu_project_backward and read back by the two (A/B switch; tools/dev/fusetime.py).
PROJECT_BACKWARD_FUSED = True


def _fused_addend(t, H, W, m1, m2, adjoint):
    """(t, operand tables) for _native.dft2d_inverse(addend=): t (B, C, Hs, Ws) float32 is the low-resolution tensor whose resampling to
    (H, W) - resample_forward, or resample_adjoint of an (H, W) input grid when `adjoint` - is to be added to the inverse transform of a
    (B, C, 2 m1, m2) spectrum; None where the fused kernel does not apply (the caller runs the two kernels)."""
    if not FUSE_UPSAMPLE_ADD or t.dtype != torch.float32 or t.dim() != 4 or not t.is_cuda:
        return None
    Hs, Ws = t.shape[-2], t.shape[-1]
    if Hs * Ws >= H * W or not _native.dft2d_inverse_add_applies(t.shape[0] * t.shape[1], H, W, m1, m2, Hs, Ws):
        return None
    from .resample import upsample_add_tables
    tabs = upsample_add_tables(Hs, Ws, H, W, str(t.device), bool(adjoint))
    return None if tabs is None else (t, tabs)


def _spectral_backward(gs, xt, w1, w2, H, W, need_gx, need_gw, both_gw, leaves, stack, join=None, addend=None):
    """Backward of the spectral branch: -> (gx or None, gw1, gw2 as autograd should receive them, whether the layer's stack took the call).
    leaves = (weights1, weights2) as the caller passed them (in-place gradient targets); stack = (stack, slot) of the forward pass
    or None; join: GradJoin whose deferred spectra are merged into this layer's before the inverse transform; addend: _fused_addend(...)
    of the point-wise branch's contribution to gx (float32 only) - the call then runs stage by stage."""
    B, Co = gs.shape[:2]
    Ci, _, m1, m2 = w1.shape[:4]
    gslot = _stack_grad_slot(stack[0], stack[1], Co) if (stack is not None and need_gw) else None
    merging = join is not None and need_gx and bool(join.spectra)
    if addend is not None and not need_gx:
        raise RuntimeError("uno_amd: an addend for the input gradient needs the input gradient")
    if gslot is None and not merging and addend is None and not (PAIR_BACKWARD_GEMMS and need_gx and need_gw and w1.dtype == torch.complex64):
        tg = _grad_targets(leaves) if (need_gw and both_gw) else None
        if need_gw:
            _note_use(leaves[0])
        gx, gw1, gw2 = _native.spectral_conv2d_backward(gs, xt, w1, w2, H, W, need_gx=need_gx, need_gw=need_gw,
                                                        gw_out=(tg[0][0], tg[1][0]) if tg else None,
                                                        accumulate_gw=bool(tg and tg[0][1]))
        if tg:
            gw1, gw2 = tg[0][2], tg[1][2]
        return gx, gw1, gw2, False
    # stage by stage: the gradient spectrum goes to its slot of the layer's stack and / or the deferred gradient spectra of x's
    # other consumer are added to this layer's before ONE inverse transform
    gO = _native.dft2d_forward(gs, m1, m2, 1.0, True, True, out=gslot)
    gw1 = gw2 = None
    gX = None
    if gslot is not None:
        gw1, gw2 = _stack_arrived(stack[0], stack[1], leaves, w1.shape, both_gw)
    elif need_gw:
        tg = _grad_targets(leaves) if both_gw else None
        _note_use(leaves[0])
        if need_gx and w1.dtype == torch.complex64:
            # both per-mode GEMMs of the pass from one launch (uno_mode_backward)
            gX, (gw1, gw2) = _native.mode_backward(xt, gO, [w1, w2], out=[tg[0][0], tg[1][0]] if tg else None, accumulate=bool(tg and tg[0][1]))
            gX = gX.view(B, Ci, 2 * m1, m2)
        else:
            gw1, gw2 = _native.mode_wgrad(xt, gO, tuple(w1.shape[:4]), 2, out=[tg[0][0], tg[1][0]] if tg else None,
                                          accumulate=bool(tg and tg[0][1]))
        if tg:
            gw1, gw2 = tg[0][2], tg[1][2]
    gx = None
    if need_gx:
        if gX is None:
            gX = _native.mode_mix(gO.view(B, Co, 2, m1 * m2), [w1, w2], 1).view(B, Ci, 2 * m1, m2)
        if merging:
            gX = join.merge(gX, (H, W))
        # addend: the (adjoint-)resampled point-wise contribution joins the transform's result before the tile is written
        gx = _native.dft2d_inverse(gX, H, W, 1.0 / (H * W), False, False, dtype=gs.dtype, addend=addend)
    return gx, gw1, gw2, gslot is not None


class GradJoin:
    """One gradient buffer for a tensor with TWO consumers (a skip connection: reference darcy_flow_uno2d.py:117-127 feeds `x_c0`
    to conv1 and, concatenated, to conv5; `x_fc0` to conv0 and to fc1) instead of two gradient tensors and an element-wise sum.

    The consumer that comes LATER in the forward pass (its backward runs first) is called with `defer_grad=join`: its backward does
    not materialise its contribution; it leaves (a) its truncated gradient spectrum for that input - the inverse transform is
    linear, two spectra on one grid are added in the (tiny) spectral domain and transformed ONCE - and (b) closures that accumulate
    its point-wise contribution into a given buffer.  The consumer that comes FIRST in the forward pass (`join=join`; its backward
    runs last - it depends on everything downstream of its output) merges the spectra into its own before the inverse transform,
    then lets the closures accumulate into its gradient buffer, and returns the complete gradient.  The deferring consumer returns
    None for that input.  Used by the harness models; without a join object every layer behaves as before."""

    def __init__(self):
        self.owner = False          # set by the first consumer's forward when it will produce the joined gradient
        self.spectra = []           # (gX (B, C, 2 m1, m2) c64, grid (H, W)) left by deferring consumers
        self.pending = []           # (callable(out, dgelu_of=None), fusable): accumulate into out (B, C, H, W); a fusable one can also
                                    # multiply the completed sum by gelu'(dgelu_of) in its epilogue
        # the joined tensor is the ACTIVATION of a block without normalisation (`out_join=` of the block that produces it): its
        # pre-activation sum, and whether the gradient handed back to that block has already been multiplied by gelu'(pre)
        self.pre = None
        self.dgelu_applied = False
        # the joined tensor is the output of the lift (lift_gelu_pad(grad_join=)): its backward kernel streams the gradient once and can add
        # a second tensor as it reads.  A deferring consumer whose contribution is a plain windowed tensor then leaves it in `extra` (no
        # accumulation pass into the owner's buffer); the lift's backward takes it.  (tensor, window) pairs.
        self.accepts_extra = False
        self.extra = []

    def reset(self):
        self.owner = False
        self.spectra, self.pending = [], []

    def take_extra(self):
        out, self.extra = self.extra, []
        return out

    def void(self):
        """A consumer or producer that was handed this join cannot honour it (it runs a stock-op path): the fused GELU derivative is
        off for this pass - the producer block applies gelu'(pre) itself to the SUM of the gradients autograd delivers, which is
        correct whatever path each consumer took."""
        self.pre = None
        self.dgelu_applied = False

    def late(self, g):
        """Gradient contribution of a consumer whose backward runs AFTER the owner's (graph order did not put it first, so it could
        not defer): when the owner has already multiplied its result by gelu'(pre) - the producer will then skip its own GELU
        backward - this contribution needs the factor as well."""
        if g is not None and self.dgelu_applied and self.pre is not None:
            g = torch.ops.aten.gelu_backward(g.contiguous(), self.pre.view(g.shape))
        return g

    def merge(self, gX, grid):
        """own gradient spectrum (B, C, 2 m1, m2) + the deferred ones, embedded by frequency into the largest mode box"""
        if not self.spectra:
            return gX
        specs = [gX] + [s for s, g in self.spectra if g == tuple(grid)]
        if len(specs) != len(self.spectra) + 1:
            raise RuntimeError("GradJoin: a deferred gradient spectrum belongs to another grid")
        M1 = max(s.shape[2] // 2 for s in specs)
        M2 = max(s.shape[3] for s in specs)
        base = next((s for s in specs[1:] if s.shape[2] // 2 == M1 and s.shape[3] == M2), None)   # a deferred copy is ours to modify
        if base is None:
            base = torch.zeros((*gX.shape[:2], 2 * M1, M2), dtype=gX.dtype, device=gX.device)
        for s in specs:
            if s is base:
                continue
            m1, m2 = s.shape[2] // 2, s.shape[3]
            base[:, :, :m1, :m2] += s[:, :, :m1]                         # frequencies 0 .. m1 - 1
            base[:, :, 2 * M1 - m1:, :m2] += s[:, :, m1:]                # frequencies -m1 .. -1
        self.spectra = []
        return base

    def apply(self, out, final_dgelu=None):
        """run the deferred accumulations; with final_dgelu (the producer block's pre-activation sum) the LAST one - if it is a
        channel-mix call - also multiplies the completed gradient by gelu'(final_dgelu).  -> True when that happened"""
        fused = False
        n = len(self.pending)
        for k, (fn, fusable) in enumerate(self.pending):
            if final_dgelu is not None and fusable and k == n - 1:
                fn(out, final_dgelu)
                fused = True
            else:
                fn(out)
        self.pending = []
        return fused


# ---- a layer on the channel concatenation of two tensors, never built: one pass over every operand where the kernels' split
# rules allow (csrc/channel_mix.hip: sources split at a multiple of 16 channels, destinations / weight-gradient tiles at 64),
# two accumulating calls otherwise
def _mix2_forward(x1, x2, w, bias, act_in=False, out=None, accumulate=False):
    """Wm . cat(x1, x2) + bias -> (B, Co, P); w (Co, C1 + C2).  out + accumulate: out += ..."""
    C1 = x1.shape[1]
    if C1 % 16 == 0:
        return _native.channel_mix2(x1, x2, w, bias, act_in=act_in, out=out, accumulate=accumulate)
    w1, w2 = w[:, :C1].contiguous(), w[:, C1:].contiguous()
    if out is None:
        out = _native.channel_mix(x1, w1, bias, act_in=act_in)
    elif accumulate:
        _native.channel_mix(x1, w1, bias, act_in=act_in, out=out)        # out= of the one-source call accumulates
    else:
        out.copy_(_native.channel_mix(x1, w1, bias, act_in=act_in))
    _native.channel_mix(x2, w2, None, out=out)
    return out


def _mix2_input_grads(gy, w, C1, dgelu_of=None, out1=None, out2=None):
    """(W[:, :C1]^T gy [* gelu'(dgelu_of)], W[:, C1:]^T gy) from one read of gy; out1 / out2: accumulate into these."""
    if C1 % 64 == 0 and (w.shape[1] - C1) >= 1:
        if out1 is not None and out2 is not None:
            _native.channel_mix2(gy, None, w, None, transpose_w=True, out=out1, out2=out2, split_out=C1, dgelu_of=dgelu_of, accumulate=True)
            return out1, out2
        if out1 is None and out2 is None:
            return _native.channel_mix2(gy, None, w, None, transpose_w=True, split_out=C1, dgelu_of=dgelu_of)
    w1, w2 = w[:, :C1].contiguous(), w[:, C1:].contiguous()
    g1 = _native.channel_mix(gy, w1, None, transpose_w=True, dgelu_of=dgelu_of, out=out1)
    g2 = _native.channel_mix(gy, w2, None, transpose_w=True, out=out2)
    return g1, g2


def _mix2_wgrad(gy, x1, x2, need_bias, act_x=False):
    """gw (Co, C1 + C2), gb of a two-source layer."""
    if x1.shape[1] % 64 == 0 and gy.shape[2] >= 64:
        return _native.channel_wgrad2(gy, x1, x2, need_bias=need_bias, act_x=act_x)
    gw1, gb = _native.channel_wgrad(gy, x1, need_bias=need_bias, act_x=act_x)
    gw2, _ = _native.channel_wgrad(gy, x2, need_bias=False)
    return torch.cat([gw1, gw2], dim=1), gb


class _ChannelMixCatFn(torch.autograd.Function):
    """y[b] = W . cat(a1[b], x2[b]) + bias without the concatenation: W[:, :C1] . a1 writes y, W[:, C1:] . x2
    accumulates into it; the input gradients come out as two contiguous tensors (no strided slices of a joint one).
    gelu_first: a1 = gelu(x1) with x1 kept pre-activation - the GELU is applied as K8 / K9 read x1, and the input-gradient
    call returns the gradient of x1 itself (its epilogue multiplies by gelu'(x1)): the activation tensor never exists."""

    @staticmethod
    def forward(ctx, x1, x2, w, bias, gelu_first, defer=None, grid=None, leaves=None):
        ctx.leaves = leaves
        x1, x2, w = _plain(x1), _plain(x2), _plain(w)
        y = _mix2_forward(x1, x2, w, None if bias is None else _plain(bias), act_in=gelu_first)
        ctx.save_for_backward(x1, x2, w)
        ctx.has_bias = bias is not None
        ctx.gelu_first = gelu_first
        ctx.defer = defer if (defer is not None and defer.owner and ctx.needs_input_grad[1]) else None
        ctx.grid = grid
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x1, x2, w = ctx.saved_tensors[:3]
        return _ChannelMixCatFn._backward(ctx, x1, x2, w, _plain(gy)) + (None, None, None, None)

    @staticmethod
    def _backward(ctx, x1, x2, w, gy):
        """(g1, g2, gw, gb) of y = W . cat([gelu](x1), x2) + b for the output gradient gy (shared with the fused-projection form).
        ctx.window (fused-projection form): gy is valid on that window of its planes only; the gradients come out as whole planes
        with a cleared border."""
        window = getattr(ctx, "window", None)
        if window is not None:
            return _ChannelMixCatFn._backward_window(ctx, x1, x2, w, gy, window)
        C1 = x1.shape[1]
        g1 = g2 = None
        if ctx.defer is not None and ctx.defer.owner and ctx.needs_input_grad[1]:     # owner still pending: its backward has not run yet
            # x2's gradient is accumulated later into the buffer of x2's other consumer (GradJoin): no tensor, no sum
            if ctx.needs_input_grad[0]:
                g1 = _native.channel_mix(gy, w[:, :C1].contiguous(), None, transpose_w=True, dgelu_of=x1 if ctx.gelu_first else None)
            w2 = w[:, C1:].contiguous()
            B, C2 = x2.shape[0], x2.shape[1]
            ctx.defer.pending.append((lambda out, dg=None: _native.channel_mix(
                gy, w2, None, transpose_w=True, out=out.view(B, C2, -1), dgelu_of=None if dg is None else dg.view(B, C2, -1),
                dgelu_total=dg is not None), True))
        elif ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            g1, g2 = _mix2_input_grads(gy, w, C1, dgelu_of=x1 if ctx.gelu_first else None)
        elif ctx.needs_input_grad[0]:
            g1 = _native.channel_mix(gy, w[:, :C1].contiguous(), None, transpose_w=True, dgelu_of=x1 if ctx.gelu_first else None)
        elif ctx.needs_input_grad[1]:
            g2 = _native.channel_mix(gy, w[:, C1:].contiguous(), None, transpose_w=True)
        if ctx.defer is not None and g2 is not None:        # the owner's backward came first after all
            g2 = ctx.defer.late(g2)
        gw, gb = _wgrad_into(ctx.leaves, gy, x1, x2, ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3], act_x=ctx.gelu_first)
        return g1, g2, gw, gb

    @staticmethod
    def _backward_window(ctx, x1, x2, w, gy, window):
        rows, cols, pitch = window
        C1 = x1.shape[1]
        B, C2 = x2.shape[0], x2.shape[1]
        dg = x1 if ctx.gelu_first else None

        def cleared(g):
            # what the windowed kernel did not write: the columns right of the window, the rows below it (a gradient's consumers -
            # the transforms and resampling of the producing block - read whole planes)
            _native.clear_border(g.view(g.shape[0], g.shape[1], -1, pitch), rows, cols)
            return g

        g1 = g2 = None
        deferred = ctx.defer is not None and ctx.defer.owner and ctx.needs_input_grad[1]
        if deferred and ctx.defer.accepts_extra and ctx.needs_input_grad[0] and C1 % 64 == 0 and not ctx.defer.extra:
            # x2 is the lift's output: both input gradients from ONE pass over gy (two destinations); x2's stays a tensor of its own,
            # valid on the window, that the lift's backward kernel adds to the owner's gradient as it reads the two (no border to clear:
            # that kernel reads the domain only)
            g1, g2w = _native.channel_mix2(gy, None, w, None, transpose_w=True, split_out=C1, dgelu_of=dg, window=window)
            cleared(g1)
            ctx.defer.extra.append((g2w, window))
        elif deferred:
            if ctx.needs_input_grad[0]:
                g1 = cleared(_native.channel_mix(gy, w[:, :C1].contiguous(), None, transpose_w=True, dgelu_of=dg, window=window))
            w2 = w[:, C1:].contiguous()
            # accumulates into the window of the other consumer's (whole-plane) gradient: nothing to clear.  NOT fusable: a fused
            # gelu'(pre) epilogue would reach the window only, and the border of the joined gradient holds the owner's own non-zero
            # contribution - the owner applies gelu' to the whole plane itself (GradJoin.apply reports "not fused")
            ctx.defer.pending.append((lambda out, dgo=None: _native.channel_mix(
                gy, w2, None, transpose_w=True, out=out.view(B, C2, -1), window=window), False))
        else:
            if ctx.needs_input_grad[0]:
                g1 = cleared(_native.channel_mix(gy, w[:, :C1].contiguous(), None, transpose_w=True, dgelu_of=dg, window=window))
            if ctx.needs_input_grad[1]:
                g2 = cleared(_native.channel_mix(gy, w[:, C1:].contiguous(), None, transpose_w=True, window=window))
                if ctx.defer is not None:        # the owner's backward came first after all
                    g2 = ctx.defer.late(g2)
        gw, gb = _wgrad_into(ctx.leaves, gy, x1, x2, ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3], act_x=ctx.gelu_first,
                             window=window)
        return g1, g2, gw, gb


class _ChannelMixCatProjectFn(torch.autograd.Function):
    """out[b, p] = b2 + sum_o w2[o] gelu(y[b, o, p]),  y = W . cat([gelu](x1), x2) + b: the end of the models, `fc2(F.gelu(fc1(cat)))`
    with one output channel (reference darcy_flow_uno2d.py:122-131), in ONE pass - the channel-mix kernel that produces y (kept:
    its GELU derivative is needed backward) also reduces its 64-channel tile to the projected value, so y is not read again."""

    @staticmethod
    def forward(ctx, x1, x2, w, bias, w2, b2, gelu_first, defer=None, leaves=None, window=None):
        ctx.leaves = leaves
        ctx.window = window
        x1, x2, w, w2 = _plain(x1), _plain(x2), _plain(w), _plain(w2)
        y, out = _native.channel_mix2(x1, x2, w, None if bias is None else _plain(bias), act_in=gelu_first,
                                      project=(w2, None if b2 is None else _plain(b2)), window=window)
        ctx.save_for_backward(x1, x2, w, y, w2)
        ctx.has_bias, ctx.has_b2 = bias is not None, b2 is not None
        ctx.gelu_first = gelu_first
        ctx.defer = defer if (defer is not None and defer.owner and ctx.needs_input_grad[1]) else None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x1, x2, w, y, w2 = ctx.saved_tensors
        mode = _ChannelMixCatProjectFn._fused_mode(ctx, x1, x2, y)
        if mode:
            return _ChannelMixCatProjectFn._backward_fused(ctx, x1, x2, w, y, w2, _plain(gout), mode)
        gy, gw2, gb2 = _native.gelu_project_backward(y, w2, _plain(gout), need_bias=ctx.has_b2, window=ctx.window)
        g1, g2, gw, gb = _ChannelMixCatFn._backward(ctx, x1, x2, w, gy)
        return g1, g2, gw, gb, gw2, gb2, None, None, None, None

    @staticmethod
    def _fused_mode(ctx, x1, x2, y):
        """0: the three-call backward pass; 1: uno_project_backward, both input gradients returned; 2: the same with x2's gradient handed
        to the owner of the joined gradient as a second tensor (the lift's backward kernel adds the two as it reads them)."""
        if not PROJECT_BACKWARD_FUSED or x1.dtype != torch.float32 or not all(ctx.needs_input_grad[:3]) or not ctx.needs_input_grad[4]:
            return 0
        B, C1, P = x1.shape
        if not _native.project_backward_applies(B, C1, C1 + x2.shape[1], y.shape[1], P, ctx.window):
            return 0
        if ctx.defer is None:
            return 1
        if ctx.window is not None and ctx.defer.accepts_extra and not ctx.defer.extra:
            return 2
        return 0

    @staticmethod
    def _backward_fused(ctx, x1, x2, w, y, w2, gout, mode):
        window = ctx.window
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        Co, Ci = y.shape[1], x1.shape[1] + x2.shape[1]
        tg = None
        if ctx.leaves is not None and (ctx.leaves[1] is not None) == need_b:
            tg = _grad_targets([ctx.leaves[0]] + ([ctx.leaves[1]] if need_b else []))        # committed: the call below writes them
        g1, g2, gw, gb, gw2, gb2 = _native.project_backward(
            x1, x2, w, y, w2, gout, act_in=ctx.gelu_first, need_bias=need_b, need_bias2=ctx.has_b2 and ctx.needs_input_grad[5], window=window,
            out_w=None if tg is None else tg[0][0], out_b=None if tg is None or not need_b else tg[1][0],
            accumulate=False if tg is None else tg[0][1])
        if tg is not None:
            gw = None if tg[0][2] is None else tg[0][2].view(Co, Ci)
            gb = tg[1][2] if need_b else None
        if window is not None:
            rows, cols, pitch = window
            _native.clear_border(g1.view(g1.shape[0], g1.shape[1], -1, pitch), rows, cols)
            if mode == 2:
                ctx.defer.extra.append((g2, window))        # (no border to clear: the lift's backward kernel reads the domain only)
                g2 = None
            else:
                _native.clear_border(g2.view(g2.shape[0], g2.shape[1], -1, pitch), rows, cols)
        return g1, g2, gw, gb, gw2, gb2, None, None, None, None


def channel_mix_cat_project(xs, weight, bias, weight2, bias2, gelu_first: bool = False, defer_grad=None, crop=None):
    """gelu_project(channel_mix_cat(xs, weight, bias, gelu_first), weight2, bias2) - `fc2(F.gelu(fc1(torch.cat(xs, 1))))` of the
    models - as one forward kernel where the shapes allow (two device tensors split at a multiple of 16 channels, at most 64
    channels between the two layers, ONE output channel).
    crop = (S1, S2): the caller keeps only out[..., :S1, :S2] (the reference removes the domain padding BEFORE these layers,
    darcy_flow_uno2d.py:125): the kernels then work on that window of the padded tensors - forward and backward - and the rest of
    the returned (B, 1, H, W) tensor is undefined.  Ignored where the windowed kernels do not apply."""
    Co = weight.shape[0]
    if (len(xs) == 2 and all(_dev_act(x) for x in xs) and xs[0].dtype == xs[1].dtype and weight.dtype == torch.float32
            and weight2.shape[0] == 1 and Co <= 64 and xs[0].shape[1] % 16 == 0 and weight2.dtype == torch.float32):
        x1, x2 = xs
        B = x1.shape[0]
        w = weight.reshape(Co, -1)
        window = None
        if crop is not None and x1.dim() == 4 and x1.dtype == torch.float32 and x1.shape[1] % 64 == 0:
            H, W = x1.shape[2:]
            rows, cols = int(crop[0]), (int(crop[1]) + 3) & ~3
            if 0 < rows <= H and 260 <= cols <= W and rows * cols < (1 << 24) and (rows < H or cols < W) and tuple(x2.shape[2:]) == (H, W):
                window = (rows, cols, W)
        out = _ChannelMixCatProjectFn.apply(x1.reshape(B, x1.shape[1], -1), x2.reshape(B, x2.shape[1], -1), w, bias,
                                            weight2.reshape(Co), bias2, bool(gelu_first), defer_grad, (weight, bias), window)
        return out.view(B, 1, *x1.shape[2:])
    return gelu_project(channel_mix_cat(xs, weight, bias, gelu_first=gelu_first, defer_grad=defer_grad), weight2, bias2)


def channel_mix_cat(xs, weight: torch.Tensor, bias: torch.Tensor | None, gelu_first: bool = False, defer_grad=None) -> torch.Tensor:
    """channel_mix(torch.cat(xs, dim=1), weight, bias) - the projection after a skip connection (reference
    darcy_flow_uno2d.py:122-127: `torch.cat([x_c5, x_fc0], dim=1)` then `fc1`) - without materialising the
    concatenation when there are two float32 device tensors.  gelu_first: xs[0] is a PRE-activation tensor and stands
    for gelu(xs[0]) (the block in front deferred its GELU to this consumer).  defer_grad: a GradJoin whose owner is xs[1]'s other
    consumer - xs[1]'s gradient is then accumulated into that consumer's buffer (fused device path only)."""
    if len(xs) == 2 and all(_dev_act(x) for x in xs) and xs[0].dtype == xs[1].dtype and weight.dtype == torch.float32:
        x1, x2 = xs
        B = x1.shape[0]
        w = weight.reshape(weight.shape[0], -1)
        y = _ChannelMixCatFn.apply(x1.reshape(B, x1.shape[1], -1), x2.reshape(B, x2.shape[1], -1), w, bias, bool(gelu_first),
                                   defer_grad, tuple(x2.shape[2:]), (weight, bias))
        return y.view(B, w.shape[0], *x1.shape[2:])
    xs = list(xs)
    if defer_grad is not None:
        defer_grad.void()
    if gelu_first:
        xs[0] = F.gelu(xs[0])
    return channel_mix(torch.cat(xs, dim=1), weight, bias)


class _GeluChannelMixFn(torch.autograd.Function):
    """y[b] = W . gelu(pre[b]) + bias with `pre` kept pre-activation (the lift `fc0(F.gelu(fc_n1(x)))`, reference
    darcy_flow_uno2d.py:98-101): GELU on read in K8 / K9, gelu'(pre) in the input-gradient epilogue."""

    @staticmethod
    def forward(ctx, pre, w, bias, leaves=None):
        pre, w = _plain(pre), _plain(w)
        y = _native.channel_mix(pre, w, None if bias is None else _plain(bias), act_in=True)
        ctx.save_for_backward(pre, w)
        ctx.has_bias = bias is not None
        ctx.leaves = leaves
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        pre, w = ctx.saved_tensors
        gy = _plain(gy)
        g_pre = _native.channel_mix(gy, w, None, transpose_w=True, dgelu_of=pre) if ctx.needs_input_grad[0] else None
        gw, gb = _wgrad_into(ctx.leaves, gy, pre, None, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], act_x=True)
        return g_pre, gw, gb, None


def gelu_channel_mix(pre: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """channel_mix(F.gelu(pre), weight, bias) without the activation tensor (float32 device tensors)."""
    B, Ci = pre.shape[0], pre.shape[1]
    w = weight.reshape(weight.shape[0], Ci)
    if _dev_act(pre) and w.dtype == torch.float32:
        y = _GeluChannelMixFn.apply(pre.reshape(B, Ci, -1), w, bias, (weight, bias))
        return y.view(B, w.shape[0], *pre.shape[2:])
    return channel_mix(F.gelu(pre), weight, bias)


class _GeluChannelMixPadFn(torch.autograd.Function):
    """zero-pad(gelu(W . gelu(pre) + bias)): the second lift layer, its activation and the domain padding (reference
    darcy_flow_uno2d.py:100-107) from ONE forward kernel - the layer's store epilogue writes the padded activation and nothing else
    (uno_channel_mix_act_padded without y).  The pre-activation result is not kept: the backward pass RECOMPUTES it from the layer's
    input (32 channels against the 64 it would store and re-read) inside the kernel that multiplies gelu' into the cropped
    gradient (uno_channel_mix_dgelu_padded), then runs the layer's two gradient kernels as in _GeluChannelMixFn."""

    @staticmethod
    def forward(ctx, pre, w, bias, Hp, Wp, leaves=None):
        pre, w = _plain(pre), _plain(w)
        bias = None if bias is None else _plain(bias)
        _, act = _native.channel_mix_act_padded(pre, w, bias, Hp, Wp, act_in=True, keep_y=False)
        ctx.save_for_backward(pre, w, bias)
        ctx.has_bias = bias is not None
        ctx.leaves = leaves
        return act

    @staticmethod
    @once_differentiable
    def backward(ctx, gact):
        pre, w, bias = ctx.saved_tensors
        B, Ci, Co = pre.shape[0], pre.shape[1], w.shape[0]
        gz = _native.channel_mix_dgelu_padded(pre, w, bias, _plain(gact), act_in=True).view(B, Co, -1)
        pre3 = pre.view(B, Ci, -1)
        g_pre = _native.channel_mix(gz, w, None, transpose_w=True, dgelu_of=pre3).view(pre.shape) if ctx.needs_input_grad[0] else None
        gw, gb = _wgrad_into(ctx.leaves, gz, pre3, None, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], act_x=True)
        return g_pre, gw, gb, None, None, None


def gelu_channel_mix_pad(pre: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, pad_h: int, pad_w: int) -> torch.Tensor:
    """gelu_pad2d(gelu_channel_mix(pre, weight, bias), pad_h, pad_w) - `F.pad(F.gelu(fc0(F.gelu(pre))), [0, pad_w, 0, pad_h])` - with the
    activation and the padding written by fc0's own kernel where the shapes allow (4-D float32 device tensor, width >= 260)."""
    if pre.dim() == 4 and _dev_act(pre) and weight.dtype == torch.float32 and pad_h >= 0 and pad_w >= 0:
        Hp, Wp = pre.shape[2] + int(pad_h), pre.shape[3] + int(pad_w)
        if _native.channel_mix_act_padded_ok(pre, Hp, Wp):
            return _GeluChannelMixPadFn.apply(pre, weight.reshape(weight.shape[0], pre.shape[1]), bias, Hp, Wp, (weight, bias))
    return gelu_pad2d(gelu_channel_mix(pre, weight, bias), pad_h, pad_w)


class _LiftFn(torch.autograd.Function):
    """The whole lift - zero-pad(gelu(fc0(gelu(fc_n1(x))))), reference darcy_flow_uno2d.py:98-107 - with neither layer's output stored
    (uno_lift_forward / uno_lift_backward): the first layer has 3 input channels, so every kernel that needs its 32-channel result
    evaluates it from x.  x is data: no gradient for it."""

    @staticmethod
    def forward(ctx, x, w1, b1, w0, b0, Hp, Wp, grad_join=None):
        x, w1, w0 = _plain(x), _plain(w1), _plain(w0)
        b1 = None if b1 is None else _plain(b1)
        b0 = None if b0 is None else _plain(b0)
        ctx.save_for_backward(x, w1, w0, *[t for t in (b1, b0) if t is not None])
        ctx.has = (b1 is not None, b0 is not None)
        ctx.join = None
        if grad_join is not None:
            grad_join.accepts_extra = bool(_native.lift_backward_takes_second(x, w1, w0, Hp, Wp))
            grad_join.extra = []
            ctx.join = grad_join
        return _native.lift_forward(x, w1, b1, w0, b0, Hp, Wp)

    @staticmethod
    @once_differentiable
    def backward(ctx, gact):
        x, w1, w0, *bs = ctx.saved_tensors
        b1 = bs.pop(0) if ctx.has[0] else None
        b0 = bs.pop(0) if ctx.has[1] else None
        gact = _plain(gact)
        g2 = None
        if ctx.join is not None:
            ctx.join.accepts_extra = False
            H, W = x.shape[-2:]
            for t, (rows, cols, pitch) in ctx.join.take_extra():
                t = t.view(gact.shape)
                if g2 is None and rows >= H and cols >= W and pitch == gact.shape[-1]:
                    g2 = t                  # covers the domain on the same planes: the kernel adds it as it reads
                else:                       # (not reached by the harness models) any other extra: a windowed element-wise sum
                    gact = gact.clone()
                    gact[..., :rows, :cols] += t[..., :rows, :cols]
        gw1, gb1, gw0, gb0 = _native.lift_backward(x, w1, b1, w0, b0, gact, g2)
        return None, gw1, gb1, gw0, gb0, None, None, None


def lift_gelu_pad(x: torch.Tensor, fc_n1: nn.Module, fc0: nn.Module, pad_h: int, pad_w: int, grad_join=None) -> torch.Tensor:
    """F.pad(F.gelu(fc0(F.gelu(fc_n1(x)))), [0, pad_w, 0, pad_h]) for a channels-first x (B, Cin, H, W) and two nn.Linear layers, as one
    forward kernel and one backward kernel that store neither intermediate nor their gradients, where the shapes allow (at most 3 input channels, 16 or 32
    in the middle, width >= 260, float32, x without gradient); the layer-by-layer forms otherwise.
    grad_join: the GradJoin of the RESULT (it feeds two layers: reference darcy_flow_uno2d.py:108, :127) - where the one-kernel backward
    runs, a deferring consumer may leave its contribution as a tensor of its own (GradJoin.extra) and that kernel adds it while reading."""
    w1, w0 = fc_n1.weight, fc0.weight
    if x.dim() == 4 and _dev_act(x) and not x.requires_grad and w1.dtype == torch.float32 and w0.dtype == torch.float32 and pad_h >= 0 and pad_w >= 0:
        Hp, Wp = x.shape[2] + int(pad_h), x.shape[3] + int(pad_w)
        if _native.lift_ok(x, w1, w0, Hp, Wp):
            return _LiftFn.apply(x, w1, fc_n1.bias, w0, fc0.bias, Hp, Wp, grad_join)
    if grad_join is not None:
        grad_join.accepts_extra = False
    return gelu_channel_mix_pad(channel_mix(x, w1, fc_n1.bias), w0, fc0.bias, pad_h, pad_w)


class _GeluProjectFn(torch.autograd.Function):
    """out[b, p] = bias + sum_c w[c] gelu(pre[b, c, p]) (K11, csrc/pointwise_fused.hip)."""

    @staticmethod
    def forward(ctx, pre, w, bias):
        pre, w = _plain(pre), _plain(w)
        out = _native.gelu_project_forward(pre, w, None if bias is None else _plain(bias))
        ctx.save_for_backward(pre, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        pre, w = ctx.saved_tensors
        gpre, gw, gb = _native.gelu_project_backward(pre, w, _plain(gout), need_bias=ctx.has_bias)
        return gpre, gw, gb


def gelu_project(pre: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """channel_mix(F.gelu(pre), weight, bias) for the models' final projection (reference darcy_flow_uno2d.py:128-131:
    `F.gelu(self.fc1(x))` then `self.fc2`, fc2 = Linear(C, 1)): with ONE output channel on a HIP device the GELU and
    the projection are a single streaming pass (no GELU output tensor, no one-row GEMM)."""
    if weight.shape[0] == 1 and _dev_act(pre) and weight.dtype == torch.float32 and pre.shape[1] <= 1024:
        B, C = pre.shape[0], pre.shape[1]
        out = _GeluProjectFn.apply(pre.reshape(B, C, -1), weight.reshape(C), bias)
        return out.view(B, 1, *pre.shape[2:])
    return channel_mix(F.gelu(pre), weight, bias)


class _GeluPadFn(torch.autograd.Function):
    """zero-pad(gelu(s)) at the end of the last two axes (K12)."""

    @staticmethod
    def forward(ctx, s, Hp, Wp):
        s = _plain(s)
        ctx.save_for_backward(s)
        return _native.gelu_pad(s, Hp, Wp)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (s,) = ctx.saved_tensors
        return _native.gelu_pad_backward(s, _plain(gy)), None, None


def gelu_pad2d(s: torch.Tensor, pad_h: int, pad_w: int) -> torch.Tensor:
    """F.pad(F.gelu(s), [0, pad_w, 0, pad_h]) - the lift's last activation and the domain padding (reference
    darcy_flow_uno2d.py:103-107) in one pass over the tensor on a HIP device."""
    if _dev_act(s) and s.dim() >= 2 and pad_h >= 0 and pad_w >= 0:
        return _GeluPadFn.apply(s, s.shape[-2] + int(pad_h), s.shape[-1] + int(pad_w))
    return F.pad(F.gelu(s), [0, pad_w, 0, pad_h])


class _InstanceNormGeluFn(torch.autograd.Function):
    """[gelu](InstanceNorm(x) * weight + bias) with K13 (csrc/instnorm.hip); saves x and the per-row mean / rstd."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, gelu):
        x = _plain(x)
        w = None if weight is None else _plain(weight)
        b = None if bias is None else _plain(bias)
        y, mean, rstd = _native.instnorm_forward(x, w, b, eps, gelu)
        ctx.save_for_backward(x, w, b, mean, rstd)
        ctx.gelu = gelu
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, b, mean, rstd = ctx.saved_tensors
        gx, s1, s2 = _native.instnorm_backward(x, _plain(gy), w, b, mean, rstd, ctx.gelu)
        gw = s2.sum(0) if w is not None and ctx.needs_input_grad[1] else None
        gb = s1.sum(0) if b is not None and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None


def instance_norm_gelu(x: torch.Tensor, norm: nn.Module, gelu: bool) -> torch.Tensor:
    """`norm(x)` followed, if `gelu`, by F.gelu - for an nn.InstanceNorm{2,3}d without running statistics on a float32
    HIP tensor both run as one kernel; anything else takes the stock modules."""
    if (_dev_act(x) and isinstance(norm, (nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d))
            and not norm.track_running_stats and x.dim() >= 3 and x.shape[1] == norm.num_features):
        if x.numel() // max(x.shape[0] * x.shape[1], 1) <= 1:           # torch.nn.functional.instance_norm refuses this too
            raise ValueError(f"Expected more than 1 spatial element when training, got input size {x.size()}")
        return _InstanceNormGeluFn.apply(x, norm.weight, norm.bias, norm.eps, bool(gelu))
    out = norm(x)
    return F.gelu(out) if gelu else out


class _OperatorBlock2dFn(torch.autograd.Function):
    """s = SpectralConv2d_Uno(x) + pointwise_op_2D(x) in ONE buffer (reference integral_operators.py:270-273:
    `x1_out = self.conv(x, ...); x2_out = self.w(x, ...); x_out = x1_out + x2_out`).

    The spectral branch's inverse DFT writes s; the last kernel of the point-wise branch (the channel mix when the
    block does not up-sample, the resampling otherwise) accumulates into it.  In the backward pass the spectral
    branch writes grad_x and the point-wise branch's last kernel accumulates into that.  Neither sum exists as a
    separate element-wise pass."""

    @staticmethod
    def forward(ctx, x, w1, w2, cw, cb, Ho, Wo, half_weights=False, fuse_gelu=False, join=None, out_join=None):
        """fuse_gelu (blocks with Non_Lin and no normalisation, reference integral_operators.py:282-283): returns gelu(s); where the
        channel mix is the kernel that completes s (no up-sampling) it writes the activation in the same pass.
        join: GradJoin of x - this block is x's FIRST consumer and returns x's complete gradient (see GradJoin).
        out_join (with fuse_gelu): the GradJoin of this block's OUTPUT; the block leaves its pre-activation sum there, and the
        consumer that completes the output's gradient multiplies it by gelu'(pre) in its last accumulating kernel - this block's
        backward then receives the gradient at the pre-activation sum and runs no GELU-backward pass."""
        from .resample import resample_forward
        ctx.leaves = (w1, w2, cw, cb)
        ctx.join = None
        if join is not None:
            join.reset()
            if ctx.needs_input_grad[0]:
                join.owner = True
                ctx.join = join
        x = _plain(x)
        B, Ci, H, W = x.shape
        ctx.stack = _stack_take(w1, (B, Ci, 2 * w1.shape[2], w1.shape[3]), x.device, _stack_wanted(ctx, 1, x, half_weights))
        w1, w2 = _plain(w1), _plain(w2)
        if half_weights:
            w1, w2 = _half_weights(w1, w2)
        Co = cw.shape[0]
        cwm = _plain(cw).reshape(Co, Ci)
        cb = None if cb is None else _plain(cb)
        same = (H, W) == (Ho, Wo)
        mix_last = same or Ho * Wo < H * W          # the 1x1 convolution runs on whichever side has fewer pixels
        t = fused = None
        if not mix_last and not half_weights and x.dtype == torch.float32:
            # up-sampling block: the 1x1 convolution first, its result joins the inverse transform's (one pass over the output)
            t = _native.channel_mix(x.view(B, Ci, -1), cwm, cb).view(B, Co, H, W)
            fused = _fused_addend(t, Ho, Wo, w1.shape[2], w1.shape[3], False)
        if fused is not None:
            m1, m2 = w1.shape[2], w1.shape[3]
            xt = torch.empty((B, Ci, 2 * m1, m2), dtype=torch.complex64, device=x.device) if ctx.stack is None else ctx.stack[0].X[ctx.stack[1]]
            _native.dft2d_forward(x, m1, m2, 1.0 / (H * W), out=xt, channel_offset=0)
            O = _native.mode_mix(xt.view(B, Ci, 2, m1 * m2), [w1, w2], 0)
            s = _native.dft2d_inverse(O.view(B, Co, 2 * m1, m2), Ho, Wo, 1.0, True, True, addend=fused)
        else:
            pre_act = None
            if mix_last and not same and REVERSE_SWEEP_RESAMPLE:
                pre_act = resample_forward(x, Ho, Wo, reverse=True)         # right before K1 reads x, in the opposite image order
            s, xt = _native.spectral_conv2d_forward(x, w1, w2, Ho, Wo, xt_out=None if ctx.stack is None else ctx.stack[0].X[ctx.stack[1]])
        out = s
        if fused is not None:
            act = x
            if fuse_gelu:
                out = F.gelu(s)
        elif mix_last:
            act = x if same else (pre_act if pre_act is not None else resample_forward(x, Ho, Wo))
            if fuse_gelu:
                _, out = _native.channel_mix2(act.view(B, Ci, -1), None, cwm, cb, out=s.view(B, Co, -1), accumulate=True, y_act=True)
                out = out.view(B, Co, Ho, Wo)
            else:
                _native.channel_mix(act.view(B, Ci, -1), cwm, cb, out=s.view(B, Co, -1))
        else:
            act = x
            if t is None:
                t = _native.channel_mix(x.view(B, Ci, -1), cwm, cb).view(B, Co, H, W)
            resample_forward(t, Ho, Wo, out=s)
            if fuse_gelu:
                out = F.gelu(s)
        ctx.save_for_backward(xt, w1, w2, cwm, act, s if fuse_gelu else None)
        ctx.geom = (H, W, same, mix_last, cb is not None, tuple(cw.shape))
        ctx.out_join = None
        if fuse_gelu and out_join is not None:
            out_join.pre, out_join.dgelu_applied = s, False
            ctx.out_join = out_join
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gs):
        from .resample import resample_adjoint
        xt, w1, w2, cwm, act, pre = ctx.saved_tensors
        H, W, same, mix_last, has_bias, cw_shape = ctx.geom
        gs = _plain(gs)
        if pre is not None:                 # the block's GELU: gradient at the pre-activation sum
            oj = ctx.out_join
            if oj is not None and oj.dgelu_applied and oj.pre is not None and oj.pre.data_ptr() == pre.data_ptr():
                oj.dgelu_applied = False    # the consumer's last kernel already multiplied by gelu'(pre)
            else:
                gs = torch.ops.aten.gelu_backward(gs, pre)
            if oj is not None:
                oj.pre = None
        B, Co, Ho, Wo = gs.shape
        Ci = cwm.shape[1]
        need_gx = ctx.needs_input_grad[0]
        need_gw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need_gc = ctx.needs_input_grad[3] or (has_bias and ctx.needs_input_grad[4])
        join = ctx.join
        lw1, lw2, lcw, lcb = ctx.leaves
        # down-sampling block (forward: act = R x; s += Wm act): the point-wise part of gx is the ADJOINT resampling of Wm^T gs, an
        # up-sampling - it joins the spectral part inside the inverse transform (one pass over gx) where the fused kernel applies
        g_act = addend = g_t = None
        if not mix_last and REVERSE_SWEEP_RESAMPLE:
            g_t = resample_adjoint(gs, H, W, reverse=True).view(B, Co, -1)          # right before K1 reads gs, in the opposite image order
        if mix_last and not same and need_gx and gs.dtype == torch.float32:
            g_act = _native.channel_mix(gs.view(B, Co, -1), cwm, None, transpose_w=True).view(B, Ci, Ho, Wo)
            addend = _fused_addend(g_act, H, W, w1.shape[2], w1.shape[3], True)
        gx, gw1, gw2, stacked = _spectral_backward(gs, xt, w1, w2, H, W, need_gx, need_gw, ctx.needs_input_grad[1] and ctx.needs_input_grad[2],
                                                   (lw1, lw2), ctx.stack, join, addend)
        pstack = ctx.stack if stacked else None         # the 1x1 convolution's weight gradient follows the spectral layer's stack
        gcw = gcb = None
        # x is the activation of a fused-GELU block (join.pre): the gradient this block returns must be multiplied by gelu'(pre).
        # The LAST kernel that accumulates into gx does it - a deferred closure if any is pending, else this block's own
        # transposed channel mix where that comes last; otherwise a separate pass at the end
        xpre = join.pre if (join is not None and need_gx) else None
        own_last = xpre is not None and not join.pending
        dg_view = xpre.view(B, Ci, -1) if own_last else None
        dg_done = False
        if mix_last:
            # forward: act = R x;  s += Wm act + b
            if need_gx:
                if same:
                    _native.channel_mix(gs.view(B, Co, -1), cwm, None, transpose_w=True, out=gx.view(B, Ci, -1), dgelu_of=dg_view,
                                        dgelu_total=own_last)
                    dg_done = own_last
                elif addend is None:
                    if g_act is None:
                        g_act = _native.channel_mix(gs.view(B, Co, -1), cwm, None, transpose_w=True).view(B, Ci, Ho, Wo)
                    resample_adjoint(g_act, H, W, out=gx)
            if need_gc:
                gcw, gcb = _wgrad_into((lcw, lcb), gs.view(B, Co, -1), act.view(B, Ci, -1), None, ctx.needs_input_grad[3],
                                       has_bias and ctx.needs_input_grad[4], stack=pstack)
        else:
            # forward: t = Wm x + b;  s += R t
            if g_t is None:
                g_t = resample_adjoint(gs, H, W).view(B, Co, -1)
            if need_gx:
                _native.channel_mix(g_t, cwm, None, transpose_w=True, out=gx.view(B, Ci, -1), dgelu_of=dg_view, dgelu_total=own_last)
                dg_done = own_last
            if need_gc:
                gcw, gcb = _wgrad_into((lcw, lcb), g_t, act.view(B, Ci, -1), None, ctx.needs_input_grad[3],
                                       has_bias and ctx.needs_input_grad[4], stack=pstack)
        if gcw is not None:
            gcw = gcw.view(cw_shape)
        if join is not None:
            if need_gx:
                # the point-wise contributions of x's other consumer accumulate into this buffer (the last one applies gelu'(pre))
                dg_done = join.apply(gx, xpre if not dg_done else None) or dg_done
                if xpre is not None:
                    if not dg_done:
                        gx = torch.ops.aten.gelu_backward(gx, xpre)
                    join.dgelu_applied = True
            join.reset()
        return gx, gw1, gw2, gcw, gcb, None, None, None, None, None, None


class _OperatorBlock2dCatFn(torch.autograd.Function):
    """_OperatorBlock2dFn for an input that the reference builds with torch.cat([x1, x2], dim=1) (skip connections,
    reference darcy_flow_uno2d.py:117-125), without building it: K1 transforms the two sources into the channel ranges
    of one truncated spectrum, the point-wise branch mixes the two sources with the two column blocks of the 1x1
    weight, and the backward pass returns the two input gradients as separate contiguous tensors (no strided slices
    of a joint gradient to copy or accumulate)."""

    @staticmethod
    def forward(ctx, x1, x2, w1, w2, cw, cb, Ho, Wo, half_weights=False, defer=None):
        from .resample import resample_forward
        ctx.leaves = (w1, w2, cw, cb)
        ctx.defer = defer if (defer is not None and defer.owner and ctx.needs_input_grad[1]) else None
        x1, x2 = _plain(x1), _plain(x2)
        B, C1, H, W = x1.shape
        C2 = x2.shape[1]
        Ci, Co, m1, m2 = w1.shape
        ctx.stack = _stack_take(w1, (B, Ci, 2 * m1, m2), x1.device, _stack_wanted(ctx, 2, x1, half_weights))
        w1, w2 = _plain(w1), _plain(w2)
        if half_weights:
            w1, w2 = _half_weights(w1, w2)
        cwm = _plain(cw).reshape(Co, Ci)
        cb = None if cb is None else _plain(cb)
        # spectral branch, stage by stage (the composite entry point takes a single source)
        xt = torch.empty((B, Ci, 2 * m1, m2), dtype=torch.complex64, device=x1.device) if ctx.stack is None else ctx.stack[0].X[ctx.stack[1]]
        _native.dft2d_forward(x1, m1, m2, 1.0 / (H * W), out=xt, channel_offset=0)
        _native.dft2d_forward(x2, m1, m2, 1.0 / (H * W), out=xt, channel_offset=C1)
        O = _native.mode_mix(xt.view(B, Ci, 2, m1 * m2), [w1, w2], 0)
        same = (H, W) == (Ho, Wo)
        mix_last = same or Ho * Wo < H * W
        t = fused = None
        if not mix_last and not half_weights and x1.dtype == torch.float32:
            # up-sampling block: the 1x1 convolution first, its result joins the inverse transform's (one pass over the output)
            t = _mix2_forward(x1.view(B, C1, -1), x2.view(B, C2, -1), cwm, cb).view(B, Co, H, W)
            fused = _fused_addend(t, Ho, Wo, m1, m2, False)
        s = _native.dft2d_inverse(O.view(B, Co, 2 * m1, m2), Ho, Wo, 1.0, True, True, dtype=x1.dtype, addend=fused)
        # point-wise branch accumulates into s
        if fused is not None:
            a1, a2 = x1, x2
        elif mix_last:
            a1 = x1 if same else resample_forward(x1, Ho, Wo)
            a2 = x2 if same else resample_forward(x2, Ho, Wo)
            _mix2_forward(a1.view(B, C1, -1), a2.view(B, C2, -1), cwm, cb, out=s.view(B, Co, -1), accumulate=True)
        else:
            a1, a2 = x1, x2
            if t is None:
                t = _mix2_forward(x1.view(B, C1, -1), x2.view(B, C2, -1), cwm, cb).view(B, Co, H, W)
            resample_forward(t, Ho, Wo, out=s)
        ctx.save_for_backward(xt, w1, w2, cwm, a1, a2)
        ctx.geom = (H, W, same, mix_last, cb is not None, tuple(cw.shape))
        return s

    @staticmethod
    @once_differentiable
    def backward(ctx, gs):
        from .resample import resample_adjoint
        xt, w1, w2, cwm, a1, a2 = ctx.saved_tensors
        H, W, same, mix_last, has_bias, cw_shape = ctx.geom
        gs = _plain(gs)
        B, Co, Ho, Wo = gs.shape
        Ci, _, m1, m2 = w1.shape[:4]
        C1, C2 = a1.shape[1], a2.shape[1]
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_gw = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        need_gc = ctx.needs_input_grad[4] or (has_bias and ctx.needs_input_grad[5])
        lw1, lw2, lcw, lcb = ctx.leaves
        both_gw = ctx.needs_input_grad[2] and ctx.needs_input_grad[3]
        gslot = _stack_grad_slot(ctx.stack[0], ctx.stack[1], Co) if (ctx.stack is not None and need_gw) else None
        g_pre = None
        if not mix_last and REVERSE_SWEEP_RESAMPLE:
            g_pre = resample_adjoint(gs, H, W, reverse=True).view(B, Co, -1)       # right before K1 reads gs, in the opposite image order
        gO = _native.dft2d_forward(gs, m1, m2, 1.0, True, True, out=gslot)             # c (.) keep (.) DFT_trunc(gs)
        gw1 = gw2 = gXp = None
        if gslot is not None:
            gw1, gw2 = _stack_arrived(ctx.stack[0], ctx.stack[1], (lw1, lw2), w1.shape, both_gw)
        elif need_gw:
            _note_use(lw1)
            tg = _grad_targets((lw1, lw2)) if both_gw else None
            if (need1 or need2) and w1.dtype == torch.complex64:
                gXp, (gw1, gw2) = _native.mode_backward(xt, gO, [w1, w2], out=[tg[0][0], tg[1][0]] if tg else None,
                                                        accumulate=bool(tg and tg[0][1]))
                gXp = gXp.view(B, Ci, 2 * m1, m2)
            else:
                gw1, gw2 = _native.mode_wgrad(xt, gO, tuple(w1.shape[:4]), 2, out=[tg[0][0], tg[1][0]] if tg else None,
                                              accumulate=bool(tg and tg[0][1]))
            if tg:
                gw1, gw2 = tg[0][2], tg[1][2]
        gx1 = gx2 = None
        defer = ctx.defer if (need2 and ctx.defer is not None and ctx.defer.owner) else None    # owner's backward still to come
        if need1 or need2:
            gX = gXp if gXp is not None else _native.mode_mix(gO.view(B, Co, 2, m1 * m2), [w1, w2], 1).view(B, Ci, 2 * m1, m2)
            if need1:
                gx1 = _native.dft2d_inverse(gX, H, W, 1.0 / (H * W), False, False, channels=C1, channel_offset=0, dtype=gs.dtype)
            if defer is not None:
                # x2's gradient is completed by x2's first consumer (GradJoin): leave the spectrum, transform nothing
                defer.spectra.append((gX[:, C1:].contiguous(), (H, W)))
            elif need2:
                gx2 = _native.dft2d_inverse(gX, H, W, 1.0 / (H * W), False, False, channels=C2, channel_offset=C1, dtype=gs.dtype)
        gcw = gcb = None
        both = gx1 is not None and gx2 is not None
        if defer is not None:
            # point-wise part of x2's gradient: accumulated into the joined buffer later; x1's part now
            cw2 = cwm[:, C1:].contiguous()
            def mix_into(out, dg=None):
                _native.channel_mix(g_src, cw2, None, transpose_w=True, out=out.view(B, C2, -1),
                                    dgelu_of=None if dg is None else dg.view(B, C2, -1), dgelu_total=dg is not None)
            if mix_last:
                g_src = gs.view(B, Co, -1)
                if same:
                    defer.pending.append((mix_into, True))
                else:
                    defer.pending.append((lambda out: resample_adjoint(
                        _native.channel_mix(g_src, cw2, None, transpose_w=True).view(B, C2, Ho, Wo), H, W, out=out), False))
            else:
                g_src = g_pre if g_pre is not None else resample_adjoint(gs, H, W).view(B, Co, -1)
                defer.pending.append((mix_into, True))
            if gx1 is not None:
                cw1 = cwm[:, :C1].contiguous()
                if mix_last and not same:
                    resample_adjoint(_native.channel_mix(g_src, cw1, None, transpose_w=True).view(B, C1, Ho, Wo), H, W, out=gx1)
                else:
                    _native.channel_mix(g_src, cw1, None, transpose_w=True, out=gx1.view(B, C1, -1))
            if need_gc:
                gcw, gcb = _wgrad_into((lcw, lcb), g_src, a1.view(B, C1, -1), a2.view(B, C2, -1), ctx.needs_input_grad[4],
                                       has_bias and ctx.needs_input_grad[5])
                gcw = None if gcw is None else gcw.view(cw_shape)
            return gx1, None, gw1, gw2, gcw, gcb, None, None, None, None
        if mix_last:
            g_src = gs.view(B, Co, -1)
            if both and same:
                _mix2_input_grads(g_src, cwm, C1, out1=gx1.view(B, C1, -1), out2=gx2.view(B, C2, -1))
            elif both:
                g_a1, g_a2 = _mix2_input_grads(g_src, cwm, C1)
                resample_adjoint(g_a1.view(B, C1, Ho, Wo), H, W, out=gx1)
                resample_adjoint(g_a2.view(B, C2, Ho, Wo), H, W, out=gx2)
            else:
                for gx, cwx, Cx in ((gx1, cwm[:, :C1], C1), (gx2, cwm[:, C1:], C2)):
                    if gx is None:
                        continue
                    if same:
                        _native.channel_mix(g_src, cwx.contiguous(), None, transpose_w=True, out=gx.view(B, Cx, -1))
                    else:
                        g_act = _native.channel_mix(g_src, cwx.contiguous(), None, transpose_w=True)
                        resample_adjoint(g_act.view(B, Cx, Ho, Wo), H, W, out=gx)
        else:
            g_src = g_pre if g_pre is not None else resample_adjoint(gs, H, W).view(B, Co, -1)
            if both:
                _mix2_input_grads(g_src, cwm, C1, out1=gx1.view(B, C1, -1), out2=gx2.view(B, C2, -1))
            else:
                if gx1 is not None:
                    _native.channel_mix(g_src, cwm[:, :C1].contiguous(), None, transpose_w=True, out=gx1.view(B, C1, -1))
                if gx2 is not None:
                    _native.channel_mix(g_src, cwm[:, C1:].contiguous(), None, transpose_w=True, out=gx2.view(B, C2, -1))
        if need_gc:
            gcw, gcb = _wgrad_into((lcw, lcb), g_src, a1.view(B, C1, -1), a2.view(B, C2, -1), ctx.needs_input_grad[4],
                                   has_bias and ctx.needs_input_grad[5])
            gcw = None if gcw is None else gcw.view(cw_shape)
        if ctx.defer is not None and gx2 is not None:       # the owner's backward came first after all
            gx2 = ctx.defer.late(gx2)
        return gx1, gx2, gw1, gw2, gcw, gcb, None, None, None, None


def spectral_conv2d(x, weights1, weights2, dim1, dim2):
    """Functional form of SpectralConv2d_Uno.forward (reference integral_operators.py:181-207)."""
    return _SpectralConv2dFn.apply(x, weights1, weights2, dim1, dim2)


def spectral_conv2d_mixed(x, weights1, weights2, dim1, dim2):
    """Mixed-precision form of the 2-D Fourier integral operator (BASELINE.json config 5: bf16 activations, half-precision
    weight storage, f32 accumulation).  Opt-in: the reference - and SpectralConv2d_Uno.forward here - raise on bf16 input
    (integral_operators.py:187).

    x (B, Ci, H, W) bfloat16 -> (B, Co, dim1, dim2) bfloat16; gradients: gx bfloat16, weights in their own dtype.
    weights1/2: complex64 (Ci, Co, m1, m2), or their half-precision storage (Ci, Co, m1, m2, 2) float16 (re, im), which the
    per-mode GEMM reads as it is (widened in registers; 33 MB at the C5 size).  The pruned DFT kernels read / write the bf16
    tensors directly; the truncated spectrum, the per-mode GEMM and every accumulation are f32 / c64, so the result equals the
    f32 operator applied to the widened inputs, rounded once (to nearest even) on the way out - tests/test_hip_mixed.py."""
    if x.dtype != torch.bfloat16:
        raise RuntimeError(f"spectral_conv2d_mixed: input must be bfloat16 (got {x.dtype})")
    for w in (weights1, weights2):
        if w.dtype == torch.float16 and w.shape[-1] != 2:
            raise RuntimeError("spectral_conv2d_mixed: half-precision weights are stored as (..., 2) = (re, im)")
    if weights1.dtype == torch.float16:
        return _SpectralConv2dHalfFn.apply(x, weights1, weights2, dim1, dim2)
    return _SpectralConv2dFn.apply(x, weights1, weights2, dim1, dim2)


class _SpectralConv2dHalfFn(torch.autograd.Function):
    """spectral_conv2d_mixed with the weights GIVEN in half-precision (re, im) storage: K2 reads them as they are (no widened
    copy); their gradients are accumulated in complex64 and returned rounded once to the storage format."""

    @staticmethod
    def forward(ctx, x, w1h, w2h, Ho, Wo):
        x, w1h, w2h = _plain(x), _plain(w1h), _plain(w2h)
        y, xt = _native.spectral_conv2d_forward(x, w1h, w2h, int(Ho), int(Wo))
        ctx.save_for_backward(xt, w1h, w2h)
        ctx.in_hw = (x.shape[-2], x.shape[-1])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xt, w1h, w2h = ctx.saved_tensors
        need_gx = ctx.needs_input_grad[0]
        need_gw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gx, gw1, gw2 = _native.spectral_conv2d_backward(_plain(gy), xt, w1h, w2h, ctx.in_hw[0], ctx.in_hw[1],
                                                        need_gx=need_gx, need_gw=need_gw)
        if need_gw:
            gw1, gw2 = torch.view_as_real(gw1).half(), torch.view_as_real(gw2).half()
        return gx, gw1, gw2, None, None


# --------------------------------------------------------------------------------------------- 2-D
class SpectralConv2d_Uno(nn.Module):
    """2-D Fourier integral operator (reference integral_operators.py:127-207).

    in_codim / out_codim : input / output co-domain dimension (channels; floats are truncated)
    dim1, dim2           : default output grid size
    modes1, modes2       : Fourier modes kept along each axis; modes1 <= min(dim1, input_dim1) and
                           modes2 <= min(dim2, input_dim2)//2 + 1 (defaults dim1//2-1, dim2//2)
    """

    def __init__(self, in_codim, out_codim, dim1, dim2, modes1=None, modes2=None):
        super().__init__()
        in_codim, out_codim = int(in_codim), int(out_codim)
        self.in_channels = in_codim
        self.out_channels = out_codim
        self.dim1 = dim1
        self.dim2 = dim2
        if modes1 is not None:
            self.modes1, self.modes2 = modes1, modes2
        else:
            self.modes1, self.modes2 = dim1 // 2 - 1, dim2 // 2
        self.scale = (1 / (2 * in_codim)) ** (1.0 / 2.0)
        shape = (in_codim, out_codim, self.modes1, self.modes2)
        self.weights1 = nn.Parameter(self.scale * torch.randn(*shape, dtype=torch.cfloat))
        self.weights2 = nn.Parameter(self.scale * torch.randn(*shape, dtype=torch.cfloat))
        self.mixed_precision = False        # enable_mixed_precision(): accept bfloat16 activations (the reference raises)

    def forward(self, x, dim1=None, dim2=None):
        if dim1 is not None:        # persistent override, as in the reference (:182-184)
            self.dim1 = dim1
            self.dim2 = dim2
        if self.mixed_precision and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == self.in_channels:
            if not self._bf16_kernels():
                # mode counts beyond the MFMA kernels' compiled range run the any-mode forms, which are float32 only: widen the
                # activations for this layer instead of raising at run time (the reference's DEFAULT modes land here)
                return spectral_conv2d(x.float(), self.weights1, self.weights2, self.dim1, self.dim2).to(torch.bfloat16)
            return _SpectralConv2dFn.apply(x, self.weights1, self.weights2, self.dim1, self.dim2, True)
        _check_input(x, 4, self.in_channels, "SpectralConv2d_Uno")
        return spectral_conv2d(x, self.weights1, self.weights2, self.dim1, self.dim2)

    def _bf16_kernels(self):
        """True when the bf16-image transform kernels cover this layer's mode counts (csrc/capi.hip: modes1 <= 40, modes2 <= 48)."""
        return self.modes1 <= 40 and self.modes2 <= 48


class pointwise_op_2D(nn.Module):
    """1x1 convolution followed by bicubic anti-aliased resampling to (dim1, dim2)
    (reference integral_operators.py:210-243)."""

    def __init__(self, in_codim, out_codim, dim1, dim2):
        super().__init__()
        self.conv = nn.Conv2d(int(in_codim), int(out_codim), 1)
        self.dim1 = int(dim1)
        self.dim2 = int(dim2)

    def forward(self, x, dim1=None, dim2=None):
        if dim1 is None:
            dim1, dim2 = self.dim1, self.dim2
        if not (_dev_act(x) and x.dim() == 4):
            return F.interpolate(self.conv(x), size=(dim1, dim2), mode="bicubic", align_corners=True, antialias=True)
        # HIP path: the resampling is a separable banded operator with the reference's weights (uno_amd/resample.py).
        # It commutes with the 1x1 convolution (both linear, resampling rows sum to 1 so the bias passes through),
        # so the convolution runs on whichever side has fewer pixels.
        from .resample import resample2d_bicubic_aa
        if dim1 * dim2 < x.shape[-2] * x.shape[-1]:
            return channel_mix(resample2d_bicubic_aa(x, dim1, dim2), self.conv.weight, self.conv.bias)
        return resample2d_bicubic_aa(channel_mix(x.contiguous(), self.conv.weight, self.conv.bias), dim1, dim2)


class OperatorBlock_2D(nn.Module):
    """gelu( [InstanceNorm]( SpectralConv2d(x) + pointwise(x) ) )  (reference integral_operators.py:246-284)."""

    def __init__(self, in_codim, out_codim, dim1, dim2, modes1, modes2, Normalize=False, Non_Lin=True):
        super().__init__()
        self.conv = SpectralConv2d_Uno(in_codim, out_codim, dim1, dim2, modes1, modes2)
        self.w = pointwise_op_2D(in_codim, out_codim, dim1, dim2)
        self.normalize = Normalize
        self.non_lin = Non_Lin
        if Normalize:
            self.normalize_layer = nn.InstanceNorm2d(int(out_codim), affine=True)

    def forward(self, x, dim1=None, dim2=None, *, join=None, out_join=None):
        """join / out_join (optional, beyond the reference signature): GradJoin objects of the input / of this block's output - see
        GradJoin and _OperatorBlock2dFn."""
        if self.non_lin and not self.normalize:
            return self._branches(x, dim1, dim2, gelu=True, join=join, out_join=out_join)
        if out_join is not None:            # no GELU straight after the sum: this block leaves no pre-activation tensor to a join
            out_join.void()
        out = self._branches(x, dim1, dim2, join=join)
        if self.normalize:
            return instance_norm_gelu(out, self.normalize_layer, self.non_lin)
        return out

    def forward_cat(self, xs, dim1=None, dim2=None, defer_gelu=False, defer_grad=None):
        """self(torch.cat(xs, dim=1), dim1, dim2) for a skip connection (reference darcy_flow_uno2d.py:117-125) - two
        float32 device tensors are consumed in place, the concatenation is never built.  defer_gelu (blocks without
        normalisation only): return the PRE-activation sum; the caller's consumer applies the GELU as it reads it."""
        if defer_gelu and (self.normalize or not self.non_lin):
            raise ValueError("defer_gelu needs a block with Non_Lin=True and Normalize=False")
        xs = list(xs)
        conv, w = self.conv, self.w
        d1, d2 = (dim1, dim2) if dim1 is not None else (w.dim1, w.dim2)
        cdims = (dim1, dim2) if dim1 is not None else (conv.dim1, conv.dim2)
        fused = (len(xs) == 2 and all(self._takes(x) for x in xs) and xs[0].dtype == xs[1].dtype
                 and xs[0].shape[0] == xs[1].shape[0] and xs[0].shape[2:] == xs[1].shape[2:]
                 and xs[0].shape[1] + xs[1].shape[1] == conv.in_channels and cdims == (d1, d2)
                 and w.conv.weight.dtype == torch.float32)
        if not fused:
            if defer_grad is not None:      # xs[1]'s gradient reaches its producer through autograd, without the join's gelu' factor
                defer_grad.void()
            if defer_gelu:
                return self._branches(torch.cat(xs, dim=1), dim1, dim2)
            return self.forward(torch.cat(xs, dim=1), dim1, dim2)
        if dim1 is not None:
            conv.dim1, conv.dim2 = dim1, dim2
        out = _OperatorBlock2dCatFn.apply(xs[0], xs[1], conv.weights1, conv.weights2, w.conv.weight, w.conv.bias, int(d1), int(d2),
                                          xs[0].dtype == torch.bfloat16, defer_grad)
        if defer_gelu:
            return out
        if self.normalize:
            return instance_norm_gelu(out, self.normalize_layer, self.non_lin)
        return F.gelu(out) if self.non_lin else out

    def _branches(self, x, dim1, dim2, gelu=False, join=None, out_join=None):
        """conv(x) + w(x) [then GELU when `gelu`: written by the kernel that completes the sum where the fused path runs]"""
        conv, w = self.conv, self.w
        if dim1 is not None:        # the spectral layer keeps a call-time override, the point-wise one does not (:182-184, :236-238)
            conv.dim1, conv.dim2 = dim1, dim2
            d1, d2 = dim1, dim2
        else:
            d1, d2 = w.dim1, w.dim2
        fused = (self._takes(x) and (conv.dim1, conv.dim2) == (d1, d2)
                 and x.shape[1] == conv.in_channels and w.conv.weight.dtype == torch.float32)
        if not fused:               # CPU tensors raise inside the spectral layer; mismatched grids raise at the sum
            for j in (join, out_join):      # stock-op path: nothing here completes a joined gradient or leaves a pre-activation sum
                if j is not None:
                    j.void()
            out = conv(x) + w(x, d1, d2)
            return F.gelu(out) if gelu else out
        return _OperatorBlock2dFn.apply(x, conv.weights1, conv.weights2, w.conv.weight, w.conv.bias, int(d1), int(d2),
                                        x.dtype == torch.bfloat16, bool(gelu), join, out_join)

    def _takes(self, x):
        """4-D device tensor the fused block kernels take: float32, or bfloat16 once the spectral layer is in mixed-precision mode."""
        return (x.is_cuda and x.dim() == 4 and
                (x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and getattr(self.conv, "mixed_precision", False)
                                              and self.conv._bf16_kernels())))


# --------------------------------------------------------------------------------------------- 3-D
class SpectralConv3d_Uno(nn.Module):
    """3-D Fourier integral operator (reference integral_operators.py:287-427): rfftn over the last
    three axes, four low-frequency corners (weights1..4 = (lo,lo), (hi,lo), (lo,hi), (hi,hi)),
    zero-padded irfftn to (dim1, dim2, dim3)."""

    def __init__(self, in_codim, out_codim, dim1, dim2, dim3, modes1=None, modes2=None, modes3=None):
        super().__init__()
        in_codim, out_codim = int(in_codim), int(out_codim)
        self.in_channels = in_codim
        self.out_channels = out_codim
        self.dim1, self.dim2, self.dim3 = dim1, dim2, dim3
        if modes1 is not None:
            self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3
        else:
            self.modes1, self.modes2, self.modes3 = dim1, dim2, dim3 // 2 + 1
        self.scale = (1 / (2 * in_codim)) ** (1.0 / 2.0)
        shape = (in_codim, out_codim, self.modes1, self.modes2, self.modes3)
        self.weights1 = nn.Parameter(self.scale * torch.randn(*shape, dtype=torch.cfloat))
        self.weights2 = nn.Parameter(self.scale * torch.randn(*shape, dtype=torch.cfloat))
        self.weights3 = nn.Parameter(self.scale * torch.randn(*shape, dtype=torch.cfloat))
        self.weights4 = nn.Parameter(self.scale * torch.randn(*shape, dtype=torch.cfloat))

    def forward(self, x, dim1=None, dim2=None, dim3=None):
        if dim1 is not None:
            self.dim1, self.dim2, self.dim3 = dim1, dim2, dim3
        _check_input(x, 5, self.in_channels, "SpectralConv3d_Uno")
        from .spectral3d import spectral_conv3d
        return spectral_conv3d(x, [self.weights1, self.weights2, self.weights3, self.weights4],
                               self.dim1, self.dim2, self.dim3)


def _kept_indices(n_in: int, n_out: int):
    """Spectrum indices along a complex axis that survive the reference's corner copies into an input-sized zero spectrum
    (`ft_u[:h] = ft[:h]`, `ft_u[-h:] = ft[-h:]`, h = n_out // 2 - with Python's `-0:` meaning everything) and irfftn's trimming
    to n_out entries (integral_operators.py:450-463)."""
    h = n_out // 2
    idx = set(range(0, min(h, n_in)))
    idx |= set(range(n_in)) if h == 0 else set(range(max(n_in - h, 0), n_in))
    return sorted(r for r in idx if r < min(n_in, n_out))


_RESAMPLE3D_TABLES = {}


def _resample3d_plan(din, dout, device):
    """(f1, f2, m3) for _native.fft_resample3d, or None when the shape is outside the kernels' range (odd row counts, too many
    rows or bins, planes too large): the caller then takes the stock FFT path."""
    key = (tuple(din), tuple(dout), str(device))
    if key not in _RESAMPLE3D_TABLES:
        plan = None
        k1, k2 = _kept_indices(din[0], dout[0]), _kept_indices(din[1], dout[1])
        m3 = min(dout[2] // 2, din[2] // 2 + 1)
        ok = (len(k1) >= 2 and len(k1) % 2 == 0 and len(k1) <= 80 and len(k2) >= 2 and len(k2) % 2 == 0 and len(k2) <= 48
              and 1 <= m3 <= 16 and 16 <= din[1] * din[2] <= 1792 and 16 <= dout[1] * dout[2] <= 1792
              and din[2] <= 64 and dout[2] <= 64)
        if ok:
            t1 = _native.table_to_device(torch.tensor(k1, dtype=torch.int32), device)
            t2 = _native.table_to_device(torch.tensor(k2, dtype=torch.int32), device)
            plan = (t1, t2, m3)
        _RESAMPLE3D_TABLES[key] = plan
    return _RESAMPLE3D_TABLES[key]


class _FftResample3dFn(torch.autograd.Function):
    """irfftn(corner-copy(rfftn(x)), s=size) of pointwise_op_3D on the pruned-DFT kernels (K1p, K5, K6, K3p with explicit
    frequency tables); backward is the transpose: the same kernels with sizes swapped and the Hermitian weights on the other side."""

    @staticmethod
    def forward(ctx, x, size, plan):
        t1, t2, m3 = plan
        ctx.plan, ctx.din, ctx.dout = plan, tuple(x.shape[-3:]), tuple(size)
        scale = 1.0 / (size[0] * size[1] * size[2])
        return _native.fft_resample3d(_plain(x), size, (t1, t1), (t2, t2), m3, scale, adjoint=False)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        t1, t2, m3 = ctx.plan
        scale = 1.0 / (ctx.dout[0] * ctx.dout[1] * ctx.dout[2])
        return _native.fft_resample3d(_plain(gy), ctx.din, (t1, t1), (t2, t2), m3, scale, adjoint=True), None, None


# pointwise_op_3D on a grid its pruned-DFT resampling kernels do not take (_resample3d_plan is None): False (default) - raise, naming the
# limits; True - run the reference's op sequence on torch.fft (rocFFT on the device: a stock-library dispatch the caller asked for)
STOCK_FFT_RESAMPLE3D = False


class pointwise_op_3D(nn.Module):
    """1x1x1 convolution + the reference's FFT crop/resample (quirks kept bug-for-bug: unnormalised
    forward transform, corners copied into an INPUT-sized zero spectrum, irfftn(s=output dims) that
    trims/zero-pads at the END of each axis, identity trilinear resize) - reference
    integral_operators.py:430-468.  The convolution runs on the channel-mix kernels (K8 / K9) for float32 device
    tensors - MIOpen executes a 1x1x1 Conv3d with its naive direct kernels, 0.9 s of a 2.2 s first NS-3D step - the
    FFT resampling runs on the pruned-DFT kernels (_FftResample3dFn; outside their shape range the layer raises unless STOCK_FFT_RESAMPLE3D allows torch.fft); the trilinear resize to the size the tensor already has is an exact
    identity under align_corners=True and is skipped on the device."""

    def __init__(self, in_codim, out_codim, dim1, dim2, dim3):
        super().__init__()
        self.conv = nn.Conv3d(int(in_codim), int(out_codim), 1)
        self.dim1, self.dim2, self.dim3 = int(dim1), int(dim2), int(dim3)

    def _corner_mask(self, spec, h1, h2, h3):
        """(1, 1, D1, D2, D3/2+1) float mask of the entries the reference copies (integral_operators.py:450-461), cached per shape."""
        key = (tuple(spec.shape[2:]), h1, h2, h3, str(spec.device))
        cache = self.__dict__.setdefault("_mask_cache", {})
        if key not in cache:
            m = torch.zeros((1, 1, *spec.shape[2:]), dtype=torch.float32)
            for rows in (slice(None, h1), slice(-h1, None)):
                for cols in (slice(None, h2), slice(-h2, None)):
                    m[:, :, rows, cols, :h3] = 1.0
            cache[key] = m.to(spec.device)
        return cache[key]

    def forward(self, x, dim1=None, dim2=None, dim3=None):
        if dim1 is None:
            dim1, dim2, dim3 = self.dim1, self.dim2, self.dim3
        on_device = x.is_cuda and x.dtype == torch.float32 and x.dim() == 5
        out = channel_mix(x.contiguous(), self.conv.weight, self.conv.bias) if on_device else self.conv(x)
        if on_device:
            plan = _resample3d_plan(out.shape[-3:], (dim1, dim2, dim3), out.device)
            if plan is not None:
                return _FftResample3dFn.apply(out, (dim1, dim2, dim3), plan)
        if on_device and not STOCK_FFT_RESAMPLE3D:
            # no silent dispatch to a stock library from a product component: the pruned-DFT resampling kernels do not cover this grid
            raise RuntimeError(
                f"pointwise_op_3D: the FFT crop / resample {tuple(out.shape[-3:])} -> {(dim1, dim2, dim3)} is outside the range of the "
                "pruned-DFT kernels (they take an even number of kept rows per complex axis - at most 80 / 48 - and (W, T) planes of at most "
                "1792 elements with T <= 64); set uno_amd.integral_operators.STOCK_FFT_RESAMPLE3D = True to run this layer's resampling "
                "through torch.fft (rocFFT) instead")
        spec = torch.fft.rfftn(out, dim=[-3, -2, -1])
        h1, h2, h3 = dim1 // 2, dim2 // 2, dim3 // 2
        if on_device:
            # the four corner copies into a zero spectrum == one multiplication by a 0 / 1 mask (same values bit for bit;
            # one pass forward and backward instead of zeros_like + 4 slice copies and their CopySlices backward chain)
            out = torch.fft.irfftn(spec * self._corner_mask(spec, h1, h2, h3), s=(dim1, dim2, dim3))
            return out
        kept = torch.zeros_like(spec)
        for rows in (slice(None, h1), slice(-h1, None)):
            for cols in (slice(None, h2), slice(-h2, None)):
                kept[:, :, rows, cols, :h3] = spec[:, :, rows, cols, :h3]
        out = torch.fft.irfftn(kept, s=(dim1, dim2, dim3))
        if on_device:
            return out
        return F.interpolate(out, size=(dim1, dim2, dim3), mode="trilinear", align_corners=True)


ONE_BUFFER_3D = True        # OperatorBlock_3D in one buffer (_OperatorBlock3dFn); False: the two branches and stock sum / GELU (A/B switch)


class _OperatorBlock3dFn(torch.autograd.Function):
    """s = SpectralConv3d_Uno(x) + pointwise_op_3D(x) in ONE buffer (reference integral_operators.py:506-512: `x1_out = self.conv(...);
    x2_out = self.w(...); x_out = x1_out + x2_out`, then F.gelu for blocks without normalisation).

    The spectral branch's inverse transform writes s; the point-wise branch - 1x1x1 convolution (K8), then the reference's FFT crop /
    resample on the pruned-DFT kernels - ends in a plane-batched inverse transform that ACCUMULATES into s and, for a block whose sum is
    followed directly by the GELU, writes the activation in the same pass (uno_fft_resample3d_acc).  Backward: the spectral branch
    writes grad_x, the transposed 1x1x1 convolution accumulates into it.  The element-wise sum (three passes over the output), the
    GELU (two) and autograd's sum of the two input gradients (three over the input) are gone."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, w4, cw, cb, dims, plan, fuse_gelu):
        ctx.leaves = (cw, cb)
        x = _plain(x)
        ws = [_plain(w) for w in (w1, w2, w3, w4)]
        B, Ci = x.shape[0], x.shape[1]
        din = tuple(x.shape[2:])
        Co = cw.shape[0]
        cwm = _plain(cw).reshape(Co, Ci)
        cbp = None if cb is None else _plain(cb)
        s, xt = _native.spectral_conv3d_forward(x, ws, *dims)
        t = _native.channel_mix(x.view(B, Ci, -1), cwm, cbp).view(B, Co, *din)
        t1, t2, m3 = plan
        scale = 1.0 / (dims[0] * dims[1] * dims[2])
        if fuse_gelu:
            s, out = _native.fft_resample3d(t, dims, (t1, t1), (t2, t2), m3, scale, adjoint=False, out=s, act=True)
        else:
            out = _native.fft_resample3d(t, dims, (t1, t1), (t2, t2), m3, scale, adjoint=False, out=s)
        ctx.save_for_backward(xt, *ws, cwm, x, s if fuse_gelu else None)
        ctx.geom = (din, tuple(dims), plan, cb is not None, tuple(cw.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xt, w1, w2, w3, w4, cwm, x, pre = ctx.saved_tensors
        din, dims, plan, has_bias, cw_shape = ctx.geom
        g = _plain(g)
        if pre is not None:
            g = torch.ops.aten.gelu_backward(g, pre)
        B, Co = g.shape[0], g.shape[1]
        Ci = cwm.shape[1]
        need_gx = ctx.needs_input_grad[0]
        need_gw = any(ctx.needs_input_grad[1:5])
        need_gc = ctx.needs_input_grad[5] or (has_bias and ctx.needs_input_grad[6])
        gx, gws = _native.spectral_conv3d_backward(g, xt, [w1, w2, w3, w4], *din, need_gx=need_gx, need_gw=need_gw)
        gws = gws or [None] * 4
        gcw = gcb = None
        if need_gx or need_gc:
            t1, t2, m3 = plan
            scale = 1.0 / (dims[0] * dims[1] * dims[2])
            g_t = _native.fft_resample3d(g, din, (t1, t1), (t2, t2), m3, scale, adjoint=True).view(B, Co, -1)
            if need_gx:
                _native.channel_mix(g_t, cwm, None, transpose_w=True, out=gx.view(B, Ci, -1))        # accumulates into the spectral branch's gx
            if need_gc:
                gcw, gcb = _wgrad_into(ctx.leaves, g_t, x.view(B, Ci, -1), None, ctx.needs_input_grad[5], has_bias and ctx.needs_input_grad[6])
                gcw = None if gcw is None else gcw.view(cw_shape)
        return (gx, *gws, gcw, gcb, None, None, None)


class OperatorBlock_3D(nn.Module):
    """gelu( [InstanceNorm3d]( SpectralConv3d(x) + pointwise(x) ) )  (reference integral_operators.py:471-513)."""

    def __init__(self, in_codim, out_codim, dim1, dim2, dim3, modes1, modes2, modes3, Normalize=False, Non_Lin=True):
        super().__init__()
        self.conv = SpectralConv3d_Uno(in_codim, out_codim, dim1, dim2, dim3, modes1, modes2, modes3)
        self.w = pointwise_op_3D(in_codim, out_codim, dim1, dim2, dim3)
        self.normalize = Normalize
        self.non_lin = Non_Lin
        if Normalize:
            self.normalize_layer = nn.InstanceNorm3d(int(out_codim), affine=True)

    def forward(self, x, dim1=None, dim2=None, dim3=None):
        fused = self._fused(x, dim1, dim2, dim3)
        if fused is not None:
            out, activated = fused
            if self.normalize:
                return instance_norm_gelu(out, self.normalize_layer, self.non_lin)
            return out if (activated or not self.non_lin) else F.gelu(out)
        out = self.conv(x, dim1, dim2, dim3) + self.w(x, dim1, dim2, dim3)
        if self.normalize:
            return instance_norm_gelu(out, self.normalize_layer, self.non_lin)
        if self.non_lin:
            out = F.gelu(out)
        return out

    def _fused(self, x, dim1, dim2, dim3):
        """(conv(x) + w(x) [activated], whether the GELU has been applied) through the one-buffer form, or None when the layer has to take
        the two branches separately (CPU tensors, other dtypes, grids outside the pruned-DFT resampling kernels' range, mismatched grids)."""
        conv, w = self.conv, self.w
        if not (ONE_BUFFER_3D and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == conv.in_channels
                and w.conv.weight.dtype == torch.float32):
            return None
        if dim1 is not None:        # the spectral layer keeps a call-time override, the point-wise one does not (:391-394, :444-446)
            dims = (int(dim1), int(dim2), int(dim3))
        else:
            dims = (int(w.dim1), int(w.dim2), int(w.dim3))
            if (conv.dim1, conv.dim2, conv.dim3) != dims:
                return None
        plan = _resample3d_plan(tuple(x.shape[-3:]), dims, x.device)
        if plan is None:
            return None
        if dim1 is not None:
            conv.dim1, conv.dim2, conv.dim3 = dim1, dim2, dim3
        gelu = self.non_lin and not self.normalize
        out = _OperatorBlock3dFn.apply(x, conv.weights1, conv.weights2, conv.weights3, conv.weights4, w.conv.weight, w.conv.bias,
                                       dims, plan, gelu)
        return out, gelu


# --------------------------------------------------------------------------------------------- 1-D
# Not on the north-star path and unused by every reference model; the
# names stay importable so `from integral_operators import *` keeps its surface.  Stock torch ops.
class SpectralConv1d_Uno(nn.Module):
    """1-D Fourier layer (reference integral_operators.py:7-72): single corner [:modes1]."""

    def __init__(self, in_codim, out_codim, dim1, modes1=None):
        super().__init__()
        in_codim, out_codim = int(in_codim), int(out_codim)
        self.in_channels, self.out_channels = in_codim, out_codim
        self.dim1 = dim1
        self.modes1 = modes1 if modes1 is not None else dim1 // 2
        self.scale = (1 / (2 * in_codim)) ** (1.0 / 2.0)
        self.weights1 = nn.Parameter(self.scale * torch.randn(in_codim, out_codim, self.modes1, dtype=torch.cfloat))

    def forward(self, x, dim1=None):
        if dim1 is not None:
            self.dim1 = dim1
        if x.is_cuda:
            # the 1-D layer is the 2-D one on a grid of one row: the row DFT of length 1 is the identity, both "corners" are the
            # row of frequency 0 and the later-wins rule keeps the second one - weights1 in both slots, the masked slot's
            # gradient is exactly zero
            _check_input(x, 3, self.in_channels, "SpectralConv1d_Uno")
            w = self.weights1.unsqueeze(2)
            return spectral_conv2d(x.unsqueeze(2), w, w, 1, self.dim1).squeeze(2)
        spec = torch.fft.rfft(x, norm="forward")
        out = torch.zeros(x.shape[0], self.out_channels, self.dim1 // 2 + 1, dtype=torch.cfloat, device=x.device)
        out[:, :, : self.modes1] = torch.einsum("bix,iox->box", spec[:, :, : self.modes1], self.weights1)
        return torch.fft.irfft(out, n=self.dim1, norm="forward")


class pointwise_op_1D(nn.Module):
    """1x1 conv + linear resampling (reference integral_operators.py:75-93; the reference's
    antialias=True is rejected by torch >= 2 for mode='linear', so it is not requested here)."""

    def __init__(self, in_codim, out_codim, dim1):
        super().__init__()
        self.conv = nn.Conv1d(int(in_codim), int(out_codim), 1)
        self.dim1 = int(dim1)

    def forward(self, x, dim1=None):
        if dim1 is None:
            dim1 = self.dim1
        return F.interpolate(self.conv(x), size=dim1, mode="linear", align_corners=True)


class OperatorBlock_1D(nn.Module):
    """reference integral_operators.py:96-124."""

    def __init__(self, in_codim, out_codim, dim1, modes1, Normalize=True, Non_Lin=True):
        super().__init__()
        self.conv = SpectralConv1d_Uno(in_codim, out_codim, dim1, modes1)
        self.w = pointwise_op_1D(in_codim, out_codim, dim1)
        self.normalize = Normalize
        self.non_lin = Non_Lin
        if Normalize:
            self.normalize_layer = nn.InstanceNorm1d(int(out_codim), affine=True)

    def forward(self, x, dim1=None):
        out = self.conv(x, dim1) + self.w(x, dim1)
        if self.normalize:
            out = self.normalize_layer(out)
        if self.non_lin:
            out = F.gelu(out)
        return out
