"""A/B of the fused `inverse transform + up-sampled addend` kernel (K3-A) in one process on one box:
  (1) kernel level at the two Darcy shapes: K3 alone, K3 + accumulating K7, K3-A
  (2) the training step with each of the round's switches on / off (alternating groups): integral_operators.FUSE_UPSAMPLE_ADD,
      PROJECT_BACKWARD_FUSED, PAIR_BACKWARD_GEMMS, REVERSE_SWEEP_RESAMPLE, uno_sweep_alternation
python tools/dev/fusetime.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native, resample as rs
import uno_amd.integral_operators as io
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch

dev = torch.device("cuda:0")


def timed(fn, iters=20, reps=5, warm=3):
    for _ in range(warm):
        fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    out.sort()
    return out[len(out) // 2]


g = torch.Generator().manual_seed(0)
for (B, C, Hs, H, m, adj) in ((16, 64, 223, 446, 18, False), (16, 128, 111, 223, 8, False), (16, 64, 223, 446, 18, True), (16, 128, 111, 223, 8, True)):
    spec = torch.randn(B, C, 2 * m, m, dtype=torch.complex64, generator=g).to(dev)
    t = torch.randn(B, C, Hs, Hs, generator=g).to(dev)
    tabs = rs.upsample_add_tables(Hs, Hs, H, H, str(dev), adj)
    k3 = timed(lambda: _native.dft2d_inverse(spec, H, H, 1.0, True, True))

    def two():
        s = _native.dft2d_inverse(spec, H, H, 1.0, True, True)
        (rs.resample_adjoint if adj else rs.resample_forward)(t, H, H, out=s)
    k37 = timed(two)
    k3a = timed(lambda: _native.dft2d_inverse(spec, H, H, 1.0, True, True, addend=(t, tabs)))
    by = B * C * (H * H + Hs * Hs) * 4
    print(f"{B}x{C} {Hs}^2 -> {H}^2 modes {m} adjoint={adj}: K3 {k3:.1f} us | K3 + K7acc {k37:.1f} us | K3-A {k3a:.1f} us = {by / k3a / 1e6:.2f} TB/s of out + t", flush=True)
    del spec, t

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
model = UNO_9(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(16, 421, 1234, dev)


def ab(label, on_label, off_label, setter):
    res = {True: [], False: []}
    for rnd in range(3):
        for on in (True, False):
            setter(on)
            for _ in range(3):
                tr.step(a, u)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = tr.step(a, u)
            torch.cuda.synchronize()
            res[on].append((time.perf_counter() - t0) / steps * 1e3)
            print(f"round {rnd} {label}={on}: {res[on][-1]:.3f} ms/step loss {float(loss):.6f}", flush=True)
    setter(True)
    print(on_label, min(res[True]), off_label, min(res[False]), flush=True)


ab("fuse", "fused", "two-kernel", lambda v: setattr(io, "FUSE_UPSAMPLE_ADD", v))
ab("project_backward_fused", "fc1 - GELU - fc2 backward without the stored gradient", "three calls", lambda v: setattr(io, "PROJECT_BACKWARD_FUSED", v))
ab("pair_backward_gemms", "paired", "composite with side stream", lambda v: setattr(io, "PAIR_BACKWARD_GEMMS", v))
ab("reverse_sweep_resample", "reverse sweep before K1", "after the spectral branch", lambda v: setattr(io, "REVERSE_SWEEP_RESAMPLE", v))
ab("sweep_alternation", "alternating sweeps", "all front to back", _native.sweep_alternation)
