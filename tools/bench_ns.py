# Throughput of the other two training workloads of the reference on one MI355X (configs C3 and C4 of SURVEY.md 8(d)):
#   C3  NS-2D  UNO(14, 32), 64x64, batch 32, autoregressive roll-out of T_f = 40 steps, one backward through the chain
#   C4  NS-3D  Uno3D_T20(6, w, pad=3), 64x64x10 -> 64x64x20, batch 8, w = 8 (reference default) and 32
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
for _a in sys.argv[1:]:
    if _a.endswith(".so"):          # a library variant (tools/dev/mkvariant.py)
        from uno_amd import _native
        _native.LIB_PATH = os.path.abspath(_a)
from uno_amd.harness import UNO, Uno3D_T20, ComplexAdam, GraphedStep, ns2d_rollout_loss, ns3d_loss
dev = torch.device("cuda:0")


GRAPH = "--graph" in sys.argv       # forward + backward replayed from a HIP graph (harness.GraphedStep)
if "--no-time-batch" in sys.argv:   # A/B: the spectral weight gradient of the roll-out use by use (40 accumulating GEMMs per layer)
    import uno_amd.integral_operators as _io
    _io.TIME_BATCHED_WGRAD = False


def run(name, model, closure, samples, steps=5, warmup=2, inputs=()):
    opt = ComplexAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, capturable=GRAPH and "--eager-adam" not in sys.argv)
    if GRAPH:
        gs = GraphedStep(model, opt, closure, inputs)
        name += " [HIP graph]"
        def step():
            return gs.step(*inputs)
    else:
        def step():
            opt.zero_grad(set_to_none=True)
            loss = closure(*inputs)
            loss.backward()
            opt.step()
            return loss
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {dt*1e3:8.1f} ms/step  {samples/dt:8.1f} samples/s  (loss {float(loss):.4f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB)", flush=True)


torch.manual_seed(0)
m = UNO(14, 32).to(dev)
xx = torch.randn(32, 64, 64, 10, device=dev); yy = torch.randn(32, 64, 64, 40, device=dev)
run("C3 NS-2D UNO(14,32) 64^2 B=32 T_f=40", m, lambda a, b: ns2d_rollout_loss(m, a, b, T_f=40, step=1), 32, inputs=(xx, yy))
del m
for w in (() if "--c3" in sys.argv else (8, 32)):
    torch.manual_seed(0)
    m3 = Uno3D_T20(6, w, pad=3).to(dev)
    x = torch.randn(8, 64, 64, 10, 1, device=dev); y = torch.randn(8, 64, 64, 20, device=dev)
    run(f"C4 NS-3D Uno3D_T20(6,{w},pad=3) 64x64x10 B=8", m3, lambda a, b: ns3d_loss(m3, a, b), 8, inputs=(x, y))
    del m3
