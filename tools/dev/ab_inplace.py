"""A/B on one box: weight gradients written in place (integral_operators.INPLACE_PARAM_GRADS) on / off - NS-2D graph step, Darcy step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import uno_amd.integral_operators as io
from uno_amd.harness import UNO, UNO_9, ComplexAdam, DarcyTrainer, GraphedStep, ns2d_rollout_loss, synthetic_darcy_batch
dev = torch.device("cuda:0")


def timeit(step, n=5, reps=3):
    for _ in range(2):
        step()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3


for rnd in range(2):
    for flag in (False, True):
        io.INPLACE_PARAM_GRADS = flag
        torch.manual_seed(0)
        m = UNO(14, 32).to(dev)
        xx, yy = torch.randn(32, 64, 64, 10, device=dev), torch.randn(32, 64, 64, 40, device=dev)
        opt = ComplexAdam(m.parameters(), lr=1e-3, weight_decay=1e-4)
        gs = GraphedStep(m, opt, lambda a_, b_: ns2d_rollout_loss(m, a_, b_, T_f=40, step=1), (xx, yy))
        t_ns = timeit(lambda: gs.step(xx, yy), n=4)
        del gs, m, opt
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        model = UNO_9(3, 64, pad=5).to(dev)
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
        a, u = synthetic_darcy_batch(16, 421, 1234, dev)
        t_d = timeit(lambda: tr.step(a, u), n=10)
        del tr, model
        torch.cuda.empty_cache()
        print(f"round {rnd} inplace={flag}: NS-2D graph step {t_ns:.2f} ms, Darcy step {t_d:.3f} ms", flush=True)
