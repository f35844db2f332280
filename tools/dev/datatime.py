"""Does the time of the fc1-shaped kernels depend on the DATA (clock / power management)?  Same calls on random data, on data that
is zero outside the 421 x 421 domain, on all-zero and on small-magnitude data: python tools/dev/datatime.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
B, C, H = 16, 64, 446
torch.manual_seed(0)
w = (torch.randn(64, 128) / 11).to(dev); b = torch.randn(64).to(dev); w2 = torch.randn(64).to(dev); b2 = torch.randn(1).to(dev)
wt = w[:, :64].contiguous()


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l)
    except Exception as e:
        return repr(e)


def case(tag, make):
    x1, x2, gy = make(), make(), make()
    out = torch.empty(B, C, H * H, device=dev)
    t = [timed(lambda: _native.channel_mix2(x1, x2, w, b, act_in=True, project=(w2, b2))),
         timed(lambda: _native.channel_mix2(gy, None, wt, None, transpose_w=True, dgelu_of=x1, out=out)),
         timed(lambda: _native.channel_wgrad2(gy, x1, x2, act_x=True)),
         timed(lambda: out.copy_(x1))]
    print(f"{tag:34s} fwd+proj {t[0]:6.1f}  igrad+dgelu {t[1]:6.1f}  wgrad {t[2]:6.1f}  copy {t[3]:6.1f} us", flush=True)


def rnd(scale=1.0):
    return lambda: torch.randn(B, C, H * H, device=dev) * scale


def domain_only():
    t = torch.randn(B, C, H, H, device=dev)
    t[:, :, 421:] = 0
    t[:, :, :, 421:] = 0
    return t.view(B, C, -1)


print(clocks())
case("random N(0, 1)", rnd())
case("random, zero outside 421^2", domain_only)
case("random N(0, 1e-3)", rnd(1e-3))
case("all zero", lambda: torch.zeros(B, C, H * H, device=dev))
case("constant 1", lambda: torch.ones(B, C, H * H, device=dev))
case("random N(0, 1) again", rnd())
print(clocks())
