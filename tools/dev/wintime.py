"""fc1-shaped calls (B = 16, 64 + 64 -> 64 channels) dense against windowed at several pitches: python tools/dev/wintime.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uno_amd import _native
dev = torch.device("cuda:0")
B, C = 16, 64
torch.manual_seed(0)
w = (torch.randn(64, 128) / 11).to(dev); b = torch.randn(64).to(dev); w2 = torch.randn(64).to(dev); b2 = torch.randn(1).to(dev)
wt = w[:, :64].contiguous()


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(tag, plane, win):
    x1 = torch.randn(B, C, plane, device=dev); x2 = torch.randn(B, C, plane, device=dev)
    gy = torch.randn(B, C, plane, device=dev); gout = torch.randn(B, plane, device=dev)
    out = torch.empty(B, C, plane, device=dev)
    t = [timed(lambda: _native.channel_mix2(x1, x2, w, b, act_in=True, project=(w2, b2), window=win)),
         timed(lambda: _native.gelu_project_backward(x1, w2, gout, window=win)),
         timed(lambda: _native.channel_mix2(gy, None, wt, None, transpose_w=True, dgelu_of=x1, out=out, window=win)),
         timed(lambda: _native.channel_wgrad2(gy, x1, x2, act_x=True, window=win))]
    P = win[0] * win[1] if win else plane
    print(f"{tag:34s} P={P:7d}  fwd+proj {t[0]:6.1f}  gelu_proj_bwd {t[1]:6.1f}  igrad+dgelu {t[2]:6.1f}  wgrad {t[3]:6.1f} us   per Mpx: " + " ".join(f"{v / P * 1e6 / 1e3:5.2f}" for v in t), flush=True)


case("dense 446^2", 446 * 446, None)
case("dense 421 x 424", 421 * 424, None)
case("window 421 x 424 of 446 x 446", 446 * 446, (421, 424, 446))
case("window 421 x 424 of 446 x 448", 446 * 448, (421, 424, 448))
case("window 421 x 444 of 446 x 446", 446 * 446, (421, 444, 446))
case("window 445 x 444 of 446 x 446", 446 * 446, (445, 444, 446))
case("window 421 x 448 of 446 x 448", 446 * 448, (421, 448, 448))
