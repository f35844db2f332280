"""bf16-activation forms of the block kernels K7 - K13 and of the fp16-weight per-mode GEMM (BASELINE.json configs[4]).

Contract of every `_bf16` entry point: it IS the float32 kernel on the widened inputs with one round-to-nearest-even on the way
out.  So each test runs the float32 kernel on `x.float()` and requires the bf16 kernel's output to equal `.bfloat16()` of that
result up to ONE bf16 ulp where the two kernels may order a float32 sum differently (tolerance 2^-8 relative L2 = 4e-3, i.e.
a fraction of an ulp on average), and float32 outputs (weight gradients, statistics) to agree to 1e-5.  pytest -m gpu"""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL_BF = 4e-3
TOL_F = 1e-5


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _bf(t):
    return t.to(torch.bfloat16)


def _cmp_bf(got, want_f32):
    assert got.dtype == torch.bfloat16
    assert rel_err(got.float().cpu().numpy(), _bf(want_f32).float().cpu().numpy()) < TOL_BF


@pytest.mark.parametrize("shape", [(2, 32, 64, 431 * 3), (1, 64, 128, 128 * 7), (3, 5, 7, 50), (2, 16, 64, 3), (1, 48, 130, 1000), (2, 128, 64, 515)])
@pytest.mark.parametrize("variant", ["plain", "transpose", "accumulate", "act_in", "dgelu"])
def test_channel_mix_bf16(shape, variant):
    from uno_amd import _native
    B, Ci, Co, P = shape
    g = torch.Generator().manual_seed(Ci * 7 + P)
    x = _bf(torch.randn(B, Ci, P, generator=g)).to(dev())
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).to(dev())
    b = torch.randn(Co, generator=g).to(dev())
    kw = {}
    wq = w
    if variant == "transpose":
        wq = w.t().contiguous()
        kw["transpose_w"] = True
    if variant == "act_in":
        kw["act_in"] = True
    pre = None
    if variant == "dgelu":
        pre = _bf(torch.randn(B, Co, P, generator=g)).to(dev())
    if variant == "accumulate":
        base = _bf(torch.randn(B, Co, P, generator=g)).to(dev())
        got = _native.channel_mix(x, wq, b, out=base.clone(), **kw)
        want = _native.channel_mix(x.float(), wq, b, out=base.float(), **kw)
    elif variant == "dgelu":
        got = _native.channel_mix(x, wq, None, dgelu_of=pre, **kw)
        want = _native.channel_mix(x.float(), wq, None, dgelu_of=pre.float(), **kw)
    else:
        got = _native.channel_mix(x, wq, b, **kw)
        want = _native.channel_mix(x.float(), wq, b, **kw)
    _cmp_bf(got, want)


@pytest.mark.parametrize("cfg", [(2, 64, 64, 64, 700), (2, 128, 128, 64, 515), (1, 16, 48, 128, 300)])
def test_two_source_forms_bf16(cfg):
    """uno_channel_mix2 / uno_channel_wgrad2 on bf16 activations == the f32 kernels on the widened inputs, rounded once."""
    from uno_amd import _native
    B, C1, C2, Co, P = cfg
    g = torch.Generator().manual_seed(C1 + C2 + P)
    x1, x2 = (_bf(torch.randn(B, c, P, generator=g)).to(dev()) for c in (C1, C2))
    w = (torch.randn(Co, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(dev())
    b = torch.randn(Co, generator=g).to(dev())
    for act in (False, True):
        _cmp_bf(_native.channel_mix2(x1, x2, w, b, act_in=act), _native.channel_mix2(x1.float(), x2.float(), w, b, act_in=act))
    y, a = _native.channel_mix2(x1, x2, w, b, y_act=True)
    yf, af = _native.channel_mix2(x1.float(), x2.float(), w, b, y_act=True)
    _cmp_bf(y, yf); _cmp_bf(a, af)
    if C1 % 64 == 0:
        gy = _bf(torch.randn(B, Co, P, generator=g)).to(dev())
        g1, g2 = _native.channel_mix2(gy, None, w, None, transpose_w=True, split_out=C1, dgelu_of=x1)
        f1, f2 = _native.channel_mix2(gy.float(), None, w, None, transpose_w=True, split_out=C1, dgelu_of=x1.float())
        _cmp_bf(g1, f1); _cmp_bf(g2, f2)
        gw, gb = _native.channel_wgrad2(gy, x1, x2, act_x=True)
        gw2, gb2 = _native.channel_wgrad2(gy.float(), x1.float(), x2.float(), act_x=True)
        assert rel_err(gw.cpu().numpy(), gw2.cpu().numpy()) < TOL_F and rel_err(gb.cpu().numpy(), gb2.cpu().numpy()) < TOL_F


@pytest.mark.parametrize("shape", [(2, 32, 64, 431 * 3), (3, 5, 7, 50), (1, 64, 64, 64 * 9 + 5), (2, 130, 20, 1000)])
@pytest.mark.parametrize("act_x", [False, True])
def test_channel_wgrad_bf16(shape, act_x):
    from uno_amd import _native
    B, Ci, Co, P = shape
    g = torch.Generator().manual_seed(Ci + P)
    x = _bf(torch.randn(B, Ci, P, generator=g)).to(dev())
    gy = _bf(torch.randn(B, Co, P, generator=g)).to(dev())
    gw, gb = _native.channel_wgrad(gy, x, need_bias=True, act_x=act_x)
    gw2, gb2 = _native.channel_wgrad(gy.float(), x.float(), need_bias=True, act_x=act_x)
    assert gw.dtype == torch.float32
    assert rel_err(gw.cpu().numpy(), gw2.cpu().numpy()) < TOL_F and rel_err(gb.cpu().numpy(), gb2.cpu().numpy()) < TOL_F


@pytest.mark.parametrize("cfg", [(6, 90, 90, 45, 45), (5, 45, 45, 90, 90), (3, 223, 223, 111, 111), (2, 111, 111, 223, 223), (1, 1089, 1089, 544, 544),
                                 (2, 30, 41, 30, 17)])
def test_resample_bf16(cfg):
    from uno_amd.resample import resample_adjoint, resample_forward
    n, H, W, Ho, Wo = cfg
    g = torch.Generator().manual_seed(H + Wo)
    x = _bf(torch.randn(n, 1, H, W, generator=g)).to(dev())
    _cmp_bf(resample_forward(x, Ho, Wo), resample_forward(x.float(), Ho, Wo))
    gy = _bf(torch.randn(n, 1, Ho, Wo, generator=g)).to(dev())
    _cmp_bf(resample_adjoint(gy, H, W), resample_adjoint(gy.float(), H, W))
    acc = _bf(torch.randn(n, 1, Ho, Wo, generator=g)).to(dev())
    _cmp_bf(resample_forward(x, Ho, Wo, out=acc.clone()), resample_forward(x.float(), Ho, Wo, out=acc.float()))


@pytest.mark.parametrize("shape", [(2, 8, 223 * 223), (1, 4, 1000), (3, 6, 7), (2, 16, 111 * 111 + 1)])
@pytest.mark.parametrize("gelu", [False, True])
def test_instnorm_bf16(shape, gelu):
    from uno_amd import _native
    B, C, N = shape
    g = torch.Generator().manual_seed(N)
    x = _bf(3.0 * torch.randn(B, C, N, generator=g) + 1.0).to(dev())
    gamma, beta = torch.randn(C, generator=g).to(dev()), torch.randn(C, generator=g).to(dev())
    y, mean, rstd = _native.instnorm_forward(x, gamma, beta, 1e-5, gelu)
    y2, mean2, rstd2 = _native.instnorm_forward(x.float(), gamma, beta, 1e-5, gelu)
    _cmp_bf(y, y2)
    assert rel_err(mean.cpu().numpy(), mean2.cpu().numpy()) < TOL_F and rel_err(rstd.cpu().numpy(), rstd2.cpu().numpy()) < TOL_F
    gy = _bf(torch.randn(B, C, N, generator=g)).to(dev())
    gx, s1, s2 = _native.instnorm_backward(x, gy, gamma, beta, mean, rstd, gelu)
    gx2, s12, s22 = _native.instnorm_backward(x.float(), gy.float(), gamma, beta, mean2, rstd2, gelu)
    _cmp_bf(gx, gx2)
    assert rel_err(s1.cpu().numpy(), s12.cpu().numpy()) < 1e-4 and rel_err(s2.cpu().numpy(), s22.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("shape", [(2, 64, 446 * 446), (32, 128, 64 * 64), (1, 5, 7), (3, 16, 1003)])
def test_gelu_project_bf16(shape):
    from uno_amd import _native
    B, C, P = shape
    g = torch.Generator().manual_seed(P)
    pre = _bf(torch.randn(B, C, P, generator=g)).to(dev())
    w, b = (torch.randn(C, generator=g) / C ** 0.5).to(dev()), torch.randn(1, generator=g).to(dev())
    _cmp_bf(_native.gelu_project_forward(pre, w, b), _native.gelu_project_forward(pre.float(), w, b))
    gout = _bf(torch.randn(B, P, generator=g)).to(dev())
    gpre, gw, gb = _native.gelu_project_backward(pre, w, gout)
    gpre2, gw2, gb2 = _native.gelu_project_backward(pre.float(), w, gout.float())
    _cmp_bf(gpre, gpre2)
    assert rel_err(gw.cpu().numpy(), gw2.cpu().numpy()) < 1e-4 and rel_err(gb.cpu().numpy(), gb2.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("cfg", [(3, 421, 421, 446, 446), (2, 33, 35, 40, 47), (1, 5, 3, 5, 3)])
def test_gelu_pad_bf16(cfg):
    from uno_amd import _native
    n, H, W, Hp, Wp = cfg
    g = torch.Generator().manual_seed(H)
    s = _bf(torch.randn(n, 2, H, W, generator=g)).to(dev())
    _cmp_bf(_native.gelu_pad(s, Hp, Wp), _native.gelu_pad(s.float(), Hp, Wp))
    gy = _bf(torch.randn(n, 2, Hp, Wp, generator=g)).to(dev())
    _cmp_bf(_native.gelu_pad_backward(s, gy), _native.gelu_pad_backward(s.float(), gy.float()))


@pytest.mark.parametrize("cfg", [(4, 6, 5, 2, 30), (16, 64, 64, 2, 400), (2, 200, 96, 2, 64), (3, 8, 8, 4, 100)])
@pytest.mark.parametrize("op", [0, 1])
def test_mode_mix_reads_fp16_weights(cfg, op):
    """K2 with the weights in (re, im) float16 storage == K2 on the widened weights, bit for bit (same arithmetic after the load)."""
    from uno_amd import _native
    B, Ci, Co, nc, Mc = cfg
    g = torch.Generator().manual_seed(Ci + Mc)
    cin = Ci if op == 0 else Co
    X = torch.randn(B, cin, nc, Mc, dtype=torch.cfloat, generator=g).to(dev())
    wh = [torch.view_as_real(0.3 * torch.randn(Ci, Co, Mc, dtype=torch.cfloat, generator=g)).half().to(dev()) for _ in range(nc)]
    ww = [torch.view_as_complex(w.float()) for w in wh]
    got = _native.mode_mix(X, wh, op)
    want = _native.mode_mix(X, ww, op)
    assert torch.equal(got, want)


def test_operator_block_bf16_matches_float32_block_on_rounded_input():
    """A whole OperatorBlock_2D (both branches, InstanceNorm + GELU) in mixed-precision mode against the float32 block fed the same
    (pre-rounded) input: outputs and input gradients within the accumulated bf16 roundings of the chain (each of the ~4 tensors
    between kernels is rounded once: 1e-2 relative L2), parameter gradients (accumulated in f32 from bf16 activations) within 2e-2."""
    from uno_amd.integral_operators import OperatorBlock_2D, enable_mixed_precision
    torch.manual_seed(5)
    for (Ci, Co, H, Ho, m, norm) in [(8, 12, 90, 45, 8, True), (12, 8, 45, 90, 8, False), (6, 6, 64, 64, 10, False)]:
        blk = OperatorBlock_2D(Ci, Co, Ho, Ho, m, m, Normalize=norm).to(dev())
        x = _bf(torch.randn(2, Ci, H, H)).to(dev())
        gy = _bf(torch.randn(2, Co, Ho, Ho)).to(dev())
        xf = x.float().requires_grad_(True)
        yf = blk(xf)
        yf.backward(gy.float())
        ref = {k: p.grad.clone() for k, p in blk.named_parameters()}
        blk.zero_grad()
        enable_mixed_precision(blk)
        xb = x.clone().requires_grad_(True)
        yb = blk(xb)
        assert yb.dtype == torch.bfloat16
        yb.backward(gy)
        assert xb.grad.dtype == torch.bfloat16
        assert rel_err(yb.float().detach().cpu().numpy(), yf.detach().cpu().numpy()) < 1e-2
        assert rel_err(xb.grad.float().cpu().numpy(), xf.grad.cpu().numpy()) < 2e-2
        gmax = max(float(v.norm()) for v in ref.values())
        for k, p in blk.named_parameters():
            assert p.grad.dtype == p.dtype
            if norm and k == "w.conv.bias":      # exactly-zero true gradient in front of an InstanceNorm: rounding residue only
                continue
            d = float((torch.view_as_real(p.grad) if p.is_complex() else p.grad).sub(torch.view_as_real(ref[k]) if p.is_complex() else ref[k]).norm())
            assert d <= 2e-2 * float(ref[k].norm()) + 1e-4 * gmax, k


def test_mixed_precision_training_step_tracks_float32():
    """UNO_9(3, 16, pad=5) at S = 128: three mixed-precision training steps against three float32 steps from the same init -
    losses within 2 % of each other and decreasing; every parameter receives a finite gradient of its own dtype."""
    from uno_amd.harness import DarcyTrainer, MixedDarcyTrainer, UNO_9, synthetic_darcy_batch
    a, u = synthetic_darcy_batch(4, 128, 3, dev())
    torch.manual_seed(0)
    mf = UNO_9(3, 16, pad=5).to(dev())
    torch.manual_seed(0)
    mm = UNO_9(3, 16, pad=5).to(dev())
    tf, tm = DarcyTrainer(mf, lr=1e-3, weight_decay=1e-3), MixedDarcyTrainer(mm, lr=1e-3, weight_decay=1e-3)
    lf = [float(tf.step(a, u)) for _ in range(3)]
    lm = [float(tm.step(a, u)) for _ in range(3)]
    assert all(np.isfinite(lm)) and lm[2] < lm[0]
    assert np.allclose(lm, lf, rtol=2e-2)
    for k, p in mm.named_parameters():
        assert p.grad is not None and p.grad.dtype == p.dtype and bool(torch.isfinite(torch.view_as_real(p.grad) if p.is_complex() else p.grad).all()), k
