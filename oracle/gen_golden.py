"""Generate golden vectors by importing the genuine reference (ashiq24/UNO).

Run ONLY in the build container, where the read-only reference checkout exists:

    python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]

The reference's Python files never enter this repository and never travel to the
GPU box; what is committed are the vectors (inputs, explicit weights, expected
outputs and gradients) this script writes as ``tests/golden/*.npz``.

Cases follow SURVEY.md section 8(c).  Every case stores its weights explicitly -
nothing depends on RNG parity between machines.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch


def _np(t):
    t = t.detach()
    return t.resolve_conj().cpu().numpy().copy()     # copy: later in-place updates must not alias


def spectral2d_case(io, name, B, Ci, Co, H, W, Ho, Wo, m1, m2, seed, noncontig=False, dtype=torch.float32,
                    ctor_dims=None):
    torch.manual_seed(seed)
    cd = ctor_dims or (Ho, Wo)
    mod = io.SpectralConv2d_Uno(Ci, Co, cd[0], cd[1], m1, m2)
    if noncontig:
        x = torch.randn(B, Ci, W, H, dtype=dtype).transpose(-1, -2)
    else:
        x = torch.randn(B, Ci, H, W, dtype=dtype)
    x.requires_grad_(True)
    y = mod(x, Ho, Wo) if ctor_dims else mod(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    return {
        f"{name}.x": _np(x), f"{name}.w1": _np(mod.weights1), f"{name}.w2": _np(mod.weights2),
        f"{name}.y": _np(y), f"{name}.gy": _np(gy), f"{name}.gx": _np(x.grad),
        f"{name}.gw1": _np(mod.weights1.grad), f"{name}.gw2": _np(mod.weights2.grad),
        f"{name}.meta": np.array([B, Ci, Co, H, W, Ho, Wo, m1, m2], dtype=np.int64),
    }


def spectral3d_case(io, name, B, Ci, Co, dims_in, dims_out, modes, seed):
    torch.manual_seed(seed)
    mod = io.SpectralConv3d_Uno(Ci, Co, *dims_out, *modes)
    x = torch.randn(B, Ci, *dims_in, requires_grad=True)
    y = mod(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    out = {f"{name}.x": _np(x), f"{name}.y": _np(y), f"{name}.gy": _np(gy), f"{name}.gx": _np(x.grad),
           f"{name}.meta": np.array([B, Ci, Co, *dims_in, *dims_out, *modes], dtype=np.int64)}
    for k in range(1, 5):
        w = getattr(mod, f"weights{k}")
        out[f"{name}.w{k}"] = _np(w)
        out[f"{name}.gw{k}"] = _np(w.grad)
    return out


def block2d_case(io, name, B, Ci, Co, H, W, Ho, Wo, m1, m2, normalize, non_lin, seed):
    torch.manual_seed(seed)
    blk = io.OperatorBlock_2D(Ci, Co, Ho, Wo, m1, m2, Normalize=normalize, Non_Lin=non_lin)
    if normalize:       # make the affine part non-trivial
        with torch.no_grad():
            blk.normalize_layer.weight.uniform_(0.5, 1.5)
            blk.normalize_layer.bias.uniform_(-0.5, 0.5)
    x = torch.randn(B, Ci, H, W, requires_grad=True)
    y = blk(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    out = {f"{name}.x": _np(x), f"{name}.y": _np(y), f"{name}.gy": _np(gy), f"{name}.gx": _np(x.grad),
           f"{name}.meta": np.array([B, Ci, Co, H, W, Ho, Wo, m1, m2, int(normalize), int(non_lin)], dtype=np.int64)}
    for k, v in blk.state_dict().items():
        out[f"{name}.sd.{k}"] = _np(v)
    for k, p in blk.named_parameters():
        out[f"{name}.grad.{k}"] = _np(p.grad)
    return out


def block3d_case(io, name, B, Ci, Co, dims_in, dims_out, modes, normalize, non_lin, seed):
    torch.manual_seed(seed)
    blk = io.OperatorBlock_3D(Ci, Co, *dims_out, *modes, Normalize=normalize, Non_Lin=non_lin)
    x = torch.randn(B, Ci, *dims_in, requires_grad=True)
    y = blk(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    out = {f"{name}.x": _np(x), f"{name}.y": _np(y), f"{name}.gy": _np(gy), f"{name}.gx": _np(x.grad),
           f"{name}.meta": np.array([B, Ci, Co, *dims_in, *dims_out, *modes, int(normalize), int(non_lin)],
                                    dtype=np.int64)}
    for k, v in blk.state_dict().items():
        out[f"{name}.sd.{k}"] = _np(v)
    for k, p in blk.named_parameters():
        out[f"{name}.grad.{k}"] = _np(p.grad)
    return out


def pointwise3d_case(io, name, B, Ci, Co, dims_in, dims_out, seed):
    torch.manual_seed(seed)
    pw = io.pointwise_op_3D(Ci, Co, *dims_out)
    x = torch.randn(B, Ci, *dims_in)
    y = pw(x)
    return {f"{name}.x": _np(x), f"{name}.y": _np(y), f"{name}.weight": _np(pw.conv.weight),
            f"{name}.bias": _np(pw.conv.bias),
            f"{name}.meta": np.array([B, Ci, Co, *dims_in, *dims_out], dtype=np.int64)}


def dim_mutation_case(io):
    """forward(x, d1, d2) rewrites SpectralConv.dim* but not pointwise_op.dim*
    (integral_operators.py:182-184 vs :229-231)."""
    torch.manual_seed(77)
    blk = io.OperatorBlock_2D(3, 4, 16, 16, 4, 4)
    x = torch.randn(2, 3, 20, 20)
    y_override = blk(x, 12, 12)
    state = np.array([blk.conv.dim1, blk.conv.dim2, blk.w.dim1, blk.w.dim2], dtype=np.int64)
    y_conv_after = blk.conv(x)          # uses the mutated dims
    out = {"dimmut.x": _np(x), "dimmut.y_override": _np(y_override), "dimmut.state": state,
           "dimmut.y_conv_after": _np(y_conv_after)}
    for k, v in blk.state_dict().items():
        out[f"dimmut.sd.{k}"] = _np(v)
    return out


def uno9_case(ref_dir):
    """UNO_9(3, 4, pad=5) at S=72 (smallest grid on which the class's hard-coded
    modes 18/8 are legal): output, loss, all grads, params after 3 reference-Adam
    steps with weight_decay=1e-3 (train_darcy.py:35-56)."""
    import darcy_flow_uno2d as d2
    from Adam import Adam
    from utilities3 import LpLoss

    torch.manual_seed(5)
    S, B, width = 72, 2, 4
    model = d2.UNO_9(3, width, pad=5)
    a = torch.rand(B, S, S, 1)
    u = torch.rand(B, S, S)
    out = {"uno9.a": _np(a), "uno9.u": _np(u), "uno9.meta": np.array([S, B, width, 5], dtype=np.int64)}
    for k, v in model.state_dict().items():
        out[f"uno9.sd.{k}"] = _np(v)
    loss_fn = LpLoss(size_average=False)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-3, amsgrad=False)
    losses = []
    for step in range(3):
        opt.zero_grad()
        pred = model(a).reshape(B, S, S)
        loss = loss_fn(pred.view(B, -1), u.view(B, -1))
        loss.backward()
        if step == 0:
            out["uno9.pred0"] = _np(pred)
            for k, p in model.named_parameters():
                g = p.grad
                out[f"uno9.gradnorm.{k}"] = np.array(float(torch.linalg.vector_norm(g)))
                if g.numel() <= 4096:
                    out[f"uno9.grad.{k}"] = _np(g)
        losses.append(float(loss))
        opt.step()
    out["uno9.losses"] = np.array(losses)
    for k, p in model.named_parameters():
        out[f"uno9.after3.norm.{k}"] = np.array(float(torch.linalg.vector_norm(p)))
        out[f"uno9.after3.sum.{k}"] = _np(p.sum().reshape(1))
        if p.numel() <= 4096:
            out[f"uno9.after3.{k}"] = _np(p)
    return out


def _checksums(model):
    out = {}
    for k, p in model.named_parameters():
        out[k] = np.array([float(p.detach().abs().sum()), float(torch.linalg.vector_norm(p.detach()))])
    return out


def ns2d_case():
    """UNO(14, 4) (navier_stokes_uno2d.py:145-238), S=64, 2 autoregressive steps (ns_train_2d.py:46-67): loss and
    gradient norms.  Weights are NOT stored (1.9 MB): they come from torch.manual_seed(21) + constructor order,
    verified in the test through per-parameter checksums."""
    import navier_stokes_uno2d as n2
    from utilities3 import LpLoss
    torch.manual_seed(21)
    model = n2.UNO(14, 4)
    out = {f"ns2d.ck.{k}": v for k, v in _checksums(model).items()}
    g = torch.Generator().manual_seed(22)
    xx = torch.randn(1, 64, 64, 10, generator=g)
    yy = torch.randn(1, 64, 64, 2, generator=g)
    out["ns2d.xx"], out["ns2d.yy"] = _np(xx), _np(yy)
    myloss = LpLoss(size_average=False)
    loss = 0
    x = xx
    preds = []
    for t in range(2):
        im = model(x)
        preds.append(im)
        loss = loss + myloss(im.reshape(1, -1), yy[..., t:t + 1].reshape(1, -1))
        x = torch.cat((x[..., 1:], im), dim=-1)
    loss.backward()
    out["ns2d.loss"] = np.array(float(loss))
    out["ns2d.pred"] = _np(torch.cat(preds, -1))
    for k, p in model.named_parameters():
        out[f"ns2d.gradnorm.{k}"] = np.array(float(torch.linalg.vector_norm(p.grad)))
    return out


def ns3d_case():
    """Uno3D_T20(6, 2, pad=3) (navier_stokes_uno3d.py:239-409), S=32, T 10 -> 20 (ns_train_3d.py:53-65).  Seeded
    weights (torch.manual_seed(31)) verified by checksums; input/target stored."""
    import navier_stokes_uno3d as n3
    from utilities3 import LpLoss
    torch.manual_seed(31)
    model = n3.Uno3D_T20(6, 2, pad=3)
    out = {f"ns3d.ck.{k}": v for k, v in _checksums(model).items()}
    g = torch.Generator().manual_seed(32)
    x = torch.randn(1, 32, 32, 10, 1, generator=g)
    y = torch.randn(1, 32, 32, 20, generator=g)
    out["ns3d.x"], out["ns3d.y"] = _np(x), _np(y)
    pred = model(x).view(1, 32, 32, 20)
    loss = LpLoss(size_average=False)(pred.reshape(1, -1), y.reshape(1, -1))
    loss.backward()
    out["ns3d.loss"] = np.array(float(loss))
    out["ns3d.pred"] = _np(pred)
    for k, p in model.named_parameters():
        out[f"ns3d.gradnorm.{k}"] = np.array(float(torch.linalg.vector_norm(p.grad)))
    return out


def adam_case():
    """3 reference-Adam steps on a complex and a real tensor with fixed grads."""
    from Adam import Adam
    torch.manual_seed(9)
    pc = torch.nn.Parameter(torch.randn(3, 5, dtype=torch.cfloat))
    pr = torch.nn.Parameter(torch.randn(7))
    out = {"adam.pc0": _np(pc), "adam.pr0": _np(pr)}
    opt = Adam([pc, pr], lr=1e-2, weight_decay=1e-3)
    gcs, grs = [], []
    for _ in range(3):
        gc_, gr_ = torch.randn(3, 5, dtype=torch.cfloat), torch.randn(7)
        gcs.append(_np(gc_))
        grs.append(_np(gr_))
        pc.grad, pr.grad = gc_.clone(), gr_.clone()
        opt.step()
    out.update({"adam.gc": np.stack(gcs), "adam.gr": np.stack(grs), "adam.pc3": _np(pc), "adam.pr3": _np(pr)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "tests", "golden"))
    args = ap.parse_args()
    if not os.path.isdir(args.ref):
        sys.exit(f"reference checkout not found at {args.ref}; golden vectors can only be regenerated in the build container")
    sys.path.insert(0, args.ref)
    os.environ.setdefault("MPLBACKEND", "Agg")
    cwd = os.getcwd()
    os.chdir("/tmp")
    import integral_operators as io     # the genuine reference module
    os.chdir(cwd)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    os.makedirs(args.out, exist_ok=True)

    s2 = {}
    # (name, B,Ci,Co, H,W, Ho,Wo, m1,m2, seed)
    for row in [
        ("contract_mixed", 2, 3, 4, 21, 18, 13, 10, 4, 5, 101),
        ("expand", 2, 3, 2, 16, 16, 32, 32, 5, 7, 102),
        ("nyquist_full_rows", 2, 2, 3, 12, 12, 8, 8, 4, 5, 103),
        ("identity_odd", 1, 4, 4, 15, 15, 15, 15, 3, 3, 104),
        ("overlap_out", 2, 3, 3, 20, 20, 10, 10, 8, 6, 105),
        ("prime", 2, 2, 2, 23, 23, 11, 11, 4, 4, 106),
        ("overlap_in", 1, 2, 2, 10, 14, 16, 16, 7, 5, 107),
        ("wide_modes", 1, 5, 6, 40, 44, 36, 40, 17, 20, 108),
    ]:
        s2.update(spectral2d_case(io, *row))
    s2.update(spectral2d_case(io, "noncontig", 2, 3, 3, 14, 18, 12, 12, 3, 4, 109, noncontig=True))
    # float64 input is NOT a supported reference behaviour on torch >= 2: the einsum at
    # integral_operators.py:179 raises "expected scalar type ComplexDouble but found ComplexFloat".
    try:
        spectral2d_case(io, "fp64_in", 1, 2, 2, 12, 12, 12, 12, 3, 3, 110, dtype=torch.float64)
        s2["fp64_in.raises"] = np.array(0)
    except RuntimeError:
        s2["fp64_in.raises"] = np.array(1)
    s2.update(spectral2d_case(io, "dims_override", 1, 2, 3, 18, 18, 9, 9, 3, 3, 111, ctor_dims=(30, 30)))
    np.savez_compressed(os.path.join(args.out, "spectral2d.npz"), **s2)

    s3 = {}
    s3.update(spectral3d_case(io, "basic", 1, 2, 3, (16, 16, 10), (12, 12, 16), (4, 4, 3), 201))
    s3.update(spectral3d_case(io, "overlap_T40conv3", 1, 2, 2, (16, 16, 20), (8, 8, 20), (6, 6, 7), 202))
    s3.update(spectral3d_case(io, "odd_time", 2, 2, 2, (10, 12, 13), (10, 12, 15), (3, 4, 5), 203))
    np.savez_compressed(os.path.join(args.out, "spectral3d.npz"), **s3)

    blk = {}
    k = 0
    for normalize in (False, True):
        for non_lin in (False, True):
            blk.update(block2d_case(io, f"b2d_n{int(normalize)}_g{int(non_lin)}", 2, 3, 4, 18, 20, 12, 14, 4, 5,
                                    normalize, non_lin, 300 + k))
            k += 1
    blk.update(block3d_case(io, "b3d_n1_g1", 1, 2, 3, (12, 12, 8), (8, 8, 10), (3, 3, 3), True, True, 310))
    blk.update(block3d_case(io, "b3d_n0_g0", 1, 2, 2, (8, 8, 6), (8, 8, 6), (3, 3, 2), False, False, 311))
    blk.update(pointwise3d_case(io, "pw3d_shrink", 1, 2, 2, (12, 10, 8), (8, 6, 6), 320))
    blk.update(pointwise3d_case(io, "pw3d_grow", 1, 2, 2, (8, 8, 6), (12, 10, 10), 321))
    blk.update(dim_mutation_case(io))
    np.savez_compressed(os.path.join(args.out, "blocks.npz"), **blk)

    har = {}
    har.update(uno9_case(args.ref))
    har.update(adam_case())
    np.savez_compressed(os.path.join(args.out, "harness.npz"), **har)
    ns = {}
    ns.update(ns2d_case())
    ns.update(ns3d_case())
    np.savez_compressed(os.path.join(args.out, "harness_ns.npz"), **ns)

    for f in sorted(os.listdir(args.out)):
        print(f, os.path.getsize(os.path.join(args.out, f)))


if __name__ == "__main__":
    main()
