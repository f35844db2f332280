"""Operator blocks and the UNO_9 harness model on the MI355X (HIP spectral path + stock ROCm ops for
the point-wise branch) against reference-generated golden vectors.  pytest -m gpu

Tolerances: blocks 2e-5 (spectral part) .. 1e-4 (after bicubic-AA + InstanceNorm + GELU on a
different backend); end-to-end model 1e-3 on outputs/gradient norms (5 blocks + 2 InstanceNorms)."""
import numpy as np
import pytest
import torch

from conftest import Case, load_cases, rel_err

pytestmark = pytest.mark.gpu
ZB, NAMESB = load_cases("blocks.npz")
ZH, _ = load_cases("harness.npz")
B2D = [n for n in NAMESB if n.startswith("b2d_")]


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", B2D)
def test_operator_block_2d_golden(name):
    from uno_amd.integral_operators import OperatorBlock_2D
    c = Case(ZB, name)
    B, Ci, Co, H, W, Ho, Wo, m1, m2, nrm, nl = [int(v) for v in c.meta]
    blk = OperatorBlock_2D(Ci, Co, Ho, Wo, m1, m2, Normalize=bool(nrm), Non_Lin=bool(nl))
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in c.sub("sd").items()}, strict=True)
    blk = blk.to(dev())
    x = torch.from_numpy(c.x).to(dev()).requires_grad_(True)
    y = blk(x)
    assert rel_err(y.detach().cpu().numpy(), c.y) < 1e-4
    y.backward(torch.from_numpy(c.gy).to(dev()))
    assert rel_err(x.grad.cpu().numpy(), c.gx) < 1e-4
    params = dict(blk.named_parameters())
    floor = 1e-5 * float(np.linalg.norm(c.gy))     # w.conv.bias has an exactly-zero true gradient under InstanceNorm
    for k, g in c.sub("grad").items():
        got = params[k].grad.cpu().numpy()
        assert np.linalg.norm((got - g).ravel()) <= 2e-4 * np.linalg.norm(g.ravel()) + floor, k


@pytest.mark.parametrize("name", ["pw3d_shrink", "pw3d_grow"])
def test_pointwise_op_3d_golden(name):
    """The product pointwise_op_3D (K8 1x1x1 convolution + bug-compatible FFT crop / resample) against the reference's output:
    `pw3d_grow` runs on the pruned-DFT kernels (uno_fft_resample3d), `pw3d_shrink` has an odd kept-row count and takes the
    stock-FFT branch of the same module - both must reproduce reference integral_operators.py:448-467."""
    from uno_amd.integral_operators import pointwise_op_3D, _resample3d_plan
    c = Case(ZB, name)
    B, Ci, Co, *dims = [int(v) for v in c.meta]
    din, dout = tuple(dims[:3]), tuple(dims[3:])
    pw = pointwise_op_3D(Ci, Co, *dout)
    with torch.no_grad():
        pw.conv.weight.copy_(torch.from_numpy(c.weight))
        pw.conv.bias.copy_(torch.from_numpy(c.bias))
    pw = pw.to(dev())
    import uno_amd.integral_operators as io
    assert (_resample3d_plan(din, dout, dev()) is not None) == (name == "pw3d_grow")
    x = torch.from_numpy(c.x).to(dev()).requires_grad_(True)
    if name == "pw3d_shrink":
        # outside the kernels' range the layer raises (no silent stock-library dispatch) until the caller allows torch.fft
        with pytest.raises(RuntimeError, match="STOCK_FFT_RESAMPLE3D"):
            pw(x)
        io.STOCK_FFT_RESAMPLE3D = True
    try:
        y = pw(x)
        assert tuple(y.shape) == tuple(c.y.shape)
        assert rel_err(y.detach().cpu().numpy(), c.y) < 1e-4
        # adjoint identity of the (linear) operator: <pw(x) - pw(0), g> == <x, pw^T g>
        g = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, g)
        lhs = float(((y - pw(torch.zeros_like(x))).detach() * g).double().sum())
        rhs = float((x.detach() * gx).double().sum())
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1e-6)
    finally:
        io.STOCK_FFT_RESAMPLE3D = False


def test_dim_mutation_quirk():
    from uno_amd.integral_operators import OperatorBlock_2D
    c = Case(ZB, "dimmut")
    blk = OperatorBlock_2D(3, 4, 16, 16, 4, 4)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in c.sub("sd").items()}, strict=True)
    blk = blk.to(dev())
    x = torch.from_numpy(c.x).to(dev())
    y = blk(x, 12, 12)
    assert rel_err(y.detach().cpu().numpy(), c.y_override) < 1e-4
    assert [blk.conv.dim1, blk.conv.dim2, blk.w.dim1, blk.w.dim2] == [int(v) for v in c.state]
    assert rel_err(blk.conv(x).detach().cpu().numpy(), c.y_conv_after) < 2e-5


def test_uno9_training_steps_match_reference():
    """UNO_9(3,4,pad=5), S=72: prediction, loss, gradient norms and 3 reference-Adam steps."""
    from uno_amd.harness import DarcyTrainer, UNO_9
    c = Case(ZH, "uno9")
    S, B, width, pad = [int(v) for v in c.meta]
    model = UNO_9(3, width, pad=pad)
    sd0 = {k: v.copy() for k, v in c.sub("sd").items()}
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd0.items()}, strict=True)
    model = model.to(dev())
    a, u = torch.from_numpy(c.a).to(dev()), torch.from_numpy(c.u).to(dev())
    with torch.no_grad():
        pred = model(a).reshape(B, S, S)
    assert rel_err(pred.cpu().numpy(), c.pred0) < 1e-4
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    losses = []
    gmax = max(float(getattr(c, f"gradnorm.{k}")) for k, _ in model.named_parameters())
    # a convolution bias in front of an InstanceNorm has an exactly-zero true gradient: what the reference stores for it is
    # rounding residue, and Adam turns that residue into +-lr steps of arbitrary sign
    noise = {"conv1.w.conv.bias", "conv4.w.conv.bias"}
    for step in range(3):
        losses.append(float(tr.step(a, u)))
        if step == 0:
            for k, p in model.named_parameters():
                ref = float(getattr(c, f"gradnorm.{k}"))
                assert abs(float(torch.linalg.vector_norm(p.grad)) - ref) <= 2e-4 * ref + 1e-6 * gmax, k
            # element-wise gradients of every small tensor the reference stored (oracle/gen_golden.py: numel <= 4096)
            params = dict(model.named_parameters())
            for k, g in c.sub("grad").items():
                got = params[k].grad.cpu().numpy()
                assert np.linalg.norm((got - g).ravel()) <= 2e-4 * np.linalg.norm(g.ravel()) + 1e-6 * gmax, k
    assert np.allclose(losses, c.losses, rtol=2e-4)
    params = dict(model.named_parameters())
    for k, p in model.named_parameters():
        ref = float(getattr(c, f"after3.norm.{k}"))
        assert abs(float(torch.linalg.vector_norm(p)) - ref) <= 1e-4 * ref + 1e-8, k
    # the UPDATE itself, element by element: a parameter moves ~3e-3 in three Adam steps, so compare (after - before)
    checked = 0
    for k, v in c.sub("after3").items():
        if k.startswith("norm.") or k.startswith("sum.") or k in noise:
            continue
        got = params[k].detach().cpu().numpy()
        d_ref, d_got = v - sd0[k], got - sd0[k]
        assert np.linalg.norm((d_got - d_ref).ravel()) <= 2e-2 * np.linalg.norm(d_ref.ravel()), k
        checked += 1
    assert checked >= 15


def test_uno9_reference_style_caller_matches_golden():
    """The reference's OWN calling convention (channels-last nn.Linear, F.gelu, permute, F.pad, torch.cat, positional block
    calls - darcy_flow_uno2d.py:94-133, restated in tools/reference_style_caller.py) on the product blocks: prediction, loss and
    every stored gradient against the reference-generated golden.  This is the drop-in path a user of the reference gets."""
    from tools.reference_style_caller import UNO_9_ReferenceStyle
    from uno_amd.harness import lp_loss_rel_sum
    c = Case(ZH, "uno9")
    S, B, width, pad = [int(v) for v in c.meta]
    model = UNO_9_ReferenceStyle(3, width, pad=pad)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in c.sub("sd").items()}, strict=True)
    model = model.to(dev())
    a, u = torch.from_numpy(c.a).to(dev()), torch.from_numpy(c.u).to(dev())
    pred = model(a).reshape(B, S, S)
    assert rel_err(pred.detach().cpu().numpy(), c.pred0) < 1e-4
    loss = lp_loss_rel_sum(pred.view(B, -1), u.view(B, -1))
    assert abs(float(loss) - float(c.losses[0])) < 2e-4 * abs(float(c.losses[0]))
    loss.backward()
    gmax = max(float(getattr(c, f"gradnorm.{k}")) for k, _ in model.named_parameters())
    params = dict(model.named_parameters())
    for k, p in params.items():
        ref = float(getattr(c, f"gradnorm.{k}"))
        assert abs(float(torch.linalg.vector_norm(p.grad)) - ref) <= 2e-4 * ref + 1e-6 * gmax, k
    for k, g in c.sub("grad").items():
        got = params[k].grad.cpu().numpy()
        assert np.linalg.norm((got - g).ravel()) <= 2e-4 * np.linalg.norm(g.ravel()) + 1e-6 * gmax, k


def test_model_product_vs_oracle_blocks_same_weights():
    """Same UNO_9 weights, product blocks on the GPU vs the oracle's FFT blocks on the host, at a
    non-golden size (S=100 -> padded 110, scale=2)."""
    from oracle import spectral_oracle as so
    from uno_amd.harness import UNO_9, lp_loss_rel_sum, synthetic_darcy_batch
    torch.manual_seed(3)
    ref = UNO_9(3, 8, pad=5, block_cls=so.OracleOperatorBlock2d)
    prod = UNO_9(3, 8, pad=5)
    prod.load_state_dict(ref.state_dict(), strict=True)
    prod = prod.to(dev())
    a, u = synthetic_darcy_batch(2, 100, 11, "cpu")
    lr = lp_loss_rel_sum(ref(a).reshape(2, -1), u.reshape(2, -1))
    lr.backward()
    lp = lp_loss_rel_sum(prod(a.to(dev())).reshape(2, -1), u.to(dev()).reshape(2, -1))
    lp.backward()
    assert abs(float(lp) - float(lr)) < 1e-4 * abs(float(lr))
    # Gradient tolerance: 2e-2 while the InstanceNorm layers ran on MIOpen (3e-4 fwd / 1e-3 bwd off the CPU op on odd
    # grids such as 55x55 / 27x27); with the K13 kernel every operator of the model is within f32 rounding of the
    # reference and the whole-model gradients agree to <= 3e-6 (measured); asserted at 2e-5.
    pr = dict(ref.named_parameters())
    gmax = max(float(torch.linalg.vector_norm(q.grad)) for q in pr.values())
    for k, p in prod.named_parameters():
        if k in ("conv1.w.conv.bias", "conv4.w.conv.bias"):
            continue    # in front of an InstanceNorm: true gradient is exactly zero, both sides hold only rounding residue
        g, gr = p.grad.cpu(), pr[k].grad
        n = float(torch.linalg.vector_norm(gr))
        err = float(torch.linalg.vector_norm(g - gr))
        print(f"{k:32s} rel grad err {err / max(n, 1e-30):.2e}")
        assert err <= 2e-5 * n + 1e-6 * gmax, k


@pytest.mark.parametrize("hw,out_hw", [((40, 36), (20, 18)), ((24, 28), (24, 28)), ((20, 18), (41, 37))])
@pytest.mark.parametrize("normalize", [False, True])
def test_fused_block_equals_branch_sum(hw, out_hw, normalize):
    """The one-buffer block (point-wise branch accumulating into the spectral branch's output, and likewise for
    grad_x) against the plain sum of the two branch modules - down-sampling, same-size and up-sampling blocks."""
    from uno_amd.integral_operators import OperatorBlock_2D
    torch.manual_seed(3)
    blk = OperatorBlock_2D(6, 10, out_hw[0], out_hw[1], 5, 4, Normalize=normalize).cuda()
    x = torch.randn(3, 6, *hw, device="cuda", requires_grad=True)
    params = [p for p in blk.parameters()]

    y = blk(x)
    gy = torch.randn_like(y)
    got = torch.autograd.grad(y, [x] + params, gy)

    s = blk.conv(x) + blk.w(x)                 # unfused composition of the same kernels
    if normalize:                              # in float64: torch's own InstanceNorm on ROCm (MIOpen) is 3e-4 / 1e-3 off on odd grids
        nl = blk.normalize_layer
        s = torch.nn.functional.instance_norm(s.double(), weight=nl.weight.double(), bias=nl.bias.double(), eps=nl.eps)
    y2 = torch.nn.functional.gelu(s).float()
    ref = torch.autograd.grad(y2, [x] + params, gy)

    assert ((y - y2).abs().max() / y2.abs().max()).item() < 2e-6
    # gradients: relative to the tensor's own scale, with a floor for tensors whose true gradient is zero (the 1x1
    # convolution's bias in front of an InstanceNorm: both sides are rounding noise there)
    gmax = max(float(r.abs().max()) for r in ref)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 2e-6 * gmax


@pytest.mark.parametrize("hw,out_hw", [((40, 36), (20, 18)), ((24, 28), (24, 28)), ((20, 18), (41, 37))])
@pytest.mark.parametrize("normalize", [False, True])
def test_forward_cat_equals_block_of_concatenation(hw, out_hw, normalize):
    """Two-source block (skip connection consumed without torch.cat) against the same block on the concatenated
    tensor: output and every gradient, for down-sampling, same-size and up-sampling blocks."""
    from uno_amd.integral_operators import OperatorBlock_2D
    torch.manual_seed(5)
    blk = OperatorBlock_2D(4 + 7, 9, out_hw[0], out_hw[1], 5, 4, Normalize=normalize).cuda()
    a = torch.randn(3, 4, *hw, device="cuda", requires_grad=True)
    b = torch.randn(3, 7, *hw, device="cuda", requires_grad=True)
    params = list(blk.parameters())
    y = blk.forward_cat([a, b])
    gy = torch.randn_like(y)
    got = torch.autograd.grad(y, [a, b] + params, gy)
    y2 = blk(torch.cat([a, b], dim=1))
    ref = torch.autograd.grad(y2, [a, b] + params, gy)
    assert ((y - y2).abs().max() / y2.abs().max()).item() < 2e-6
    gmax = max(float(r.abs().max()) for r in ref)
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        assert float((g - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 2e-6 * gmax
    # call-time grid override goes through the same path and persists on the spectral layer, as in forward()
    y3 = blk.forward_cat([a, b], out_hw[0] + 2, out_hw[1] + 1)
    assert y3.shape[-2:] == (out_hw[0] + 2, out_hw[1] + 1) and (blk.conv.dim1, blk.conv.dim2) == (out_hw[0] + 2, out_hw[1] + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(24, 24, 5), (24, 12, 4), (12, 24, 4)], ids=["same", "down", "up"])
def test_grad_join_equals_the_summed_gradients(geom):
    """A tensor with two consumers (skip connection): with a GradJoin the later consumer (two-source block / two-source channel mix)
    leaves its contribution - spectrum + accumulating closures - to the first consumer, which returns the complete gradient.
    Same gradients as autograd's sum of two gradient tensors (all three resampling directions of the deferring block, different
    mode counts in the two blocks)."""
    from uno_amd.integral_operators import GradJoin, OperatorBlock_2D, channel_mix_cat
    torch.manual_seed(3)
    S, So, mB = geom
    B, C = 2, 64
    blkA = OperatorBlock_2D(C, 32, 16, 16, 3, 3, Normalize=True).to(dev())            # first consumer of x (modes 3)
    blkB = OperatorBlock_2D(2 * C, 64, So, So, mB, mB).to(dev())                      # later consumer: block on cat([z, x]) (modes mB)
    fc = torch.nn.Linear(64 + C, 64).to(dev())                                        # later consumer: channel mix on cat([u, x])
    x0 = torch.randn(B, C, S, S, device=dev())
    z0 = torch.randn(B, C, S, S, device=dev())
    u0 = torch.randn(B, 64, S, S, device=dev())
    res = {}
    for mode in ("plain", "join"):
        x, z, u = (t.clone().requires_grad_(True) for t in (x0, z0, u0))
        for m in (blkA, blkB, fc):
            m.zero_grad(set_to_none=True)
        xa = x * 1.0                                                                    # x itself is a leaf: the joined tensor is xa
        jb, jf = (GradJoin(), GradJoin()) if mode == "join" else (None, None)
        a1 = blkA(xa, 16, 16, join=jb)
        b = blkB.forward_cat([z, xa], So, So, defer_gelu=True, defer_grad=jb)
        loss = a1.square().sum() + b.sin().sum()
        xb = x * 2.0
        a3 = blkA(xb, 16, 16, join=jf)
        c = channel_mix_cat([u, xb], fc.weight, fc.bias, gelu_first=True, defer_grad=jf)
        loss = loss + a3.cos().sum() + c.square().sum()
        loss.backward()
        res[mode] = [x.grad.clone(), z.grad.clone(), u.grad.clone()] + [p.grad.clone() for m in (blkA, blkB, fc) for p in m.parameters()]
    for a, b in zip(res["plain"], res["join"]):
        ar, br = (torch.view_as_real(t) if t.is_complex() else t for t in (a, b))
        assert float((ar - br).norm()) <= 2e-5 * float(ar.norm()) + 1e-12


@pytest.mark.gpu
def test_weight_gradients_are_written_in_place():
    """(a) with a FlatGradients buffer the weight-gradient kernels write each gradient into its view and autograd adopts an alias
    of it (no copy, no add); (b) a layer used several times in one graph (roll-out) accumulates into .grad in place; both give the
    gradients of the ordinary path (fresh tensors + autograd sums)."""
    import uno_amd.integral_operators as io
    from uno_amd.harness.train import FlatGradients
    from uno_amd.integral_operators import OperatorBlock_2D, channel_mix, gelu_channel_mix
    torch.manual_seed(1)
    blk = OperatorBlock_2D(8, 8, 20, 20, 4, 4).to(dev())
    lin = torch.nn.Linear(8, 8).to(dev())
    params = list(blk.parameters()) + list(lin.parameters())
    x = torch.randn(2, 8, 20, 20, device=dev())

    def loss_fn():
        h = x
        for _ in range(3):                                     # the same layers three times: 3 contributions per parameter
            h = blk(h)
            h = gelu_channel_mix(channel_mix(h, lin.weight, lin.bias), lin.weight, lin.bias)
        return h.square().sum()

    ref = {}
    for mode in (False, True):
        io.INPLACE_PARAM_GRADS = mode
        try:
            for p in params:
                p.grad = None
                if hasattr(p, "_uno_grad_buffer"):
                    del p._uno_grad_buffer
            loss_fn().backward()
            got = [p.grad.clone() for p in params]
            if not mode:
                ref = got
            else:
                for a, b in zip(ref, got):
                    ar, br = (torch.view_as_real(t) if t.is_complex() else t for t in (a, b))
                    assert float((ar - br).norm()) <= 2e-6 * float(ar.norm()) + 1e-12
        finally:
            io.INPLACE_PARAM_GRADS = True
    # (a) flat buffer: one use per parameter, every gradient lands in its view without a copy
    fg = FlatGradients(params)
    fg.zero_()
    blk(x).square().sum().backward()
    for p, v in zip(fg.params, fg.views):
        if p.grad is not None:
            assert p.grad.data_ptr() == v.data_ptr(), "gradient was not written into the flat buffer"
    single = [None if p.grad is None else p.grad.clone() for p in params]
    io.INPLACE_PARAM_GRADS = False
    try:
        for p in params:
            p.grad = None
        blk(x).square().sum().backward()
        for a, p in zip(single, params):
            if p.grad is not None:
                ar, br = (torch.view_as_real(t) if t.is_complex() else t for t in (a, p.grad))
                assert float((ar - br).norm()) <= 2e-6 * float(br.norm()) + 1e-12
    finally:
        io.INPLACE_PARAM_GRADS = True
        for p in params:
            if hasattr(p, "_uno_grad_buffer"):
                del p._uno_grad_buffer


@pytest.mark.gpu
def test_weight_gradient_batched_over_the_uses_of_a_layer():
    """A roll-out uses every spectral layer T times in one graph (reference ns_train_2d.py:46-68).  From the second backward pass
    on, the layer keeps its T truncated spectra in one stack and the last use to be back-propagated computes the weight gradient
    in ONE per-mode GEMM with K = T x batch.  Same gradients as T separate weight gradients summed by autograd; a changing number
    of uses, a second backward over a retained graph and a pass that reaches only some of the uses stay correct."""
    import warnings
    import uno_amd.integral_operators as io
    from uno_amd.harness.train import FlatGradients
    from uno_amd.integral_operators import OperatorBlock_2D, SpectralConv2d_Uno
    torch.manual_seed(3)
    blk = OperatorBlock_2D(8, 8, 24, 24, 5, 5).to(dev())
    conv = SpectralConv2d_Uno(8, 8, 24, 24, 4, 4).to(dev())
    params = list(blk.parameters()) + list(conv.parameters())
    x = torch.randn(3, 8, 24, 24, device=dev())

    def rollout(T, keep=None):
        h, outs = x, []
        for _ in range(T):
            h = torch.tanh(conv(blk(h)))
            outs.append(h)
        sel = outs if keep is None else [outs[i] for i in keep]
        return sum(o.square().sum() for o in sel)

    def grads(T, batched, keep=None):
        io.TIME_BATCHED_WGRAD = batched
        try:
            for p in params:
                p.grad = None
            rollout(T, keep).backward()
            return [p.grad.clone() for p in params]
        finally:
            io.TIME_BATCHED_WGRAD = True

    def same(a, b, tol=3e-6):
        for u, v in zip(a, b):
            ur, vr = (torch.view_as_real(t) if t.is_complex() else t for t in (u, v))
            assert float((ur - vr).norm()) <= tol * float(vr.norm()) + 1e-12

    w1 = blk.conv.weights1
    for p in (w1, conv.weights1):
        for a in ("_uno_uses", "_uno_stack", "_uno_nostack"):
            if hasattr(p, a):
                delattr(p, a)
    ref5 = grads(5, False)
    assert w1._uno_uses == 5                                    # counted by the unbatched pass
    got = grads(5, True)
    st = w1._uno_stack
    assert st.n == 5 and st.done and tuple(st.X.shape[:2]) == (5, 3), "the layer did not stack its five uses"
    assert st.Pinfo is not None and st.P is None, "the block's 1x1 convolution did not defer its second stage to the last use"
    same(got, ref5)
    # through a flat gradient buffer: the single GEMM writes into the parameter's view
    fg = FlatGradients(params)
    fg.zero_()
    rollout(5).backward()
    for p, v in zip(fg.params, fg.views):
        assert p.grad is not None and p.grad.data_ptr() == v.data_ptr()
    same([p.grad for p in params], ref5)
    for p in params:
        del p._uno_grad_buffer
    # fewer and more uses than the stack was sized for (7 = one full stack of 5 and a second one with 2)
    same(grads(3, True), grads(3, False))
    assert w1._uno_uses == 3
    ref7 = grads(7, False)
    w1._uno_uses = conv.weights1._uno_uses = 5
    same(grads(7, True), ref7)
    assert w1._uno_uses == 7                                    # every use of the pass counted: the next stack holds all seven
    same(grads(7, True), ref7)
    assert w1._uno_stack.n == 7
    # one layer on different grids within a pass: the uses share the spectral stack, the 1x1 convolution's partial sums only fit
    # the stack's first grid - the others are computed on their own (any order of fitting / non-fitting uses, also as the last one)
    def zigzag():
        h = x
        for d in (20, 16, 24, 20, 24, 16):
            h = torch.tanh(blk(h, d, d))
        return h.square().sum()
    for p in params:
        p.grad = None
    io.TIME_BATCHED_WGRAD = False
    try:
        zigzag().backward()
    finally:
        io.TIME_BATCHED_WGRAD = True
    refz = [None if p.grad is None else p.grad.clone() for p in params[:len(list(blk.parameters()))]]
    for rep in range(2):
        for p in params:
            p.grad = None
        zigzag().backward()
        same([p.grad for p in blk.parameters()], refz)
    assert w1._uno_stack.n == 6
    blk.conv.dim1 = blk.conv.dim2 = 24          # the call-time override sticks to the spectral layer (reference integral_operators.py:182-184)
    # retained graph: the second backward finds its stacks finished and runs use by use
    for p in params:
        p.grad = None
    loss = rollout(4)
    loss.backward(retain_graph=True)
    first = [p.grad.clone() for p in params]
    loss.backward()
    same([p.grad for p in params], [2 * g for g in first])
    # a pass that back-propagates only the first two of four uses: complete gradient, a warning, batching off for the layer
    ref_part = grads(4, False, keep=[1])
    w1._uno_uses = conv.weights1._uno_uses = 4
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        part = grads(4, True, keep=[1])
    same(part, ref_part)
    assert any("TIME_BATCHED_WGRAD" in str(w.message) for w in rec)
    assert w1._uno_nostack and w1._uno_uses == 0
    same(grads(4, True), grads(4, False))                       # and stays right afterwards
    for p in (w1, conv.weights1):
        for a in ("_uno_uses", "_uno_stack", "_uno_nostack"):
            if hasattr(p, a):
                delattr(p, a)


@pytest.mark.gpu
def test_weight_gradient_batched_over_the_uses_of_a_two_source_block():
    """The same batching for a block that consumes a skip connection's two tensors (forward_cat): its spectral weight gradient
    is computed once per pass over the stacked spectra; equal to the use-by-use gradients."""
    import uno_amd.integral_operators as io
    from uno_amd.integral_operators import OperatorBlock_2D
    torch.manual_seed(4)
    blk = OperatorBlock_2D(32, 16, 24, 24, 5, 4).to(dev())
    params = list(blk.parameters())
    a = torch.randn(2, 16, 24, 24, device=dev(), requires_grad=True)
    b = torch.randn(2, 16, 24, 24, device=dev(), requires_grad=True)

    def loss_fn():
        h = a
        for _ in range(4):
            h = torch.tanh(blk.forward_cat([h, b], 24, 24))
        return h.square().sum()

    def grads(batched):
        io.TIME_BATCHED_WGRAD = batched
        try:
            for p in params + [a, b]:
                p.grad = None
            loss_fn().backward()
            return [p.grad.clone() for p in params + [a, b]]
        finally:
            io.TIME_BATCHED_WGRAD = True

    w1 = blk.conv.weights1
    for attr in ("_uno_uses", "_uno_stack", "_uno_nostack"):
        if hasattr(w1, attr):
            delattr(w1, attr)
    ref = grads(False)
    assert w1._uno_uses == 4
    for rep in range(2):
        got = grads(True)
        assert w1._uno_stack.n == 4 and w1._uno_stack.done
        for u, v in zip(got, ref):
            ur, vr = (torch.view_as_real(t) if t.is_complex() else t for t in (u, v))
            assert float((ur - vr).norm()) <= 3e-6 * float(vr.norm()) + 1e-12


@pytest.mark.gpu
def test_inplace_weight_gradients_never_lose_a_contribution_silently():
    """A weight that receives gradients from the library's kernels (in place) AND from a stock torch op in the same backward pass:
    autograd may add the stock gradient out of place and so replace the tensor the kernels keep accumulating into.  The library
    must then either still produce the right gradient or fail loudly - never train on a partial sum."""
    import uno_amd.integral_operators as io
    from uno_amd.integral_operators import channel_mix
    torch.manual_seed(2)
    lin = torch.nn.Linear(8, 8).to(dev())
    x = torch.randn(2, 8, 300, device=dev())

    def loss_fn():
        y = channel_mix(x, lin.weight, lin.bias)
        y = y + torch.matmul(lin.weight, y)                    # the same weight through a stock op
        return channel_mix(y, lin.weight, lin.bias).square().sum()

    io.INPLACE_PARAM_GRADS = False
    try:
        lin.zero_grad(set_to_none=True)
        loss_fn().backward()
        ref = lin.weight.grad.clone()
    finally:
        io.INPLACE_PARAM_GRADS = True
    lin.zero_grad(set_to_none=True)
    try:
        loss_fn().backward()
    except RuntimeError as e:
        assert "INPLACE_PARAM_GRADS" in str(e)
        return
    assert float((lin.weight.grad - ref).norm()) <= 2e-6 * float(ref.norm())


@pytest.mark.gpu
@pytest.mark.parametrize("consumer", ["up", "same", "down", "joined"])
def test_gelu_backward_applied_by_the_consumers_last_kernel(consumer):
    """A block without normalisation ends in a GELU.  With `out_join` the block that completes the gradient of its output applies
    gelu'(pre) in its last accumulating channel-mix call (an up-sampling or same-size consumer: its own transposed 1x1 convolution;
    a consumer with a joined second consumer: the deferred closure; a down-sampling consumer alone: a separate pass) and the
    producer runs no GELU-backward kernel.  Same gradients as the ordinary path."""
    from uno_amd.integral_operators import GradJoin, OperatorBlock_2D
    torch.manual_seed(9)
    B, C, S = 2, 64, 24
    So = {"up": 36, "same": 24, "down": 12, "joined": 12}[consumer]
    prod = OperatorBlock_2D(C, C, S, S, 4, 4).to(dev())                       # Non_Lin, no normalisation: fused GELU
    cons = OperatorBlock_2D(C, 32, So, So, 3, 3, Normalize=True).to(dev())
    other = OperatorBlock_2D(2 * C, 32, 36, 36, 4, 4).to(dev())               # second consumer (two-source block, up-sampling)
    x0 = torch.randn(B, C, S, S, device=dev())
    z0 = torch.randn(B, C, S, S, device=dev())
    res = {}
    for mode in ("plain", "fused"):
        x = x0.clone().requires_grad_(True)
        for m in (prod, cons, other):
            m.zero_grad(set_to_none=True)
        j = GradJoin() if mode == "fused" else None
        a = prod(x, S, S, out_join=j)
        y = cons(a, So, So, join=j)
        loss = y.square().sum()
        if consumer == "joined":
            loss = loss + other.forward_cat([z0, a], 36, 36, defer_gelu=True, defer_grad=j).sin().sum()
        loss.backward()
        res[mode] = [x.grad.clone()] + [p.grad.clone() for m in (prod, cons, other) for p in m.parameters() if p.grad is not None]
    assert len(res["plain"]) == len(res["fused"])
    for a, b in zip(res["plain"], res["fused"]):
        ar, br = (torch.view_as_real(t) if t.is_complex() else t for t in (a, b))
        assert float((ar - br).norm()) <= 2e-5 * float(ar.norm()) + 1e-12


def _grads_of(params):
    return [None if p.grad is None else p.grad.clone() for p in params]


def _assert_same_grads(ref, got, tol=2e-6):
    assert len(ref) == len(got)
    for a, b in zip(ref, got):
        assert (a is None) == (b is None)
        if a is None:
            continue
        ar, br = (torch.view_as_real(t) if t.is_complex() else t for t in (a, b))
        assert float((ar - br).norm()) <= tol * float(ar.norm()) + 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("flat", [False, True])
def test_inplace_gradients_under_reentrant_checkpoint(flat):
    """A NESTED backward pass (re-entrant activation checkpointing re-runs the block's forward and back-propagates it inside the
    outer pass, as its own autograd graph task) must not disturb the outer pass's in-place gradient bookkeeping: the block is used
    twice in one graph - once under the checkpoint, once outside - so the outer pass holds a partial sum for its weights while the
    inner pass runs.  Same gradients as the ordinary path (INPLACE_PARAM_GRADS = False), with and without a registered flat buffer."""
    import torch.utils.checkpoint as cp
    import uno_amd.integral_operators as io
    from uno_amd.harness.train import FlatGradients
    torch.manual_seed(3)
    blk = io.OperatorBlock_2D(8, 8, 20, 20, 4, 4).to(dev())
    lin = torch.nn.Linear(8, 8).to(dev())
    params = list(blk.parameters()) + list(lin.parameters())
    x = torch.randn(2, 8, 20, 20, device=dev(), requires_grad=True)

    def loss_fn():
        h = blk(x)                                                                  # outer use: its backward runs LAST
        h = io.channel_mix(h, lin.weight, lin.bias)
        h = cp.checkpoint(lambda t: io.channel_mix(blk(t), lin.weight, lin.bias), h, use_reentrant=True)   # nested pass
        h = blk(h)                                                                  # outer use: its backward runs FIRST
        return h.square().sum()

    res = {}
    for mode in (False, True):
        io.INPLACE_PARAM_GRADS = mode
        try:
            for p in params:
                p.grad = None
                if hasattr(p, "_uno_grad_buffer"):
                    del p._uno_grad_buffer
            x.grad = None
            fg = FlatGradients(params) if (flat and mode) else None
            loss_fn().backward()
            if fg is not None:
                fg.finish()
            res[mode] = _grads_of(params) + [x.grad.clone()]
        finally:
            io.INPLACE_PARAM_GRADS = True
            for p in params:
                if hasattr(p, "_uno_grad_buffer"):
                    del p._uno_grad_buffer
    assert not io._PASSES, "a finished backward pass left its state behind"
    _assert_same_grads(res[False], res[True])


@pytest.mark.gpu
def test_inplace_gradients_with_autograd_grad_inside_a_hook():
    """torch.autograd.grad called from a tensor hook DURING a backward pass is a second graph task on the same thread.  It uses the
    same layers; its state must be its own, and the outer pass must go on summing into the tensors it started with."""
    import uno_amd.integral_operators as io
    torch.manual_seed(4)
    blk = io.OperatorBlock_2D(8, 8, 20, 20, 4, 4).to(dev())
    params = list(blk.parameters())
    x = torch.randn(2, 8, 20, 20, device=dev(), requires_grad=True)
    z = torch.randn(2, 8, 20, 20, device=dev(), requires_grad=True)
    seen = []

    def hook(g):
        # an independent little graph through the same block, differentiated with respect to its input only
        with torch.enable_grad():                       # hooks run with grad mode off
            (gz,) = torch.autograd.grad(blk(z).sin().sum(), z)
        seen.append(gz)
        return g

    def loss_fn():
        h = blk(x)
        h.register_hook(hook)
        h = blk(blk(h))
        return h.square().sum()

    res = {}
    for mode in (False, True):
        io.INPLACE_PARAM_GRADS = mode
        try:
            for p in params:
                p.grad = None
            x.grad = None
            seen.clear()
            loss_fn().backward()
            res[mode] = _grads_of(params) + [x.grad.clone(), seen[0].clone()]
        finally:
            io.INPLACE_PARAM_GRADS = True
    assert not io._PASSES
    _assert_same_grads(res[False], res[True])


@pytest.mark.gpu
@pytest.mark.parametrize("fallback", ["three_sources", "three_source_projection"])
def test_join_is_voided_when_a_consumer_cannot_defer(fallback):
    """out_join / defer_grad with a second consumer that falls back to the stock-op path (three sources cannot be consumed in place;
    a projection with two output channels has no fused form): that consumer's gradient reaches the producer through autograd
    WITHOUT gelu'(pre), so the join's owner must not apply the factor to its own share either - the producer applies it to the
    sum."""
    from uno_amd.integral_operators import GradJoin, OperatorBlock_2D, channel_mix_cat
    torch.manual_seed(10)
    B, C, S = 2, 64, 24
    prod = OperatorBlock_2D(C, C, S, S, 4, 4).to(dev())
    cons = OperatorBlock_2D(C, 32, S, S, 3, 3, Normalize=True).to(dev())
    other = OperatorBlock_2D(2 * C, 32, 36, 36, 4, 4).to(dev())
    lin = torch.nn.Linear(2 * C, 16).to(dev())
    x0 = torch.randn(B, C, S, S, device=dev())
    z0 = torch.randn(B, C, S, S, device=dev())
    res = {}
    for mode in ("plain", "joined"):
        x = x0.clone().requires_grad_(True)
        for m in (prod, cons, other, lin):
            m.zero_grad(set_to_none=True)
        j = GradJoin() if mode == "joined" else None
        a = prod(x, S, S, out_join=j)
        y = cons(a, S, S, join=j)
        if fallback == "three_sources":
            second = other.forward_cat([z0[:, :32], z0[:, 32:], a], 36, 36, defer_grad=j)
        else:
            second = channel_mix_cat([z0, a, z0[:, :0]], lin.weight, lin.bias, defer_grad=j)       # three sources: stock cat + one-source kernel
        (y.square().sum() + second.sin().sum()).backward()
        res[mode] = [x.grad.clone()] + [p.grad.clone() for m in (prod, cons, other, lin) for p in m.parameters() if p.grad is not None]
    _assert_same_grads(res["plain"], res["joined"], tol=2e-5)


@pytest.mark.gpu
def test_training_step_without_the_private_autograd_entry_points(monkeypatch):
    """with torch._C._current_graph_task_id / the engine's callback queue unavailable (simulated) the whole Darcy
    step - flat gradient buffer, joined skip gradients, a layer used three times - runs on the ordinary gradient path and gives the
    default path's parameters."""
    import uno_amd.integral_operators as io
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    res = []
    for available in (True, False):
        monkeypatch.setattr(io, "_PASS_STATE_AVAILABLE", available)
        torch.manual_seed(0)
        model = UNO_9(3, 8, pad=5).to(dev())
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
        a, u = synthetic_darcy_batch(2, 72, 3, dev())
        for _ in range(2):
            loss = tr.step(a, u)
        res.append([loss.clone()] + [p.detach().clone() for p in model.parameters()])
    for x, y in zip(*res):
        xr, yr = (torch.view_as_real(t) if t.is_complex() else t for t in (x, y))
        assert float((xr - yr).norm()) <= 1e-5 * float(xr.norm()) + 1e-12
