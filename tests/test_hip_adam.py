"""K10 (csrc/adam.hip) against the reference optimiser's semantics (Adam.py:27-52): golden vectors generated
from the reference (tests/golden/harness.npz, case "adam") and the multi-tensor CPU implementation on random
tensors of awkward sizes (tails, views at odd offsets of a flat buffer)."""
import numpy as np
import pytest
import torch

from conftest import Case, load_cases, rel_err
from uno_amd.harness.optim import ComplexAdam

pytestmark = pytest.mark.gpu
ZH, _ = load_cases("harness.npz")


def test_adam_kernel_matches_reference_golden():
    c = Case(ZH, "adam")
    pc = torch.nn.Parameter(torch.from_numpy(c.pc0.copy()).cuda())
    pr = torch.nn.Parameter(torch.from_numpy(c.pr0.copy()).cuda())
    opt = ComplexAdam([pc, pr], lr=1e-2, weight_decay=1e-3)
    for t in range(3):
        pc.grad = torch.from_numpy(c.gc[t].copy()).cuda()
        pr.grad = torch.from_numpy(c.gr[t].copy()).cuda()
        opt.step()
    assert rel_err(pc.detach().cpu().numpy(), c.pc3) < 1e-6
    assert rel_err(pr.detach().cpu().numpy(), c.pr3) < 1e-6


@pytest.mark.parametrize("shape,cplx", [((1,), False), ((7,), True), ((1001,), False), ((33, 5, 3), True), ((64, 128, 9), True)])
def test_adam_kernel_matches_cpu_path(shape, cplx):
    g = torch.Generator().manual_seed(sum(shape))
    dt = torch.complex64 if cplx else torch.float32
    p0 = torch.randn(*shape, dtype=dt, generator=g)
    grads = [torch.randn(*shape, dtype=dt, generator=g) for _ in range(4)]
    pc = torch.nn.Parameter(p0.clone())
    # device parameter whose gradient is a view at an odd offset of a flat buffer (4-byte alignment only)
    pd = torch.nn.Parameter(p0.clone().cuda())
    nflt = p0.numel() * (2 if cplx else 1)
    off = 2 if cplx else 3                     # complex views need 8-byte alignment, nothing needs 16
    flat = torch.zeros(nflt + off, device="cuda")
    seg = flat[off:off + nflt]
    pd.grad = torch.view_as_complex(seg.view(*shape, 2)) if cplx else seg.view(shape)
    oc = ComplexAdam([pc], lr=3e-3, weight_decay=1e-2)
    od = ComplexAdam([pd], lr=3e-3, weight_decay=1e-2)
    for gr in grads:
        pc.grad = gr.clone()
        pd.grad.copy_(gr.cuda())
        oc.step()
        od.step()
    a = torch.view_as_real(pd.detach()).cpu() if cplx else pd.detach().cpu()
    b = torch.view_as_real(pc.detach()) if cplx else pc.detach()
    assert rel_err(a.numpy(), b.numpy()) < 1e-6
    sa, sb = od.state[pd], oc.state[pc]
    assert rel_err(sa["exp_avg_sq"].cpu().numpy(), sb["exp_avg_sq"].numpy()) < 1e-6


def test_adam_multi_tensor_launches_many_mixed_tensors():
    """60 parameter tensors - real and complex, 1 to 70000 entries - through the multi-tensor kernel (24 tensors per launch: three
    launches with the type mask and the workgroup -> tensor map changing between them) against the CPU path, 3 steps."""
    rng = np.random.default_rng(12)
    g = torch.Generator().manual_seed(5)
    shapes = [tuple(int(v) for v in rng.integers(1, [70000, 300, 40][nd - 1] + 1, size=nd)) for nd in rng.integers(1, 4, size=60)]
    cplx = [bool(v) for v in rng.integers(0, 2, size=60)]
    p0 = [torch.randn(*s, dtype=torch.complex64 if c else torch.float32, generator=g) for s, c in zip(shapes, cplx)]
    pc = [torch.nn.Parameter(t.clone()) for t in p0]
    pd = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    oc = ComplexAdam(pc, lr=2e-3, weight_decay=1e-3)
    od = ComplexAdam(pd, lr=2e-3, weight_decay=1e-3)
    for _ in range(3):
        for a, b in zip(pc, pd):
            gr = torch.randn(*a.shape, dtype=a.dtype, generator=g)
            a.grad = gr.clone()
            b.grad = gr.cuda()
        oc.step()
        od.step()
    for a, b in zip(pc, pd):
        x = torch.view_as_real(b.detach()).cpu() if b.is_complex() else b.detach().cpu()
        y = torch.view_as_real(a.detach()) if a.is_complex() else a.detach()
        assert rel_err(x.numpy(), y.numpy()) < 1e-6, tuple(a.shape)


def _toy_params(seed):
    g = torch.Generator().manual_seed(seed)
    pc = torch.nn.Parameter(torch.randn(9, 5, dtype=torch.cfloat, generator=g).cuda())
    pr = torch.nn.Parameter(torch.randn(33, generator=g).cuda())
    return [pc, pr]


def _toy_grads(seed, params):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(p.shape, dtype=p.dtype, generator=g).cuda() for p in params]


def test_capturable_lr_schedule_reaches_a_replayed_update():
    """the captured update must not freeze lr / eps / weight decay.  One capture of opt.step(); between replays a
    scheduler halves group['lr'] (reference ns_train_2d.py:37,113: StepLR): the replays equal the eager (host-counted) optimiser
    bit for bit."""
    pe, pg = _toy_params(3), _toy_params(3)
    oe = ComplexAdam(pe, lr=1e-2, weight_decay=1e-3)
    og = ComplexAdam(pg, lr=1e-2, weight_decay=1e-3, capturable=True)
    for p in pg:
        p.grad = torch.zeros_like(p)
    og.init_state()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        og.step()
    # (capture does not execute: counters and parameters are untouched so far)
    for t in range(4):
        if t == 2:
            for o in (oe, og):
                o.param_groups[0]["lr"] *= 0.5
                o.param_groups[0]["weight_decay"] = 5e-4
        gs = _toy_grads(100 + t, pe)
        for p, q, g in zip(pe, pg, gs):
            p.grad = g.clone()
            q.grad.copy_(g)
        oe.step()
        og.sync_hyper()
        graph.replay()
        torch.cuda.synchronize()
        for p, q in zip(pe, pg):
            assert torch.equal(p.detach(), q.detach()), t
    assert int(og.state[pg[0]]["step"]) == 4


def test_capturable_step_count_survives_state_dict_round_trip():
    """a resumed capturable optimiser continues the bias corrections from the loaded step count (reference Adam.py
    resumes from state['step']), it does not restart at t = 1 against warm moments."""
    pa, pb, pe = _toy_params(4), _toy_params(4), _toy_params(4)
    oa = ComplexAdam(pa, lr=1e-2, weight_decay=1e-3, capturable=True)
    oe = ComplexAdam(pe, lr=1e-2, weight_decay=1e-3)
    for t in range(3):
        gs = _toy_grads(200 + t, pa)
        for p, q, g in zip(pa, pe, gs):
            p.grad = g.clone()
            q.grad = g.clone()
        oa.step()
        oe.step()
    sd = oa.state_dict()
    ob = ComplexAdam(pb, lr=1e-2, weight_decay=1e-3, capturable=True)
    with torch.no_grad():
        for p, q in zip(pb, pa):
            p.copy_(q)
    ob.load_state_dict(sd)
    for t in range(3, 5):
        gs = _toy_grads(200 + t, pb)
        for p, q, g in zip(pb, pe, gs):
            p.grad = g.clone()
            q.grad = g.clone()
        ob.step()
        oe.step()
        for p, q in zip(pb, pe):
            assert torch.equal(p.detach(), q.detach()), t
    assert int(ob.state[pb[0]]["step"]) == 5
    # and the other direction: a host-counted optimiser's state loaded into a capturable one
    oc = ComplexAdam(_toy_params(4), lr=1e-2, weight_decay=1e-3, capturable=True)
    oc.load_state_dict(oe.state_dict())
    oc.init_state()
    assert int(oc.state[oc.param_groups[0]["params"][0]]["step"]) == 5
