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
