"""NS-2D (`UNO`) and NS-3D (`Uno3D_T20`) harness models + their training losses vs reference-generated golden
values (tests/golden/harness_ns.npz).  The golden weights are seeded, not stored (3-D spectral weights are tens of
MB); every test first verifies per-parameter checksums, i.e. that the constructor reproduced the reference's
initialisation bit for bit.  CPU tests use the oracle blocks as test doubles; -m gpu tests use the product path."""
import numpy as np
import pytest
import torch

from conftest import Case, load_cases, rel_err
from oracle import spectral_oracle as so
from uno_amd.harness import UNO, Uno3D_T20, ns2d_rollout_loss, ns3d_loss

Z, _ = load_cases("harness_ns.npz")


def _check_init(model, c):
    for k, p in model.named_parameters():
        ck = getattr(c, f"ck.{k}")
        got = np.array([float(p.detach().abs().sum()), float(torch.linalg.vector_norm(p.detach()))])
        assert np.allclose(got, ck, rtol=1e-6), f"seeded init of {k} differs from the reference's"


def _check_grads(model, c, rtol):
    gmax = max(float(getattr(c, f"gradnorm.{k}")) for k, _ in model.named_parameters())
    for k, p in model.named_parameters():
        ref = float(getattr(c, f"gradnorm.{k}"))
        assert abs(float(torch.linalg.vector_norm(p.grad)) - ref) <= rtol * ref + 1e-5 * gmax, k


def _ns2d(block_cls, dev, tol_pred, tol_grad):
    c = Case(Z, "ns2d")
    torch.manual_seed(21)
    model = UNO(14, 4, block_cls=block_cls) if block_cls else UNO(14, 4)
    _check_init(model, c)
    model = model.to(dev)
    xx, yy = torch.from_numpy(c.xx).to(dev), torch.from_numpy(c.yy).to(dev)
    with torch.no_grad():
        p0 = model(xx)
    assert rel_err(p0.cpu().numpy()[..., 0], c.pred[..., 0]) < tol_pred
    loss = ns2d_rollout_loss(model, xx, yy, T_f=2, step=1)
    loss.backward()
    assert abs(float(loss) - float(c.loss)) < tol_pred * abs(float(c.loss))
    _check_grads(model, c, tol_grad)


def _ns3d(block_cls, dev, tol_pred, tol_grad):
    c = Case(Z, "ns3d")
    torch.manual_seed(31)
    model = Uno3D_T20(6, 2, pad=3, block_cls=block_cls) if block_cls else Uno3D_T20(6, 2, pad=3)
    _check_init(model, c)
    model = model.to(dev)
    x, y = torch.from_numpy(c.x).to(dev), torch.from_numpy(c.y).to(dev)
    loss = ns3d_loss(model, x, y)
    loss.backward()
    with torch.no_grad():
        pred = model(x).view(1, 32, 32, 20)
    assert rel_err(pred.cpu().numpy(), c.pred) < tol_pred
    assert abs(float(loss) - float(c.loss)) < tol_pred * abs(float(c.loss))
    _check_grads(model, c, tol_grad)


def test_ns2d_cpu_oracle_blocks():
    _ns2d(so.OracleOperatorBlock2d, "cpu", 1e-5, 5e-4)


def test_ns3d_cpu_oracle_blocks():
    _ns3d(so.OracleOperatorBlock3d, "cpu", 1e-5, 5e-4)


@pytest.mark.gpu
def test_ns2d_gpu_product():
    _ns2d(None, torch.device("cuda:0"), 1e-5, 5e-4)


@pytest.mark.gpu
def test_ns3d_gpu_product():
    # InstanceNorm3d runs on the K13 kernel (it was MIOpen's batch norm, 1e-3 / 2e-2 then); what remains is rocFFT vs
    # pocketfft inside pointwise_op_3D
    _ns3d(None, torch.device("cuda:0"), 1e-4, 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("capturable", [False, True])
def test_graphed_step_equals_eager_step(capturable):
    """harness.GraphedStep: forward + loss + backward replayed from a HIP graph give the eager step's loss, gradients and
    updated parameters bit for bit (every kernel is deterministic), for several batches through one capture.  capturable: the
    optimiser update is INSIDE the graph (step count and bias corrections on the device, reference Adam.py:27-52) - three replays
    equal three eager steps of the host-counted optimiser bit for bit, and the device counter reads 3."""
    from uno_amd.harness import ComplexAdam, GraphedStep
    dev = torch.device("cuda:0")
    def make(cap=False):
        torch.manual_seed(5)
        m = UNO(14, 4).to(dev)
        return m, ComplexAdam(m.parameters(), lr=1e-3, weight_decay=1e-4, capturable=cap)
    g = torch.Generator().manual_seed(9)
    batches = [(torch.randn(2, 64, 64, 10, generator=g).to(dev), torch.randn(2, 64, 64, 3, generator=g).to(dev)) for _ in range(3)]
    me, oe = make()
    mg, og = make(capturable)
    gs = GraphedStep(mg, og, lambda a, b: ns2d_rollout_loss(mg, a, b, T_f=3, step=1), batches[0])
    # the eager model's first backward pass runs the spectral weight gradients use by use and only the later ones batch them over
    # the roll-out (integral_operators.TIME_BATCHED_WGRAD) - the capture's warm-up passes have put the graphed model in that mode
    ns2d_rollout_loss(me, *batches[0], T_f=3, step=1).backward()
    for xx, yy in batches:
        oe.zero_grad(set_to_none=True)
        le = ns2d_rollout_loss(me, xx, yy, T_f=3, step=1)
        le.backward()
        ge = {k: p.grad.clone() for k, p in me.named_parameters()}
        oe.step()
        lg = gs.step(xx, yy)
        assert float(lg) == float(le)
        for (k, pe), (_, pg) in zip(me.named_parameters(), mg.named_parameters()):
            assert torch.equal(ge[k], pg.grad), k
            assert torch.equal(pe, pg), k
    assert gs.opt_in_graph == capturable
    if capturable:
        p0 = next(iter(mg.parameters()))
        assert int(og.state[p0]["step"]) == len(batches)
