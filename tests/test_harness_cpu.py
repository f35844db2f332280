"""Host logic of the harness on CPU: the UNO_9 counterpart, relative-L2 loss, ComplexAdam and the
data-parallel step, pinned by reference-generated golden vectors (tests/golden/harness.npz).

The spectral layers have no CPU path in the product, so these tests plug the ORACLE's operator block
in as a test double (block_cls=...) - this exercises the harness, not the HIP kernels (those are
covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import Case, load_cases, rel_err
from oracle import spectral_oracle as so
from uno_amd.harness import ComplexAdam, DarcyTrainer, UNO_9, lp_loss_rel_sum, synthetic_darcy_batch

ZH, _ = load_cases("harness.npz")


def _uno9_from_golden():
    c = Case(ZH, "uno9")
    S, B, width, pad = [int(v) for v in c.meta]
    model = UNO_9(3, width, pad=pad, block_cls=so.OracleOperatorBlock2d)
    sd = {k: torch.from_numpy(v.copy()) for k, v in c.sub("sd").items()}
    model.load_state_dict(sd, strict=True)
    return c, model, S, B


def test_state_dict_is_reference_compatible():
    c = Case(ZH, "uno9")
    product = UNO_9(3, int(c.meta[2]), pad=int(c.meta[3]))        # product blocks (constructible on CPU)
    sd = {k: torch.from_numpy(v.copy()) for k, v in c.sub("sd").items()}
    product.load_state_dict(sd, strict=True)
    assert product.conv0.conv.weights1.dtype == torch.complex64
    assert [k for k in product.state_dict()] == list(sd)


def test_uno9_forward_loss_grads_match_reference():
    c, model, S, B = _uno9_from_golden()
    a, u = torch.from_numpy(c.a), torch.from_numpy(c.u)
    pred = model(a).reshape(B, S, S)
    assert rel_err(pred.detach().numpy(), c.pred0) < 1e-5
    loss = lp_loss_rel_sum(pred.view(B, -1), u.view(B, -1))
    assert abs(float(loss) - float(c.losses[0])) < 1e-5 * abs(float(c.losses[0]))
    loss.backward()
    gmax = max(float(getattr(c, f"gradnorm.{k}")) for k, _ in model.named_parameters())
    for k, p in model.named_parameters():
        ref = float(getattr(c, f"gradnorm.{k}"))
        # floor: a conv bias in front of an InstanceNorm has a zero true gradient (what is stored is rounding residue)
        assert abs(float(torch.linalg.vector_norm(p.grad)) - ref) <= 2e-4 * ref + 1e-6 * gmax, k
    for k, g in c.sub("grad").items():
        got = dict(model.named_parameters())[k].grad.numpy()
        assert np.linalg.norm((got - g).ravel()) <= 2e-4 * np.linalg.norm(g.ravel()) + 1e-6 * gmax, k


def test_three_training_steps_match_reference_adam():
    c, model, S, B = _uno9_from_golden()
    a, u = torch.from_numpy(c.a), torch.from_numpy(c.u)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    losses = [float(tr.step(a, u)) for _ in range(3)]
    assert np.allclose(losses, c.losses, rtol=2e-4)
    for k, p in model.named_parameters():
        ref = float(getattr(c, f"after3.norm.{k}"))
        assert abs(float(torch.linalg.vector_norm(p)) - ref) <= 1e-4 * ref + 1e-9, k
    for k, v in c.sub("after3").items():
        if k.startswith("norm.") or k.startswith("sum."):
            continue
        assert rel_err(dict(model.named_parameters())[k].detach().numpy(), v) < 1e-3, k


def test_complex_adam_matches_reference():
    c = Case(ZH, "adam")
    pc = torch.nn.Parameter(torch.from_numpy(c.pc0.copy()))
    pr = torch.nn.Parameter(torch.from_numpy(c.pr0.copy()))
    opt = ComplexAdam([pc, pr], lr=1e-2, weight_decay=1e-3)
    for t in range(3):
        pc.grad = torch.from_numpy(c.gc[t].copy())
        pr.grad = torch.from_numpy(c.gr[t].copy())
        opt.step()
    assert rel_err(pc.detach().numpy(), c.pc3) < 1e-6
    assert rel_err(pr.detach().numpy(), c.pr3) < 1e-6


def test_grid_matches_numpy_linspace():
    m = UNO_9(3, 4)
    g = m.get_grid((2, 7, 5, 1), torch.device("cpu"))
    gx = np.linspace(0, 1, 7).astype(np.float32)
    gy = np.linspace(0, 1, 5).astype(np.float32)
    assert np.array_equal(g[0, :, 0, 0].numpy(), gx) and np.array_equal(g[1, 3, :, 1].numpy(), gy)


# ----------------------------------------------------------------------------- data parallel (gloo)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, out_path, bucket_mb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                 # different init per rank: broadcast must fix it
        model = UNO_9(3, 4, pad=5, block_cls=so.OracleOperatorBlock2d)
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3, bucket_mb=bucket_mb)
        if rank == 0:
            torch.save(len(tr.grads.buckets), out_path + ".nbuckets")
        a, u = synthetic_darcy_batch(4, 72, seed=7, device="cpu")      # global batch, identical on all ranks
        sl = slice(rank * 2, rank * 2 + 2)
        for _ in range(2):
            loss = tr.step(a[sl], u[sl])
        if rank == 0:
            torch.save({k: v.clone() for k, v in model.state_dict().items()}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [32.0, 0.002])
def test_data_parallel_equals_single_process_global_batch(tmp_path, bucket_mb):
    """world_size=2 gloo: two ranks on half batches == one process on the concatenated batch
    (gradients are SUMMED across ranks because the reference loss is a sum over samples) - with one
    bucket and with many small buckets all-reduced while the backward pass runs."""
    out_path = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, _free_port(), out_path, bucket_mb), nprocs=2, join=True)
    dp = torch.load(out_path)
    nb = torch.load(out_path + ".nbuckets")
    assert (nb == 1) if bucket_mb > 1 else (nb > 4)

    torch.manual_seed(100)                            # rank 0's init is what gets broadcast
    model = UNO_9(3, 4, pad=5, block_cls=so.OracleOperatorBlock2d)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(4, 72, seed=7, device="cpu")
    for _ in range(2):
        tr.step(a, u)
    for k, v in model.state_dict().items():
        assert rel_err(torch.view_as_real(dp[k]).numpy() if v.is_complex() else dp[k].numpy(),
                       torch.view_as_real(v).numpy() if v.is_complex() else v.numpy()) < 2e-4, k


def _unused_param_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from uno_amd.harness.train import FlatGradients
        torch.manual_seed(rank)
        used = torch.nn.Linear(5, 3)
        unused = torch.nn.Linear(7, 2)                    # takes no part in the backward: its bucket never completes by hooks
        tail = torch.nn.Linear(3, 1)
        params = list(used.parameters()) + list(unused.parameters()) + list(tail.parameters())
        fg = FlatGradients(params, bucket_mb=1e-5)        # one bucket per parameter
        assert len(fg.buckets) >= 4
        x = torch.full((4, 5), float(rank + 1))
        for _ in range(2):                                # hooks must re-arm every step
            fg.zero_()
            loss = tail(used(x)).sum()
            fg.arm()
            loss.backward()
            fg.finish()
        if rank == 0:
            torch.save(fg.flat.clone(), out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_with_unused_parameters(tmp_path):
    out_path = str(tmp_path / "flat.pt")
    mp.spawn(_unused_param_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    flat = torch.load(out_path)
    # both ranks: same weights? no - seeds differ, so compute the expected sum directly
    exp = 0
    for rank in range(2):
        torch.manual_seed(rank)
        used = torch.nn.Linear(5, 3)
        unused = torch.nn.Linear(7, 2)
        tail = torch.nn.Linear(3, 1)
        x = torch.full((4, 5), float(rank + 1))
        tail(used(x)).sum().backward()
        g = [p.grad if p.grad is not None else torch.zeros_like(p)
             for p in list(used.parameters()) + list(unused.parameters()) + list(tail.parameters())]
        exp = exp + torch.cat([t.reshape(-1) for t in g])
    assert torch.allclose(flat, exp, rtol=1e-6, atol=1e-7)


def _bf16_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for name, cd in (("f32", None), ("bf16", torch.bfloat16)):
            torch.manual_seed(100)
            model = UNO_9(3, 4, pad=5, block_cls=so.OracleOperatorBlock2d)
            tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3, bucket_mb=0.002, comm_dtype=cd)
            a, u = synthetic_darcy_batch(4, 72, seed=7, device="cpu")
            sl = slice(rank * 2, rank * 2 + 2)
            tr.grads.zero_()
            B = 2
            loss = lp_loss_rel_sum(model(a[sl]).reshape(B, -1), u[sl].reshape(B, -1))
            tr.grads.arm(None)
            loss.backward()
            tr.grads.finish()
            out[name] = tr.grads.flat.clone()
            tr.opt.step()                               # and the step runs on the widened buffer
            out[name + "_p"] = torch.cat([(torch.view_as_real(q.detach()) if q.is_complex() else q.detach()).reshape(-1) for q in model.parameters()])
        if rank == 0:
            torch.save(out, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_buckets_match_float32_exchange(tmp_path):
    """SURVEY 8(e) / DESIGN section 6: optional bfloat16 gradient buckets (half the bytes on the links; float32 buffer, optimiser state
    and update) - world size 2, gloo: the exchanged gradient agrees with the float32 exchange to bfloat16 rounding (relative L2 4e-3:
    one rounding of each rank's contribution + one of their sum, 2^-9 each), the parameters after the update to 1e-5."""
    out_path = str(tmp_path / "bf16.pt")
    mp.spawn(_bf16_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    r = torch.load(out_path)
    assert r["bf16"].dtype == torch.float32
    e = float((r["bf16"] - r["f32"]).norm() / r["f32"].norm())
    assert 0 < e < 4e-3, e                              # (0 < : the low-precision path really ran)
    assert float((r["bf16_p"] - r["f32_p"]).norm() / r["f32_p"].norm()) < 1e-5


def test_aborted_backward_leaves_gradient_buckets_and_pass_state_clean():
    """A backward pass that raises under arm() (an out-of-memory error the training loop catches): the trainer waits for what was
    issued, resets the bucket counters and releases the library's pass state - the next step runs as if nothing had happened."""
    import uno_amd.integral_operators as io
    from uno_amd.harness.train import FlatGradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        lin1, lin2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 1)
        tr = DarcyTrainer(torch.nn.Sequential(lin1, lin2), lr=1e-3, weight_decay=0.0, force_collectives=True, bucket_mb=1e-5)
        assert len(tr.grads.buckets) >= 3

        class Boom(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x.clone()

            @staticmethod
            def backward(ctx, g):
                raise RuntimeError("boom")
        x = torch.ones(3, 6)
        io._PASSES[424242] = {"id": 424242, "acc": {}, "stacks": {}, "uses": {}, "born": 0.0}      # what a failed pass leaves behind
        with pytest.raises(RuntimeError, match="boom"):
            tr.step_with(lambda: lin2(Boom.apply(lin1(x))).sum())          # lin2's buckets are issued, then the pass dies
        g = tr.grads
        assert not g._armed and g._works == [] and g._next == 0 and g._pending == list(g._bucket_params)
        assert not io._PASSES
        ref1, ref2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 1)
        ref1.load_state_dict(lin1.state_dict()); ref2.load_state_dict(lin2.state_dict())
        tr.grads.zero_()
        loss = lin2(lin1(x)).sum()
        tr.grads.arm(None, True)
        loss.backward()
        tr.grads.finish()
        ref2(ref1(x)).sum().backward()
        exp = torch.cat([q.grad.reshape(-1) for q in list(ref1.parameters()) + list(ref2.parameters())])
        assert torch.allclose(tr.grads.flat[:exp.numel()], exp, rtol=1e-6, atol=1e-7)
    finally:
        dist.destroy_process_group()


def test_flat_gradients_rebind_after_zero_grad_set_to_none():
    from uno_amd.harness.train import FlatGradients
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    fg = FlatGradients(lin.parameters())
    lin(torch.ones(2, 4)).sum().backward()
    fg.finish()                            # gradients autograd produced outside the buffer are collected into it
    assert fg.flat.abs().sum() > 0
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(lin.parameters(), fg.views))
    for p in lin.parameters():
        p.grad = None                      # what optimizer.zero_grad() does by default
    fg.zero_()
    assert all(p.grad is None for p in lin.parameters())     # a step starts with released gradients: written once, never added
    lin(torch.ones(2, 4)).sum().backward()
    fg.finish()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(lin.parameters(), fg.views))
    assert torch.allclose(fg.flat[:12].view(3, 4), torch.full((3, 4), 2.0)) and torch.allclose(fg.flat[12:], torch.full((3,), 2.0))


def test_flat_gradients_zero_the_segment_of_a_parameter_without_gradient():
    """zero_() releases the gradients without clearing the flat buffer; a parameter that takes part in one step but not in the
    next must contribute ZERO to the next sum over ranks, not the previous step's values."""
    from uno_amd.harness.train import FlatGradients
    torch.manual_seed(0)
    a, b = torch.nn.Linear(4, 3), torch.nn.Linear(3, 2)
    fg = FlatGradients(list(a.parameters()) + list(b.parameters()))
    fg.zero_()
    b(a(torch.ones(2, 4))).sum().backward()
    fg.finish()
    assert fg.flat[15:].abs().sum() > 0                   # b's segment is live
    fg.zero_()
    a(torch.ones(2, 4)).sum().backward()                  # b takes no part in this step
    fg.finish()
    assert b.weight.grad is None and float(fg.flat[15:].abs().sum()) == 0.0
    assert torch.allclose(fg.flat[:12].view(3, 4), torch.full((3, 4), 2.0))


def test_flat_gradients_odd_real_prefix_before_complex_param():
    """A complex parameter behind an odd number of real entries still gets a valid complex view."""
    from uno_amd.harness.train import FlatGradients
    a = torch.nn.Parameter(torch.zeros(3))
    z = torch.nn.Parameter(torch.zeros(2, 2, dtype=torch.cfloat))
    b = torch.nn.Parameter(torch.zeros(5))
    z2 = torch.nn.Parameter(torch.zeros(3, dtype=torch.cfloat))
    fg = FlatGradients([a, z, b, z2], bucket_mb=1e-5)
    assert fg.views[1].is_complex() and tuple(fg.views[1].shape) == (2, 2) and tuple(fg.views[3].shape) == (3,)
    assert z._uno_grad_buffer is fg.views[1]
    loss = (a * 2).sum() + (z * (1 + 2j)).real.sum() + (b * 3).sum() + (z2 * (0 + 1j)).imag.sum()
    loss.backward()
    fg.finish()
    assert z.grad.data_ptr() == fg.views[1].data_ptr()
    assert torch.allclose(a.grad, torch.full((3,), 2.0)) and torch.allclose(b.grad, torch.full((5,), 3.0))
    assert torch.allclose(z.grad, torch.full((2, 2), 1 - 2j, dtype=torch.cfloat))
    # every view lies inside the flat buffer and the buckets cover all of them
    lo = min(s for s, _ in fg.buckets)
    hi = max(e for _, e in fg.buckets)
    assert lo == 0 and hi <= fg.flat.numel()
    fg.zero_()
    assert z.grad is None and a.grad is None


def test_complex_adam_per_parameter_step_counts():
    """A parameter whose grad is None on some steps keeps its own step count and bias correction (reference Adam.py keeps
    `state['step']` per parameter) instead of tripping an assertion."""
    torch.manual_seed(0)
    p1 = torch.nn.Parameter(torch.randn(4))
    p2 = torch.nn.Parameter(torch.randn(3, dtype=torch.cfloat))
    q1, q2 = (torch.nn.Parameter(p.detach().clone()) for p in (p1, p2))
    opt = ComplexAdam([p1, p2], lr=1e-2)
    ref1, ref2 = ComplexAdam([q1], lr=1e-2), ComplexAdam([q2], lr=1e-2)
    for t in range(4):
        g1, g2 = torch.randn(4), torch.randn(3, dtype=torch.cfloat)
        p1.grad, q1.grad = g1.clone(), g1.clone()
        ref1.step()
        if t % 2 == 0:                      # p2 takes part in every other step only
            p2.grad, q2.grad = g2.clone(), g2.clone()
            ref2.step()
        else:
            p2.grad = None
        opt.step()
    assert opt.state[p1]["step"] == 4 and opt.state[p2]["step"] == 2
    assert torch.allclose(p1, q1) and torch.allclose(p2, q2)


def test_reference_style_caller_reproduces_the_reference_on_cpu():
    """tools/reference_style_caller.py (the reference's calling convention: channels-last Linear, permute, F.pad, torch.cat) with the
    oracle blocks on the host reproduces the reference's prediction - it is the same op sequence."""
    from tools.reference_style_caller import UNO_9_ReferenceStyle
    c = Case(ZH, "uno9")
    S, B, width, pad = [int(v) for v in c.meta]
    model = UNO_9_ReferenceStyle(3, width, pad=pad, block_cls=so.OracleOperatorBlock2d)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in c.sub("sd").items()}, strict=True)
    pred = model(torch.from_numpy(c.a)).reshape(B, S, S)
    assert rel_err(pred.detach().numpy(), c.pred0) < 1e-6


# ----------------------------------------------------------------------------- data parallel, NS-3D workload (gloo)
def _dp_workload_worker(rank, world, port, out_path, name):
    from uno_amd.harness import workloads
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # different init per rank (the trainer's broadcast must fix it), the same global batch of 4 everywhere
        w = workloads.build(name, "cpu", batch=4, seed=11, small=True, model_seed=100 + rank,
                            block_cls=so.OracleOperatorBlock3d if name == "c4" else so.OracleOperatorBlock2d, bucket_mb=0.01)
        assert len(w.trainer.grads.buckets) > 3
        for _ in range(2):
            w.step(2 * rank, 2 * rank + 2)
        if rank == 0:
            torch.save(workloads.flat_params(w.trainer.model), out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ns3d_workload_data_parallel_equals_single_process(tmp_path):
    """BASELINE.json configs[3] (NS-3D, DDP): the step bench.py --workload c4 times - Uno3D_T20 + ns3d_loss under FlatGradients -
    at world size 2 (gloo, oracle blocks): two ranks on half batches == one process on the whole batch after two steps
    (reference ns_train_3d.py:48-70: the loss is a sum over samples, so gradients are SUMMED across ranks)."""
    from uno_amd.harness import workloads
    out_path = str(tmp_path / "dp_c4.pt")
    mp.spawn(_dp_workload_worker, args=(2, _free_port(), out_path, "c4"), nprocs=2, join=True)
    got = torch.load(out_path)
    w = workloads.build("c4", "cpu", batch=4, seed=11, small=True, model_seed=100, block_cls=so.OracleOperatorBlock3d)
    for _ in range(2):
        w.step()
    ref = workloads.flat_params(w.trainer.model)
    assert float((got - ref).norm()) <= 2e-4 * float(ref.norm())
