"""RCCL smoke test of the data-parallel step on ONE GPU: a 1-rank "nccl" process group with the collectives
forced on (broadcast of parameters, all-reduce of the flat gradient buffer incl. complex grads, barrier) must
reproduce the plain single-process step bit for bit.  The multi-rank arithmetic is covered on CPU with gloo
(tests/test_harness_cpu.py); the multi-GPU run itself is the driver's.  pytest -m gpu
(The file sorts LAST on purpose: process-group set-up is the one part of the suite that depends on the box outside the GPU - a
rendezvous that hangs there ends a `-x` run after every kernel test has reported, not before 450 of them.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_group_matches_plain_step():
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    dev = torch.device("cuda:0")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    a, u = synthetic_darcy_batch(2, 72, 5, dev)

    def run(force):
        torch.manual_seed(0)
        model = UNO_9(3, 4, pad=5).to(dev)
        tr = DarcyTrainer(model, force_collectives=force)
        losses = [float(tr.step(a, u)) for _ in range(2)]
        return losses, [p.detach().clone() for p in model.parameters()]

    ref_losses, ref_params = run(False)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        losses, params = run(True)
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 1.5
    finally:
        dist.destroy_process_group()
    assert losses == ref_losses
    for p, q in zip(params, ref_params):
        assert torch.equal(p, q)


def _two_rank_worker(rank, world, port, out_path):
    import torch.distributed as dist
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")                     # both ranks share the one GPU of the test box (gloo: RCCL refuses that)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                # different init per rank: the broadcast must fix it
        model = UNO_9(3, 8, pad=5).to(dev)
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3, bucket_mb=0.05)
        assert len(tr.grads.buckets) > 3
        a, u = synthetic_darcy_batch(4, 72, 7, dev)
        sl = slice(2 * rank, 2 * rank + 2)
        for _ in range(2):
            tr.step(a[sl], u[sl])
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_device_tensors_equal_single_process(tmp_path):
    """World size 2 with the product kernels: the bucketed all-reduce is issued from the autograd thread on DEVICE gradient
    buffers while the backward pass is still running (the CPU gloo test cannot exercise that).  Two ranks on half batches ==
    one process on the whole batch."""
    import torch.multiprocessing as mp
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out_path = str(tmp_path / "dp2.pt")
    mp.spawn(_two_rank_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    dev = torch.device("cuda:0")
    torch.manual_seed(100)
    model = UNO_9(3, 8, pad=5).to(dev)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(4, 72, 7, dev)
    for _ in range(2):
        tr.step(a, u)
    for k, v in model.state_dict().items():
        r, g = v.detach().cpu(), got[k]
        if r.is_complex():
            r, g = torch.view_as_real(r), torch.view_as_real(g)
        assert float((r - g).norm()) <= 2e-4 * float(r.norm()) + 1e-12, k


def _nccl_worker(rank, world, port, out_path):
    import torch.distributed as dist
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)                 # one process per GPU
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # nccl == RCCL on ROCm (xGMI between the GPUs)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == world
        torch.manual_seed(100 + rank)
        model = UNO_9(3, 8, pad=5).to(dev)
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3, bucket_mb=0.05)          # many small buckets: overlapped issue order
        a, u = synthetic_darcy_batch(2 * world, 72, 7, dev)
        sl = slice(2 * rank, 2 * rank + 2)
        for _ in range(2):
            tr.step(a[sl], u[sl])
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X (RCCL refuses two ranks on one device)")
def test_two_ranks_rccl_equal_single_process(tmp_path):
    """The real multi-GPU path: one process per GPU, RCCL all-reduce (bucketed, overlapped with the backward pass) of the flat
    gradient buffer; N ranks on 2 samples each == one process on the 2 N samples.  Runs wherever >= 2 devices are visible
    (the driver's 8-GPU node); skipped on the 1-GPU test box, where the same arithmetic is covered over gloo above."""
    import torch.multiprocessing as mp
    from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
    world = min(torch.cuda.device_count(), 4)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out_path = str(tmp_path / "dpn.pt")
    mp.spawn(_nccl_worker, args=(world, port, out_path), nprocs=world, join=True)
    got = torch.load(out_path)
    dev = torch.device("cuda:0")
    torch.manual_seed(100)
    model = UNO_9(3, 8, pad=5).to(dev)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    a, u = synthetic_darcy_batch(2 * world, 72, 7, dev)
    for _ in range(2):
        tr.step(a, u)
    for k, v in model.state_dict().items():
        r, g = v.detach().cpu(), got[k]
        if r.is_complex():
            r, g = torch.view_as_real(r), torch.view_as_real(g)
        assert float((r - g).norm()) <= 2e-4 * float(r.norm()) + 1e-12, k


def _bench_selfcheck_worker(rank, world, port, out_path):
    import sys
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = bench.dp_selfcheck(dev, world, rank)      # raises SystemExit on a mismatch
        if rank == 0:
            torch.save(rec, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bench_data_parallel_selfcheck_passes_with_two_ranks(tmp_path):
    """bench.py refuses to time anything with N > 1 ranks before its own sharded-vs-whole-batch gradient comparison passes; run that
    comparison here with two ranks (gloo, sharing the one GPU) so that a broken check cannot wait for the first 8-GPU run to be
    found (the one-process side read its flat buffer before collect())."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out_path = str(tmp_path / "selfcheck.pt")
    mp.spawn(_bench_selfcheck_worker, args=(2, port, out_path), nprocs=2, join=True)
    rec = torch.load(out_path)
    assert rec["ranks"] == 2 and rec["buckets"] > 3 and rec["max_rel_grad_diff_over_2_steps"] < 2e-4


def _workload_worker(rank, world, port, out_path, name):
    import torch.distributed as dist
    from uno_amd.harness import workloads
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")                     # both ranks share the one GPU of the test box (gloo: RCCL refuses that)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = workloads.build(name, dev, batch=4, seed=11, small=True, model_seed=100 + rank, bucket_mb=0.05)
        assert len(w.trainer.grads.buckets) > 3
        for _ in range(2):
            w.step(2 * rank, 2 * rank + 2)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save(workloads.flat_params(w.trainer.model), out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["c4", "c5"])
def test_two_ranks_secondary_workloads_equal_single_process(tmp_path, name):
    """BASELINE.json configs[3] / [4] say "DDP over 8 MI355X": the NS-3D step (Uno3D_T20 + ns3d_loss, reference
    ns_train_3d.py:48-70) and the mixed-precision Darcy step (MixedDarcyTrainer) under FlatGradients with the product kernels,
    world size 2 on the one GPU of the test box (gloo): two ranks on half batches == one process on the whole batch after two
    steps.  (In the mixed step every sample sees the same bf16 roundings in both runs; only float32 summation order differs.)"""
    import torch.multiprocessing as mp
    from uno_amd.harness import workloads
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out_path = str(tmp_path / f"dp_{name}.pt")
    mp.spawn(_workload_worker, args=(2, port, out_path, name), nprocs=2, join=True)
    got = torch.load(out_path)
    w = workloads.build(name, torch.device("cuda:0"), batch=4, seed=11, small=True, model_seed=100)
    for _ in range(2):
        w.step()
    ref = workloads.flat_params(w.trainer.model)
    assert float((got - ref).norm()) <= (2e-4 if name == "c4" else 2e-3) * float(ref.norm())
