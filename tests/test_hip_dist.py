"""RCCL smoke test of the data-parallel step on ONE GPU: a 1-rank "nccl" process group with the collectives
forced on (broadcast of parameters, all-reduce of the flat gradient buffer incl. complex grads, barrier) must
reproduce the plain single-process step bit for bit.  The multi-rank arithmetic is covered on CPU with gloo
(tests/test_harness_cpu.py); the multi-GPU run itself is the driver's.  pytest -m gpu"""
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
