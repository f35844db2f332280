# N = 2 data parallel on ONE GPU (both ranks on device 0, gloo): after 2 steps on half batches the parameters must equal those of a
# single process stepping on the concatenated batch - the bucketed all-reduce runs from the autograd thread on device tensors here,
# which the CPU gloo test cannot exercise.  torchrun --nproc-per-node 2 tools/check_dp2.py ; then python tools/check_dp2.py verify
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
dev = torch.device("cuda:0")
S, OUT = 85, "/tmp/dp2_state.pt"
a, u = synthetic_darcy_batch(4, S, 7, dev)
if len(sys.argv) > 1 and sys.argv[1] == "verify":
    torch.manual_seed(100)
    model = UNO_9(3, 16, pad=5).to(dev)
    tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
    for _ in range(2):
        tr.step(a, u)
    ref = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    got = torch.load(OUT)
    worst = 0.0
    for k in ref:
        r, g = (torch.view_as_real(ref[k]), torch.view_as_real(got[k])) if ref[k].is_complex() else (ref[k], got[k])
        worst = max(worst, float((r - g).norm() / r.norm().clamp_min(1e-30)))
    print("max relative parameter difference, 2 ranks vs 1 process:", worst)
    assert worst < 2e-4
    sys.exit(0)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(100 + rank)
model = UNO_9(3, 16, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3, bucket_mb=0.25)
assert len(tr.grads.buckets) > 3
sl = slice(2 * rank, 2 * rank + 2)
for _ in range(2):
    tr.step(a[sl], u[sl])
torch.cuda.synchronize()
if rank == 0:
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, OUT)
dist.barrier()
dist.destroy_process_group()
