"""The reference-style Darcy caller (harness/reference_style.py) on the product blocks: ms per step with the blocks' layout rule off / on,
the transposing copy's own bandwidth, and (argument `prof`) the device-time table of one step per mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import uno_amd.integral_operators as uio
from uno_amd import _native
from uno_amd.harness import DarcyTrainer, UNO_9_ReferenceStyle, synthetic_darcy_batch
dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for shape in [(16, 32, 446, 446), (16, 128, 223, 223), (16, 256, 111, 111), (16, 64, 446, 446), (16, 96, 421, 421)]:
    B, C = shape[:2]
    x = torch.randn(B, *shape[2:], C, device=dev).movedim(-1, 1)
    y = x.contiguous()
    mb = x.numel() * 8 / 1e6
    t1 = timeit(lambda: _native.to_channels_first(x)); t2 = timeit(lambda: _native.to_channels_last(y)); t3 = timeit(lambda: x.contiguous())
    print(f"{shape}: to_channels_first {t1*1e3:.0f} us ({mb/t1/1e3:.2f} TB/s)  to_channels_last {t2*1e3:.0f} us ({mb/t2/1e3:.2f} TB/s)  torch .contiguous() {t3*1e3:.0f} us", flush=True)
    del x, y

torch.manual_seed(0)
model = UNO_9_ReferenceStyle(3, 64, pad=5).to(dev)
tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
a, u = synthetic_darcy_batch(16, 421, 1234, dev)
for mode in (False, True):
    uio.CHANNELS_LAST_IO = mode
    ms = timeit(lambda: tr.step(a, u), n=10, warm=4)
    print(f"CHANNELS_LAST_IO={mode}: {ms:.2f} ms/step = {16/ms*1e3:.1f} samples/s", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "prof":
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            tr.step(a, u); torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70), flush=True)
