import torch
dev = torch.device("cuda:0")
n = 16*64*421*421
x = torch.randn(n, device=dev); y = torch.empty_like(x)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
tf = t(lambda: y.fill_(1.0)); tz = t(lambda: y.zero_()); ts = t(lambda: x.sum()); tc = t(lambda: y.copy_(x)); tm = t(lambda: torch.mul(x, 2.0, out=y))
B = n*4
print(f"fill {tf:.1f} us {B/tf/1e6:.2f} TB/s | zero {tz:.1f} us {B/tz/1e6:.2f} | sum(read) {ts:.1f} us {B/ts/1e6:.2f} | copy {tc:.1f} us {2*B/tc/1e6:.2f} | mul {tm:.1f} us {2*B/tm/1e6:.2f}")
