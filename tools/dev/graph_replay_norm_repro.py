"""Stock torch ops only: a roll-out loss evaluated ONCE over the stacked predictions (sum over steps of ||pred - y|| / ||y||) captured
with its backward into a HIP graph gives the right value on the first replay and a wrong one (4.0 / inf / nan instead of 64.0)
on every later replay on torch 2.10.0+rocm7.0 - the norm of the tensor that does not require grad comes out wrong.  The per-step
form (harness.ns2d_rollout_loss) replays correctly, so the batched form (78.8 instead of 81.4 ms per NS-2D step) is not used.
python tools/dev/graph_replay_norm_repro.py [orig|sq|last|ynorm_only]"""
import torch, sys
V = sys.argv[1] if len(sys.argv) > 1 else 'orig'
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S, T = 32, 64, 2
w = torch.randn(10, 10, device=dev, requires_grad=True)
xx = torch.randn(B, S, S, 10, device=dev); yy = torch.randn(B, S, S, T, device=dev)
def loss_fn(xx, yy):
    preds = []
    for t in range(T):
        im = (xx @ w)[..., :1] * 1e-3
        preds.append(im.reshape(B, -1, 1))
        xx = torch.cat((xx[..., 1:], im), dim=-1)
    if V == 'orig':
        P = torch.stack(preds, dim=2)
        Y = yy.reshape(B, -1, T, 1)
        diff = torch.linalg.vector_norm(P - Y, ord=2, dim=(1, 3))
        return (diff / torch.linalg.vector_norm(Y, ord=2, dim=(1, 3))).sum()
    if V == 'sq':
        P = torch.stack(preds, dim=2)
        Y = yy.reshape(B, -1, T, 1)
        diff = (P - Y).square().sum(dim=(1, 3)).sqrt()
        return (diff / Y.square().sum(dim=(1, 3)).sqrt()).sum()
    if V == 'last':
        P = torch.stack([q.reshape(B, -1) for q in preds], dim=1)                   # (B, T, pix)
        Y = yy.reshape(B, -1, T, 1).permute(0, 2, 1, 3).reshape(B, T, -1)
        diff = torch.linalg.vector_norm(P - Y, ord=2, dim=2)
        return (diff / torch.linalg.vector_norm(Y, ord=2, dim=2)).sum()
    if V == 'ynorm_only':
        P = torch.stack(preds, dim=2)
        Y = yy.reshape(B, -1, T, 1)
        diff = torch.linalg.vector_norm(P - Y, ord=2, dim=(1, 3))
        return diff.sum()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        w.grad = None
        loss_fn(xx, yy).backward()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
w.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    L = loss_fn(xx, yy)
    L.backward()
out = []
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    gn = float(w.grad.norm())
    bad = [torch.isfinite(t).all() for t in (w, w.grad, xx)]
    out.append((float(L), gn))
print(V, out)
