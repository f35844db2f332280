import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from uno_amd import _native
if len(sys.argv) > 1 and sys.argv[1] != '-':
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
g = torch.Generator().manual_seed(5)
for (B, C1, C2, Co, act, rows) in ((3, 128, 0, 64, 0, 130), (3, 64, 64, 64, 0, 130), (3, 64, 64, 64, 0, 128)):
    cols, pitch, H = 264, 270, 133
    P, Ci = H * pitch, C1 + C2
    x1 = torch.randn(B, C1, P, generator=g).cuda()
    x2 = torch.randn(B, C2, P, generator=g).cuda() if C2 else None
    w = (torch.randn(Co, Ci, generator=g) / 11).cuda()
    pre, w2, gout = torch.randn(B, Co, P, generator=g).cuda(), torch.randn(Co, generator=g).cuda(), torch.randn(B, P, generator=g).cuda()
    pv = pre.view(B, Co, H, pitch)[:, :, :rows, :cols].double()
    gv = gout.view(B, 1, H, pitch)[:, :, :rows, :cols].double()
    ref = (F.gelu(pv) * gv).sum((0, 2, 3))
    L = _native.lib()
    nws = L.uno_project_backward_ws_bytes(B, Ci, Co, rows * cols)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    for run in range(int(os.environ.get("PB_RUNS", "12"))):
        g1, g2 = torch.zeros(B, C1, P).cuda(), torch.zeros(B, max(C2, 1), P).cuda()
        gw, gb, gw2, gb2 = torch.empty(Co, Ci).cuda(), torch.empty(Co).cuda(), torch.empty(Co).cuda(), torch.empty(1).cuda()
        ws = torch.full((nws // 4,), 0.0).cuda()
        rc = L.uno_project_backward(p(x1), p(x2), C1, p(w), p(pre), p(w2), p(gout), p(g1), p(g2 if C2 else None), p(gw), p(gb), p(gw2), p(gb2), p(ws), B, Ci, Co,
                                    rows, cols, pitch, P, act, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        err = (gw2.double() - ref).abs()
        bad = (err > 1e-3 * ref.abs().max()).nonzero().flatten().tolist()
        # which split's partial is off: recompute the sum of part2 from ws
        nsplit = (nws // 4 - 0)  # unknown here; print raw
        print(sys.argv[1:], B, C1, C2, Co, act, rows, "run", run, "bad rows", bad, [round(float(err[i]), 3) for i in bad][:8], flush=True)
