"""C5 block (f32 + mixed) and the C5 mixed model, with per-kernel times: python tools/dev/c5run.py [block] [model]"""
import json, os, sys
sys.path.insert(0, '.')
import torch
import bench
from uno_amd import _native
dev = torch.device("cuda:0")
what = sys.argv[1:] or ["block", "model"]
if "block" in what:
    # reuse bench's definition
    out = {}
    import inspect
    ex = bench.extra_workloads.__code__
    # bench.extra_workloads runs everything; replicate c5_block here
    g = torch.Generator().manual_seed(0)
    C, S5, m, B = 64, 1024, 32, 4
    x = torch.randn(B, C, S5, S5, generator=g).to(dev)
    sc = (1 / (2 * C)) ** 0.5
    w1 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
    w2 = (sc * torch.randn(C, C, m, m, dtype=torch.cfloat, generator=g)).to(dev)
    gy = torch.randn(B, C, S5, S5, generator=g).to(dev)
    xb, gyb = x.bfloat16(), gy.bfloat16()
    y, xt = _native.spectral_conv2d_forward(x, w1, w2, S5, S5)
    del x, gy
    w1h, w2h = (torch.view_as_real(w).half().contiguous() for w in (w1, w2))
    yb, xtb = _native.spectral_conv2d_forward(xb, w1h, w2h, S5, S5)
    tf = bench._timed(lambda: _native.spectral_conv2d_forward(xb, w1h, w2h, S5, S5), dev, iters=5, reps=3, warm=2)
    tb = bench._timed(lambda: _native.spectral_conv2d_backward(gyb, xtb, w1h, w2h, S5, S5), dev, iters=5, reps=3, warm=2)
    img, wb = B * C * S5 * S5 * 4, 2 * C * C * m * m * 8
    imgb, wh = img // 2, wb // 2
    print(f"C5 mixed block: fwd {tf*1e6:.1f} us = {(2*imgb+wh)/tf/8e12:.3f} of 8 TB/s, bwd {tb*1e6:.1f} us = {(2*imgb+wh+wb)/tb/8e12:.3f}; rel err vs f32 {float((yb.float()-y).norm()/y.norm()):.2e}")
    for name, fn in (("fwd", lambda: _native.spectral_conv2d_forward(xb, w1h, w2h, S5, S5)), ("bwd", lambda: _native.spectral_conv2d_backward(gyb, xtb, w1h, w2h, S5, S5))):
        fn(); torch.cuda.synchronize()
        _native.profile_begin(64)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        rec = _native.profile_end()
        agg = {}
        for k, ms, by in rec:
            agg.setdefault(k, []).append(ms)
        print("  ", name, {k: round(sum(v) / len(v) * 1e3, 1) for k, v in agg.items()})
if "model" in what:
    from uno_amd.harness.mixed import c5_mixed_model_bench
    r = c5_mixed_model_bench(dev)
    print("C5 mixed model:", round(r["ms_per_step"], 2), "ms/step", round(r["peak_mem_GiB"], 2), "GiB")
    if "prof" in what:
        from uno_amd.harness import UNO_9, synthetic_darcy_batch
        from uno_amd.harness.mixed import MixedDarcyTrainer
        for cls_name in ("mixed", "f32"):
            torch.manual_seed(0)
            model = UNO_9(3, 64, pad=5).to(dev)
            from uno_amd.harness import DarcyTrainer
            tr = (MixedDarcyTrainer if cls_name == "mixed" else DarcyTrainer)(model, lr=1e-3, weight_decay=1e-3)
            a, u = synthetic_darcy_batch(4, 1024, 1234, dev)
            for _ in range(2):
                tr.step(a, u)
            torch.cuda.synchronize()
            _native.profile_begin(4096)
            tr.step(a, u)
            torch.cuda.synchronize()
            rec = _native.profile_end()
            agg = {}
            for k, ms, by in rec:
                e = agg.setdefault(k, [0, 0.0, 0.0]); e[0] += 1; e[1] += ms; e[2] += by
            tot = sum(v[1] for v in agg.values())
            print(f"--- {cls_name}: {len(rec)} launches, sum of kernel times {tot:.2f} ms")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
                print(f"   {k:58s} n={v[0]:3d} {v[1]:7.3f} ms  {v[2] / max(v[1], 1e-9) / 1e9:7.2f} TB/s")
            del tr, model
    if "f32" in what:
        from uno_amd.harness import DarcyTrainer, UNO_9, synthetic_darcy_batch
        torch.manual_seed(0)
        model = UNO_9(3, 64, pad=5).to(dev)
        tr = DarcyTrainer(model, lr=1e-3, weight_decay=1e-3)
        a, u = synthetic_darcy_batch(4, 1024, 1234, dev)
        print("C5 f32 model:", round(bench._train_ms(lambda: tr.step(a, u), dev, steps=4, warmup=2), 2), "ms/step")
