#!/usr/bin/env python
"""Thread sweep of bench.py's CPU-baseline legs on THIS host -> profiles/<tag>_cpu_threads.txt.

    python tools/cpu_thread_sweep.py r06

BASELINE.md section 3 asks for torch.set_num_threads(os.cpu_count()); on the 256-thread GPU-box host that is the slowest
setting (oversubscribed FFT / GEMM pools), so bench.py runs its legs at the best setting of this sweep and cites this file.
Every point is bench.py's own child process (`--cpu-baseline-only`), killed by PID after its time limit (DNF)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    hw = os.cpu_count() or 1
    lines = [f"CPU thread sweep of bench.py's cpu_baseline legs; host: {bench._cpu_model()}, os.cpu_count() = {hw}",
             "config  threads  batch  steps  s/step      samples/s   (DNF = killed at the limit)"]
    plan = [("c1", t, bench.C1_BATCH, 5, 90) for t in sorted({1, 4, 8, 16, 32, 64, 128, hw}) if t <= hw]
    plan += [("c2", t, 16, 1, 150) for t in sorted({8, 16, 32, 64, hw}) if t <= hw]
    for config, threads, batch, steps, limit in plan:
        t0 = time.time()
        r = bench._cpu_child(batch, steps, threads, limit, config)
        if r:
            lines.append(f"{config:6s}  {threads:7d}  {batch:5d}  {steps:5d}  {r['s_per_step']:9.3f}  {r['samples_per_s']:10.3f}")
        else:
            lines.append(f"{config:6s}  {threads:7d}  {batch:5d}  {steps:5d}  DNF after {time.time() - t0:.0f} s (limit {limit} s)")
        print(lines[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    out = os.path.join(ROOT, "profiles", f"{tag}_cpu_threads.txt")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
