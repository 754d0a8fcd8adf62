#!/usr/bin/env python3
"""The 1 % read-error point moves between 5.1 and 7.1 ms from run to run: per-call times and per-kernel times of 40 consecutive calls."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402

gpu = capi.HypoGpu(0)
SUB = float(os.environ.get("SUB", "0.01"))
dbs = [gpu.device_batch(sim.window_batch(97078, seed=s, read_sub=SUB)) for s in (1000, 5000)]
for d in dbs:
    d.run()
torch.cuda.synchronize()
gpu.profile_begin(64)
ts = []
for i in range(40):
    t0 = time.perf_counter()
    dbs[i % 2].run()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
prof = gpu.profile_read()
for i, (t, p) in enumerate(zip(ts, prof)):
    print(f"call {i:2d}: {t:6.2f} ms  kernels [plan, c0..c5, call] {[round(x, 2) for x in p]}")
print("min / median / max ms:", round(min(ts), 2), round(sorted(ts)[len(ts) // 2], 2), round(max(ts), 2))
