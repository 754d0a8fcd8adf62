import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
from hypo_amd import capi, sim
n = 500000
rng = np.random.default_rng(5)
wl = rng.choice([3, 5, 8, 12, 16, 24, 32, 48, 64, 99], size=n, p=[.15, .15, .15, .13, .13, .1, .1, .05, .03, .01])
shapes = np.stack([wl, rng.integers(44, 58, size=n), np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)], axis=1)
b = sim.window_batch(n, seed=9, shapes=shapes, read_sub=0.001)
gpu = capi.HypoGpu(0, path=os.environ["HYPO_GPU_LIB"]) if os.environ.get("HYPO_GPU_LIB") else capi.HypoGpu(0)
db = gpu.device_batch(b)
for _ in range(2): db.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): db.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
st = db.stats()
print(f"HiFi-like shape: {n} windows, {b.n_arms} arms in {dt*1e3:.2f} ms = {n/dt/1e6:.1f} M windows/s; classes {st['n_class'][:6]} esc {st['n_escalated']}")
