#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point hypo_gpu_poa_batch (H2D + kernels + D2H + per-call
hipMalloc) on the C2 workload — the number DESIGN.md quotes beside the HBM-resident bench value."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hypo_amd import capi, sim
gpu = capi.HypoGpu(0)
b = sim.window_batch(97078, seed=1000)
off = b.slot_layout()
gpu.poa_batch(b, off=off)
t = []
for _ in range(5):
    t0 = time.perf_counter(); gpu.poa_batch(b, off=off); t.append(time.perf_counter() - t0)
best = min(t)
h2d = b.windows.nbytes + b.draft4.nbytes + b.arm_off.nbytes + b.arm_len.nbytes + b.arms2.nbytes + off.nbytes
d2h = int(off[-1]) + 5 * b.n_windows
print(f"host API: {b.n_windows / best:.0f} windows/s ({best * 1e3:.2f} ms per batch; H2D {h2d / 1e6:.1f} MB, D2H {d2h / 1e6:.1f} MB)")
