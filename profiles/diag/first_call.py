#!/usr/bin/env python3
"""What the FIRST hypo_gpu_poa_batch call of a process costs beyond a steady-state one (code object load, first touch of fresh
device memory, occupancy queries): the end-to-end binary makes exactly one such call per contig batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hypo_amd import capi, sim
gpu = capi.HypoGpu(0)
small = sim.window_batch(64, seed=5)
b = sim.window_batch(97078, seed=1000)
off = b.slot_layout()
if len(sys.argv) > 1 and sys.argv[1] == "warm":      # a tiny call first: what loading the code objects and the first occupancy queries cost
    t0 = time.perf_counter(); gpu.poa_batch(small, off=small.slot_layout()); print(f"tiny warm-up call: {(time.perf_counter() - t0) * 1e3:.2f} ms")
for i in range(3):
    t0 = time.perf_counter(); gpu.poa_batch(b, off=off); print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
