#!/usr/bin/env python3
"""What the FIRST device-resident POA call of a process costs beyond a steady-state one (a `hypo` run on one contig batch makes exactly one):
per call, with a synchronisation behind each.  usage: r03_first_device_call.py [windows=97078]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hypo_amd import capi, sim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 97078
t0 = time.perf_counter(); gpu = capi.HypoGpu(0); print(f"hypo_gpu_init: {(time.perf_counter() - t0) * 1e3:.1f} ms")
b = sim.window_batch(n, seed=1000)
t0 = time.perf_counter(); db = gpu.device_batch(b); torch.cuda.synchronize(); print(f"device_batch (upload): {(time.perf_counter() - t0) * 1e3:.1f} ms")
for i in range(4):
    t0 = time.perf_counter(); db.run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"call {i}: enqueue {(t1 - t0) * 1e3:.2f} ms, done after {(t2 - t0) * 1e3:.2f} ms")
