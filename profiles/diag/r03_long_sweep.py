#!/usr/bin/env python3
"""LONG-window parity sweep on the GPU: seeded C4 mixes (noisy long-read windows of 120-500 bp, 12-45 arms: every row width of the hybrid class) and
fuzzed LONG windows of every flavour through libhypo_gpu and the oracle, byte for byte.  usage: r03_long_sweep.py [minutes=5] [first_seed=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hypo_amd import capi, sim
from hypo_amd.batch import build_batch
import oracle
from test_gpu_fuzz import _window
from sweep_parity_gpu import compare
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gpu = capi.HypoGpu(0); orc = oracle.Oracle()
t_end = time.time() + 60 * minutes
n_long = n_all = 0
while time.time() < t_end:
    b = sim.c4_batch(2000, 3000, seed=seed, long_err=(0.05, 0.10, 0.15)[seed % 3])
    n_all += compare(gpu, orc, b, (5, -4, -8, 3, -5, -4), f"c4 seed={seed}"); n_long += 3000
    rng = np.random.default_rng(100000 + seed)
    for scores in ((5, -4, -8, 3, -5, -4), (5, -4, -8, 1, -1, -1)):
        n_all += compare(gpu, orc, build_batch([_window(rng, True) for _ in range(400)]), scores, f"long fuzz seed={seed} {scores}"); n_long += 400
    seed += 1
    print(f"seed {seed - 1}: {n_long} LONG windows ({n_all} windows) identical so far", flush=True)
print(f"OK: {n_long} LONG windows, {n_all} windows in all, 0 mismatches")
