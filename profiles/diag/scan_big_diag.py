#!/usr/bin/env python3
"""Diagnostic: the k = 17 / 512 Mbp scan mismatch.  Which part differs: marks, rank, total?  Size or k dependent?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hypo_amd import capi
import oracle
gpu = capi.HypoGpu(0); orc = oracle.Oracle()
for k, n in ((13, 512_000_000), (17, 300_000_000), (17, 512_000_000)):
    rng = np.random.default_rng(k)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    pad = codes.reshape(-1, 2); p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8); del codes, pad
    nw = (1 << (2 * k)) // 64
    bits = rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)
    bits &= rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)
    ds = gpu.device_scan(p4, n, k, bits, kids_cap=1024)
    ds.run()
    w, kids, rank, ns = ds.results()
    ow, okids, orank, ons = orc.solid_scan(p4, n, k, bits, kids_cap=1024)
    pc = int(np.unpackbits(w.view(np.uint8)).sum())
    same = w == ow
    print(f"k={k} n={n}: ns dev {ns} oracle {ons}; popcount(dev words) {pc}; words equal {bool(same.all())}; rank[-1] dev {int(rank[-1])}", flush=True)
    if not same.all():
        bad = np.flatnonzero(~same)
        print("   first/last differing word", bad[0], bad[-1], "count", bad.size, "of", w.size, "; first differing word dev/oracle", hex(int(w[bad[0]])), hex(int(ow[bad[0]])))
        q = w.size // 8
        print("   differing words per eighth:", [int((~same[i*q:(i+1)*q]).sum()) for i in range(8)])
    del ds
