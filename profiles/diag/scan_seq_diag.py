#!/usr/bin/env python3
"""Diagnostic: k = 17 scan is wrong only after earlier scans in the same process.  What is stale?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hypo_amd import capi
import oracle
gpu = capi.HypoGpu(0); orc = oracle.Oracle()
def case(k, n, tag):
    rng = np.random.default_rng(k)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    codes[rng.integers(0, n, size=n // 5000)] = 4
    pad = codes.reshape(-1, 2); p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8); del codes, pad
    nw = (1 << (2 * k)) // 64
    bits = rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)
    bits &= rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)
    cap = n // 4
    ds = gpu.device_scan(p4, n, k, bits, kids_cap=cap)
    print(tag, "ptrs: p4 %x bits %x words %x kids %x rank %x ws %x" % (ds.packed4.data_ptr(), ds.bits.data_ptr(), ds.words.data_ptr(), ds.kids.data_ptr(), ds.rank.data_ptr(), ds.workspace.data_ptr()))
    res = []
    for rep in range(3):
        ds.run()
        w, kids, rank, ns = ds.results()
        res.append((ns, int(np.unpackbits(w.view(np.uint8)).sum()), int(rank[-1])))
    ow, okids, orank, ons = orc.solid_scan(p4, n, k, bits, kids_cap=cap)
    print(tag, f"k={k} n={n}: three runs (ns, popcount(words), rank[-1]) = {res}; oracle {ons}; bits intact {bool((ds.bits.cpu().numpy().view(np.uint64) == bits).all())}; words equal {bool((w == ow).all())}", flush=True)
case(13, 100_000_000, "A")
case(15, 250_000_000, "B")
case(17, 512_000_000, "C")
