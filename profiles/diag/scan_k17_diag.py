#!/usr/bin/env python3
"""Diagnostic for the k = 17 scan mismatch: is the 2 GiB bit set intact on the device, and which k-mer ids are misjudged?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hypo_amd import capi
import oracle

gpu = capi.HypoGpu(0)
orc = oracle.Oracle()
for k in (16, 17):
    n = 20_000_000
    rng = np.random.default_rng(k)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    pad = codes.reshape(-1, 2)
    p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
    nw = (1 << (2 * k)) // 64
    bits = rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)
    ds = gpu.device_scan(p4, n, k, bits, kids_cap=n // 4)
    back = ds.bits.cpu().numpy().view(np.uint64)
    print(f"k={k}: bit set on device intact: {bool((back == bits).all())}", flush=True)
    q = nw // 4
    for i in range(4):
        print("   quarter", i, "host sum", int(bits[i*q:(i+1)*q].view(np.int64)[::4097].sum()), "dev sum", int(back[i*q:(i+1)*q].view(np.int64)[::4097].sum()))
    ds.run()
    w, kids, rank, ns = ds.results()
    ow, okids, orank, ons = orc.solid_scan(p4, n, k, bits, kids_cap=n // 4)
    print(f"k={k}: device {ns} oracle {ons}", flush=True)
    if ns != ons:
        # which positions differ, and what are their k-mer ids
        d = np.flatnonzero(np.unpackbits((w ^ ow).view(np.uint8), bitorder="little")[:n])[:2000]
        kid = np.zeros(d.size, dtype=np.uint64)
        for t in range(k):
            kid = (kid << np.uint64(2)) | codes[d + t].astype(np.uint64)
        print("   differing positions:", d.size, "top 2 bits of their ids:", np.bincount((kid >> np.uint64(2 * k - 2)).astype(np.int64), minlength=4))
        print("   bits 32-33 of ids:", np.bincount(((kid >> np.uint64(32)) & np.uint64(3)).astype(np.int64), minlength=4))
