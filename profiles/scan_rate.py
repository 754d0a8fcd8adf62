#!/usr/bin/env python3
"""Solid-kmer scan at the contig sizes of the larger configurations (HBM-bound kernel): GB/s of algorithmic bytes.
usage: scan_rate.py [n_bases] [k]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    rng = np.random.default_rng(k)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    pad = codes.reshape(-1, 2)
    p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
    bits = rng.integers(0, 1 << 63, size=(1 << (2 * k)) // 64, dtype=np.int64).view(np.uint64)
    bits &= rng.integers(0, 1 << 63, size=bits.size, dtype=np.int64).view(np.uint64)       # ~25 % of k-mers solid
    gpu = capi.HypoGpu(0)
    ds = gpu.device_scan(p4, n, k, bits, kids_cap=n // 3)
    for _ in range(2):
        ds.run()
    torch.cuda.synchronize()
    gpu.profile_begin(8)
    for _ in range(5):
        ds.run()
    torch.cuda.synchronize()
    prof = [p for p in gpu.profile_read() if len(p) == 3]
    ms = np.array(prof).mean(axis=0)
    _, _, _, ns = ds.results()
    a = (n + 1) // 2 + (n + 7) // 8 + 8 * ns + min((1 << (2 * k)) // 8, 32 * (n - k + 1))
    # (round 2: ms = mark, rank, k-mer ids; since round 3 one fused launch behind a memset: everything is in ms[0])
    print(f"scan {n / 1e6:.0f} Mbp k={k}: {ms.sum() * 1e3:.1f} us (memset + scan_fused_kernel); {ns} solid positions; "
          f"algorithmic {a / 1e6:.0f} MB -> {a / ms.sum() / 1e6:.0f} GB/s ({a / ms.sum() / 1e6 / 8000:.3f} of 8 TB/s)")

if __name__ == "__main__":
    main()
