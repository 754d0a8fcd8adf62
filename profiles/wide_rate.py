#!/usr/bin/env python3
"""Rate on wide SHORT windows (128-200 bp: the reference force-divides a weak region only above 2 x 100 bp), size class 3.
usage: wide_rate.py [length] [arms] [n_windows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    arms = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    b = sim.grid_batch(length, arms, n, 0.005, seed=3)
    gpu = capi.HypoGpu(0, path=os.environ["HYPO_GPU_LIB"]) if os.environ.get("HYPO_GPU_LIB") else capi.HypoGpu(0)
    db = gpu.device_batch(b)
    db.run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        db.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = db.stats()
    print(f"{n} windows of {length} bp x {arms} arms: {dt * 1e3:.2f} ms = {n / dt / 1e6:.2f} M windows/s, {st['dp_cells'] / dt / 1e9:.0f} GCUPS; classes {st['n_class'][:6]} esc {st['n_escalated']} failed {st['n_failed']}")


if __name__ == "__main__":
    main()
