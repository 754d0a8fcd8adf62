#!/usr/bin/env python3
"""Rate on the dense-SR window shape of the C4/C5 configurations (tiny windows, 45 % <= 8 bp,
~22 arms of about the window's length): everything lands in size class 0.   usage: dense_rate.py [n_windows]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rng = np.random.default_rng(3)
    wl = rng.choice([3, 5, 8, 12, 16, 24, 32, 48, 64, 99], size=n, p=[.15, .15, .15, .13, .13, .1, .1, .05, .03, .01])
    shapes = np.stack([wl, rng.integers(3, 45, size=n), np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)], axis=1)
    b = sim.window_batch(n, seed=9, shapes=shapes, read_sub=0.002)
    lib = os.environ.get("HYPO_GPU_LIB")
    gpu = capi.HypoGpu(0, path=lib) if lib else capi.HypoGpu(0)
    db = gpu.device_batch(b)
    for _ in range(2):
        db.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        db.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = db.stats()
    print(f"dense-SR shape: {n} windows, {b.n_arms} arms in {dt * 1e3:.2f} ms = {n / dt / 1e6:.1f} M windows/s; {st['dp_cells'] / dt / 1e9:.0f} GCUPS; "
          f"classes {st['n_class'][:6]} trivial {st['n_trivial']}")
    if os.environ.get("HYPO_CPU", "1") == "1":
        import oracle
        orc = oracle.Oracle()
        ns = min(n, 400000)
        sub = sim.window_batch(ns, seed=9, shapes=shapes[:ns], read_sub=0.002)
        best = None
        for _ in range(2):
            c0 = time.perf_counter(); orc.poa_batch_raw(sub); c1 = time.perf_counter() - c0
            best = c1 if best is None or c1 < best else best
        print(f"CPU restatement ({orc.num_threads()} threads): {ns} windows in {best * 1e3:.0f} ms = {ns / best / 1e6:.2f} M windows/s -> GPU/CPU = {(n / dt) / (ns / best):.1f}x")
    if lib and "prof" in lib:
        names = ["load_seq", "dp_rows", "traceback", "add_alignment", "toposort", "consensus", "output", "rowmeta"]
        ph = db.workspace[512:512 + 8 * 16 * 8].cpu().numpy().view(np.uint64).reshape(8, 16)
        for c in range(3):
            tot = float(ph[c, :8].sum())
            if tot:
                print(f"class {c}: " + ", ".join(f"{nm} {100 * ph[c, i] / tot:.1f}%" for i, nm in enumerate(names)),
                      f"; real alignments/window {ph[c, 11] / max(st['n_class'][c], 1) / 7:.2f} reused {ph[c, 12] / max(st['n_class'][c], 1) / 7:.2f}")


if __name__ == "__main__":
    main()
