#!/usr/bin/env python3
"""POA rate on the window mix of BASELINE config C4 (short reads + noisy long reads): C1-shaped SHORT windows and LONG windows
(120-500 bp, 12-45 arms, 10 % errors incl. indels) in one batch.  usage: c4_rate.py [n_short] [n_long]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402


def main():
    n_short = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    n_long = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    gpu = capi.HypoGpu(0)
    for ns, nl in ((n_short, n_long), (n_short, 0), (0, n_long)):
        if ns + nl == 0:
            continue
        db = gpu.device_batch(sim.c4_batch(ns, nl, seed=404))
        db.run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); db.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        gpu.profile_begin(1); db.run(); torch.cuda.synchronize()
        st = db.stats()
        print(f"{ns} SHORT + {nl} LONG windows: {dt * 1e3:.1f} ms = {(ns + nl) / dt / 1e3:.0f} k windows/s; kernels [plan, class 0..5, call] ms = "
              f"{[round(float(x), 2) for x in gpu.profile_read()[-1]]}; classes {st['n_class'][:6]} failed {st['n_failed']}", flush=True)
    import oracle
    orc = oracle.Oracle()
    sub = sim.c4_batch(n_short // 20, n_long // 20, seed=404)
    t0 = time.perf_counter(); orc.poa_batch_raw(sub); dt = time.perf_counter() - t0
    print(f"CPU oracle ({orc.num_threads()} threads): {sub.n_windows} windows of the same mix in {dt * 1e3:.0f} ms = {sub.n_windows / dt / 1e3:.1f} k windows/s")


if __name__ == "__main__":
    main()
