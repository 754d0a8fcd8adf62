#!/usr/bin/env python3
"""POA rate vs read error rate on the C1-shaped batch (97k windows): how the re-queued ("escalated") windows scale.
usage: err_rate.py <libhypo_gpu*.so> [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402


def main():
    lib = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = 97078
    gpu = capi.HypoGpu(0, path=lib)
    for sub in (0.002, 0.005, 0.01, 0.02, 0.03, 0.05):
        db = gpu.device_batch(sim.window_batch(n, seed=1000, read_sub=sub))
        db.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        db.run()
        torch.cuda.synchronize()
        cold = time.perf_counter() - t0            # second call: first with history of the same kind of batch
        t0 = time.perf_counter()
        for _ in range(steps):
            db.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = db.stats()
        gpu.profile_begin(4)
        for _ in range(4):
            db.run()
        torch.cuda.synchronize()
        prof = gpu.profile_read()[-1]
        print("   per-kernel ms [plan, class0..5, whole call]:", [round(float(x), 2) for x in prof])
        print(f"read_sub={sub}: {dt * 1e3:.2f} ms/step (2nd call {cold * 1e3:.2f})  {n / dt / 1e6:.2f} M windows/s  "
              f"classes={st['n_class']} esc={st['n_escalated']} failed={st['n_failed']}", flush=True)


if __name__ == "__main__":
    main()
