#!/usr/bin/env python3
"""POA-only rate of one build of the library on the C2 batch (diagnostic for kernel variants).
usage: poa_rate.py <libhypo_gpu*.so> [n_windows] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402


def main():
    lib = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 97078
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    gpu = capi.HypoGpu(0, path=lib)
    db = gpu.device_batch(sim.window_batch(n, seed=1000))
    for _ in range(3):
        db.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        db.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = db.stats()
    print(f"{os.path.basename(lib)}: {dt * 1e3:.3f} ms/step  {n / dt / 1e6:.2f} M windows/s  classes={st['n_class'][:5]} esc={st['n_escalated']} failed={st['n_failed']}")


if __name__ == "__main__":
    main()
