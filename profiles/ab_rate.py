#!/usr/bin/env python3
"""A/B probe for kernel variants: per-class kernel times of one library on the C2 batch, concurrently (product schedule) and
one class after the other (HYPO_POA_SEQUENTIAL=1: each class has the GPU to itself).
usage: ab_rate.py <libhypo_gpu*.so> [read_sub]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib, sub):
    import torch
    from hypo_amd import capi, sim
    gpu = capi.HypoGpu(0, path=lib)
    db = gpu.device_batch(sim.window_batch(97078, seed=1000, read_sub=sub))
    for _ in range(3):
        db.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        db.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    db.run()
    enq = time.perf_counter() - t0                      # the call itself: all kernels queued, nothing waited for
    torch.cuda.synchronize()
    gpu.profile_begin(4)
    for _ in range(4):
        db.run()
    torch.cuda.synchronize()
    prof = gpu.profile_read()[-1]
    st = db.stats()
    mode = "sequential" if os.environ.get("HYPO_POA_SEQUENTIAL") else "concurrent"
    print(f"{os.path.basename(lib):24s} {mode:10s} sub={sub}: {dt * 1e3:6.3f} ms/call  kernels [plan, c0..c5, call] = "
          f"{[round(float(x), 2) for x in prof]}  esc={st['n_escalated']}  host call returns after {enq * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    if os.environ.get("HYPO_AB_CHILD"):
        child(sys.argv[1], float(sys.argv[2]))
    else:
        sub = sys.argv[2] if len(sys.argv) > 2 else "0.002"
        for seq in ("", "1"):
            env = dict(os.environ, HYPO_AB_CHILD="1")
            if seq:
                env["HYPO_POA_SEQUENTIAL"] = "1"
            subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1], sub], env=env, check=False)
