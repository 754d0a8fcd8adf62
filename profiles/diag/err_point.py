#!/usr/bin/env python3
"""One point of the read-error table (profiles/err_rate.py) for sweeps of launch knobs.
usage: err_point.py <read_sub> [steps] [lib]     (knobs come from the environment: HYPO_POA_CAPS, HYPO_POA_POLL, HYPO_POA_POLL_WAVES ...)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402


def main():
    sub = float(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    lib = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "hypo_amd", "_build", "libhypo_gpu.so")
    n = 97078
    gpu = capi.HypoGpu(0, path=lib)
    db = gpu.device_batch(sim.window_batch(n, seed=1000, read_sub=sub))
    for _ in range(3):
        db.run()
    torch.cuda.synchronize()
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
    knobs = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("HYPO_POA"))
    print(f"sub={sub} [{knobs}] {dt * 1e3:.2f} ms  {n / dt / 1e6:.2f} M/s  kernels={[round(float(x), 2) for x in prof]} "
          f"classes={st['n_class'][:5]} esc={st['n_escalated']} carried={st.get('n_carried')}", flush=True)


if __name__ == "__main__":
    main()
