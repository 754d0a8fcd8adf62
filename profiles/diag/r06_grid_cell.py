#!/usr/bin/env python3
"""One cell of the SURVEY 8(d) grid under several schedules: where do the 8 % cells lose their time?
usage: r06_grid_cell.py length arms err [n_windows]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402


def main():
    length, arms, err = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    nwin = int(sys.argv[4]) if len(sys.argv) > 4 else max(4000, min(60000, 24_000_000 // (length * arms)))
    gpu = capi.HypoGpu(0)
    b = sim.grid_batch(length, arms, nwin, err, seed=11)
    db = gpu.device_batch(b)
    for tag, env, mc in (("default", {}, 0), ("no polling launch", {"HYPO_POA_POLL": "0"}, 0), ("plan waited for", {"HYPO_POA_SYNC_PLAN": "1"}, 0),
                         ("sequential classes", {"HYPO_POA_SEQUENTIAL": "1"}, 0), ("start in class 1", {}, 1), ("start in class 2", {}, 2), ("start in class 3", {}, 3)):
        for k in ("HYPO_POA_POLL", "HYPO_POA_SYNC_PLAN", "HYPO_POA_SEQUENTIAL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        assert gpu.lib.hypo_gpu_set_option(b"poa_min_class", C.c_int(mc)) == 0
        for _ in range(2):
            db.run()
            torch.cuda.synchronize()
        gpu.profile_begin(8)
        t0 = time.perf_counter()
        for _ in range(3):
            db.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        prof = gpu.profile_read()
        s = db.stats()
        print(f"{tag:22s} {dt * 1e3:9.3f} ms  {nwin / dt / 1e6:8.3f} M windows/s  classes {s['n_class'][:6]} re-queued {s['n_escalated']} carried {s['n_carried']} "
              f"kernels ms [plan, c0..c5, call] {[round(x, 2) for x in prof[-1]]}", flush=True)
    gpu.lib.hypo_gpu_set_option(b"poa_min_class", C.c_int(0))


if __name__ == "__main__":
    main()
