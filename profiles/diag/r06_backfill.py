#!/usr/bin/env python3
"""Backfill launches of class 2 behind classes 0 / 1 (HYPO_POA_BACKFILL) and the polling launch's waves per CU (HYPO_POA_CAPS[3]):
median ms per call on the non-i.i.d. batch and on C2-shaped batches at several read-error rates."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import e2e_util as eu  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402

gpu = capi.HypoGpu(0)
d = tempfile.mkdtemp(prefix="hypo_real_")
real = eu.realistic_window_batch(d)[0]
sets = {"real": [gpu.device_batch(real) for _ in range(2)]}
for sub in (0.002, 0.005, 0.01, 0.02):
    sets[f"sub{sub}"] = [gpu.device_batch(sim.window_batch(97078, seed=s, read_sub=sub)) for s in (1000, 5000)]
ENVS = [{}] + [{"HYPO_POA_BACKFILL": v} for v in sys.argv[1].split(";") if v] + [{"HYPO_POA_CAPS": v} for v in sys.argv[2].split(";") if v]
KEYS = ("HYPO_POA_BACKFILL", "HYPO_POA_CAPS")
print("env".ljust(40), *[k.rjust(18) for k in sets])
for env in ENVS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    row = []
    for name, dbs in sets.items():
        for x in dbs:
            x.run()
            x.run()
        torch.cuda.synchronize()
        ts = []
        for i in range(16):
            t0 = time.perf_counter()
            dbs[i % 2].run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        row.append(f"{ts[0]:.2f}/{ts[8]:.2f}/{ts[-1]:.2f}")
    print(str(env or "default").ljust(40), *[r.rjust(18) for r in row], flush=True)
