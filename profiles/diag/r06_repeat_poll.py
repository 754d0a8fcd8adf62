#!/usr/bin/env python3
"""value_repeat's batch (the real windows of the non-i.i.d. set) under several sizes of class 3's polling launch."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import e2e_util as eu  # noqa: E402
from hypo_amd import capi  # noqa: E402

d = tempfile.mkdtemp(prefix="hypo_real_")
b, cons, man, rr = eu.realistic_window_batch(d)
gpu = capi.HypoGpu(0)
dbs = [gpu.device_batch(b) for _ in range(2)]
for env in ({}, {"HYPO_POA_POLL": "0"}):
    for k in ("HYPO_POA_POLL_WAVES", "HYPO_POA_POLL", "HYPO_POA_SEQUENTIAL", "HYPO_POA_ORDER", "HYPO_POA_CAPS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for x in dbs:
        x.run()
    torch.cuda.synchronize()
    ts = []
    gpu.profile_begin(12)
    for i in range(12):
        t0 = time.perf_counter()
        dbs[i % 2].run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    for t, pr in zip(ts, gpu.profile_read()):
        print("   ", round(t, 2), [round(x, 2) for x in pr], flush=True)
    print(env or "default", "min / median / max ms:", round(min(ts), 2), round(sorted(ts)[6], 2), round(max(ts), 2), flush=True)
