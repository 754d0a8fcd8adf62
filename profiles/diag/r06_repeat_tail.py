#!/usr/bin/env python3
"""Which windows of the non-i.i.d. batch set the length of the call: the costliest windows (rows x arms), each alone in a batch."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import e2e_util as eu  # noqa: E402
from hypo_amd import capi, dist  # noqa: E402

d = tempfile.mkdtemp(prefix="hypo_real_")
b, cons, man, rr = eu.realistic_window_batch(d)
w = b.windows
narm = w["n_internal"].astype(np.int64) + w["n_prefix"] + w["n_suffix"]
cost = (w["draft_len"].astype(np.int64) + 2) * (narm + 1)
order = np.argsort(-cost)[:24]
gpu = capi.HypoGpu(0)
for i in order:
    sb = dist.take_windows(b, int(i), int(i) + 1)
    db = gpu.device_batch(sb)
    db.run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); db.run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    st = db.stats()
    print(f"window {int(i):6d} type {int(w['type'][i])} draft {int(w['draft_len'][i]):4d} arms {int(narm[i]):4d} (int {int(w['n_internal'][i])} pre {int(w['n_prefix'][i])} suf {int(w['n_suffix'][i])})"
          f"  alone: {min(ts):6.2f} ms  class counts {st['n_class']} escalated {st['n_escalated']} cells {st['dp_cells']} scored {st['cells_scored']} aligns {st['n_alignments']} reused {st['n_reused']} threaded {st['n_threaded']}", flush=True)
