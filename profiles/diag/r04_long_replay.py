#!/usr/bin/env python3
"""The resident LONG batch of an end-to-end run (HYPO_DUMP_LONG=<file>, DeviceArms::polish_impl) through hypo_gpu_poa_batch_device
again: is the time of hypo_gpu_arms_poa_long the kernel's, and which windows take it?
usage: r04_long_replay.py <dump> [shuffle|sorted|firstN]      (HYPO_GPU_LIB=..._prof.so adds the phase split of class 4)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import abi, capi  # noqa: E402
from hypo_amd.batch import HostBatch  # noqa: E402


def load(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n, na, a2b, d4b = (int(x) for x in raw[:32].view(np.uint64))
    at = 32
    win = raw[at:at + 40 * n].view(abi.WINDOW_DTYPE).copy(); at += 40 * n
    alen = raw[at:at + 4 * na].view(np.uint32).copy(); at += 4 * na
    aoff = raw[at:at + 8 * na].view(np.uint64).copy(); at += 8 * na
    arms2 = raw[at:at + a2b].copy(); at += a2b
    draft4 = raw[at:at + d4b].copy()
    return HostBatch(windows=win, draft4=draft4, arm_off=aoff, arm_len=alen, arms2=arms2)


def main():
    b = load(sys.argv[1])
    mode = sys.argv[2] if len(sys.argv) > 2 else "asis"
    w = b.windows
    narm = (w["n_internal"] + w["n_prefix"] + w["n_suffix"]).astype(np.int64)
    cost = narm * w["draft_len"].astype(np.int64)
    if mode == "shuffle":
        perm = np.random.default_rng(1).permutation(b.n_windows)
        b = HostBatch(windows=w[perm], draft4=b.draft4, arm_off=b.arm_off, arm_len=b.arm_len, arms2=b.arms2)
    elif mode == "sorted":                                   # the most expensive windows first
        perm = np.argsort(-cost, kind="stable")
        b = HostBatch(windows=w[perm], draft4=b.draft4, arm_off=b.arm_off, arm_len=b.arm_len, arms2=b.arms2)
    elif mode.startswith("first"):
        k = int(mode[5:])
        b = HostBatch(windows=w[:k], draft4=b.draft4, arm_off=b.arm_off, arm_len=b.arm_len, arms2=b.arms2)
    print(f"{b.n_windows} windows, {int(narm.sum())} arms; arms per window: mean {narm.mean():.1f}, max {narm.max()}; "
          f"arms x draft per window: mean {cost.mean():.0f}, max {cost.max()}, p99 {np.percentile(cost, 99):.0f}; n_prefix+n_suffix {int((w['n_prefix'] + w['n_suffix']).sum())}, n_empty {int(w['n_empty'].sum())}")
    lib = os.environ.get("HYPO_GPU_LIB")
    gpu = capi.HypoGpu(0, path=lib) if lib else capi.HypoGpu(0)
    db = gpu.device_batch(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); db.run(); torch.cuda.synchronize()
    print(f"[{mode}] FIRST call of the process: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    t0 = time.perf_counter(); db.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = db.stats()
    gpu.profile_begin(2)
    for _ in range(2):
        db.run()
    torch.cuda.synchronize()
    print(f"[{mode}] kernel times of a call, ms [plan, class 0..5, call]:", [round(float(x), 2) for x in gpu.profile_read()[-1]], "escalated", st["n_escalated"])
    print(f"[{mode}] {b.n_windows} LONG windows in {dt * 1e3:.1f} ms; cells {st['dp_cells'] / 1e9:.2f} G -> {st['dp_cells'] / dt / 1e9:.1f} GCUPS; classes {st['n_class'][:6]} failed {st['n_failed']}; "
          f"alignments {st['n_alignments']} reused {st['n_reused']} threaded {st['n_threaded']}")
    if lib and "prof" in lib:
        names = ["load_seq", "dp_rows", "traceback", "add_alignment", "toposort", "consensus", "output", "rowmeta", "exact_rows"]
        NP = len(names)
        ph = db.workspace[512:512 + 8 * 24 * 8].cpu().numpy().view(np.uint64).reshape(8, 24)
        c = 4
        tot = float(ph[c, :NP].sum())
        nw = max(st['n_class'][c], 1)
        D = NP + 2
        print(f"class 4: waves={int(ph[c, NP + 1])} cycles/window={ph[c, NP] / nw / 1e3:.0f}k accounted={100 * tot / float(ph[c, NP]):.1f}%")
        print("class 4 phases: " + ", ".join(f"{n} {100 * ph[c, i] / tot:.1f}%" for i, n in enumerate(names)))
        print(f"    per window: rows={ph[c, D] / nw:.0f} alignments={ph[c, D + 1] / nw:.1f} reused={ph[c, D + 2] / nw:.1f} toposorts={ph[c, D + 3] / nw:.1f}"
              f" toposort {ph[c, 4] / max(ph[c, D + 3], 1) / 1e3:.1f} kcycles each, {ph[c, D + 10] / max(ph[c, D + 3], 1):.0f} DFS steps + {ph[c, D + 11] / max(ph[c, D + 3], 1):.0f} run steps")


if __name__ == "__main__":
    main()
