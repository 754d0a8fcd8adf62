#!/usr/bin/env python3
"""Rate of LONG windows (size class 4, state in HBM scratch): the 163 real LONG windows of tests/golden replicated.
usage: long_rate.py [replicas]            (163 x replicas real LONG windows)
       long_rate.py <n> c4                 (n noisy LONG windows of the C4 mix, sim.c4_batch: 120-500 bp, 12-45 arms, 10 % errors)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import golden_util as gu  # noqa: E402
from hypo_amd import capi  # noqa: E402
from hypo_amd.batch import build_batch  # noqa: E402


def main():
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    recs = [r for r in gu.load_jsonl("windows_real_long.jsonl.gz") if r["long"]]
    wins = [gu.to_window(r) for r in recs] * rep
    b = build_batch(wins)
    if len(sys.argv) > 2 and sys.argv[2] == "c4":
        from hypo_amd import sim
        if len(sys.argv) > 7:                             # long_rate.py <n> c4 <len_lo> <len_hi> <arms_lo> <arms_hi> <err>: the shapes of an end-to-end run
            b = sim.c4_batch(0, rep, seed=404, long_err=float(sys.argv[7]), long_len=(int(sys.argv[3]), int(sys.argv[4])), long_arms=(int(sys.argv[5]), int(sys.argv[6])))
        else:
            b = sim.c4_batch(0, rep, seed=404)
        wins = [None] * rep
    lib = os.environ.get("HYPO_GPU_LIB")
    gpu = capi.HypoGpu(0, path=lib) if lib else capi.HypoGpu(0)
    db = gpu.device_batch(b)
    db.run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); db.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = db.stats()
    gpu.profile_begin(2)
    for _ in range(2):
        db.run()
    torch.cuda.synchronize()
    print("kernel times of a call, ms [plan, class 0..5, call]:", [round(float(x), 2) for x in gpu.profile_read()[-1]], "escalated", st["n_escalated"])
    print(f"GPU: {len(wins)} LONG windows in {dt * 1e3:.1f} ms = {len(wins) / dt:.0f} windows/s; cells {st['dp_cells'] / 1e9:.2f} G -> {st['dp_cells'] / dt / 1e9:.1f} GCUPS; classes {st['n_class'][:5]} failed {st['n_failed']}")
    if lib and "prof" in lib:
        import numpy as np
        names = ["load_seq", "dp_rows", "traceback", "add_alignment", "toposort", "consensus", "output", "rowmeta", "exact_rows"]
        NP = len(names)                                   # then lifetime, waves, 10 counters (poa_kernel.hip)
        ph = db.workspace[512:512 + 6 * 32 * 8].cpu().numpy().view(np.uint64).reshape(6, 32)
        c = 4
        tot = float(ph[c, :NP].sum())
        nw = max(st['n_class'][c], 1)
        D = NP + 2
        print(f"class 4: waves={int(ph[c, NP + 1])} cycles/window={ph[c, NP] / nw / 1e3:.0f}k accounted={100 * tot / float(ph[c, NP]):.1f}%")
        print("class 4 phases: " + ", ".join(f"{n} {100 * ph[c, i] / tot:.1f}%" for i, n in enumerate(names)))
        print(f"    per window: rows={ph[c, D] / nw:.0f} alignments={ph[c, D + 1] / nw:.1f} reused={ph[c, D + 2] / nw:.1f} toposorts={ph[c, D + 3] / nw:.1f}"
              f" serial consensus={ph[c, D + 4] / nw:.2f}; score rows {ph[c, D + 9] / nw:.0f} at {ph[c, 1] / max(ph[c, D + 9], 1):.0f} cycles/row;"
              f" toposort {ph[c, 4] / max(ph[c, D + 3], 1) / 1e3:.1f} kcycles each, {ph[c, D + 10] / max(ph[c, D + 3], 1):.0f} DFS steps + {ph[c, D + 11] / max(ph[c, D + 3], 1):.0f} run steps")
        rows = max(float(ph[c, D + 9]), 1.0)
        print(f"    rows by kind (of the scored rows): several predecessors {100 * ph[c, D + 5] / rows:.1f} % (three or more {100 * ph[c, D + 12] / rows:.1f} %), one predecessor that is not the row before "
              f"{100 * ph[c, D + 8] / rows:.1f} %, first predecessor beyond the LDS ring {100 * ph[c, D + 13] / rows:.1f} %, rows with the end-cell check {100 * ph[c, D + 6] / rows:.1f} %")
    import oracle
    orc = oracle.Oracle()
    if wins[0] is None:
        return
    sub = build_batch(wins[:len(recs) * min(rep, 2)])
    t0 = time.perf_counter(); orc.poa_batch_raw(sub); dt2 = time.perf_counter() - t0
    print(f"CPU oracle ({orc.num_threads()} threads): {sub.n_windows} windows in {dt2 * 1e3:.0f} ms = {sub.n_windows / dt2:.0f} windows/s")


if __name__ == "__main__":
    main()
