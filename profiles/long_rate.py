#!/usr/bin/env python3
"""Rate of LONG windows (size class 4, state in HBM scratch): the 163 real LONG windows of tests/golden replicated.
usage: long_rate.py [replicas]"""
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
    lib = os.environ.get("HYPO_GPU_LIB")
    gpu = capi.HypoGpu(0, path=lib) if lib else capi.HypoGpu(0)
    db = gpu.device_batch(b)
    db.run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); db.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = db.stats()
    print(f"GPU: {len(wins)} LONG windows in {dt * 1e3:.1f} ms = {len(wins) / dt:.0f} windows/s; cells {st['dp_cells'] / 1e9:.2f} G -> {st['dp_cells'] / dt / 1e9:.1f} GCUPS; classes {st['n_class'][:5]} failed {st['n_failed']}")
    if lib and "prof" in lib:
        import numpy as np
        names = ["load_seq", "dp_rows", "traceback", "add_alignment", "toposort", "consensus", "output", "rowmeta"]
        ph = db.workspace[512:512 + 8 * 16 * 8].cpu().numpy().view(np.uint64).reshape(8, 16)
        c = 4
        tot = float(ph[c, :8].sum())
        print("class 4 phases: " + ", ".join(f"{n} {100 * ph[c, i] / tot:.1f}%" for i, n in enumerate(names)),
              f"; rows/window {ph[c, 10] / max(st['n_class'][c], 1) / 2:.0f} (per run), cycles/row {ph[c, 1] / max(ph[c, 10], 1):.0f}")
    import oracle
    orc = oracle.Oracle()
    sub = build_batch(wins[:len(recs) * min(rep, 2)])
    t0 = time.perf_counter(); orc.poa_batch_raw(sub); dt2 = time.perf_counter() - t0
    print(f"CPU oracle ({orc.num_threads()} threads): {sub.n_windows} windows in {dt2 * 1e3:.0f} ms = {sub.n_windows / dt2:.0f} windows/s")


if __name__ == "__main__":
    main()
