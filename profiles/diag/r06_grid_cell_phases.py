#!/usr/bin/env python3
"""Per-phase cycles of one grid cell in the diagnostic library (libhypo_gpu_prof.so).  usage: r06_grid_cell_phases.py length arms err [min_class]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402

NAMES = ["load_seq", "score_rows", "traceback", "add_alignment", "toposort", "consensus", "output", "rowmeta", "exact_rows"]
DBG = ["rows", "real_alignments", "reused", "toposorts", "serial_consensus", "slow_rows", "threading_attempts", "threading_hits", "guided_hits", "score_rows", "topo_dfs", "topo_run",
       "one_sub_hits", "cols_hits", "topo_inserts", "lazy_updates", "tie_sorts", "cycles_in_account", "of_those_spill_pool_allocation", "of_those_spill"]


def main():
    length, arms, err = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    mc = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    nwin = max(4000, min(60000, 24_000_000 // (length * arms)))
    gpu = capi.HypoGpu(0, path=os.path.join(ROOT, "hypo_amd", "_build", "libhypo_gpu_prof.so"))
    assert gpu.lib.hypo_gpu_set_option(b"poa_min_class", C.c_int(mc)) == 0
    db = gpu.device_batch(sim.grid_batch(length, arms, nwin, err, seed=11))
    for _ in range(3):
        db.run()
        torch.cuda.synchronize()
    st = db.stats()
    ph = db.workspace[512:512 + 6 * 32 * 8].cpu().numpy().view(np.uint64).reshape(6, 32)
    print(f"# cell {length} x {arms} @ {err}, {nwin} windows, min class {mc}: windows per class {st['n_class'][:6]} re-queued {st['n_escalated']}")
    for c in range(6):
        if ph[c, 10] == 0:
            continue
        tot, life, waves = float(ph[c, :9].sum()), float(ph[c, 9]), int(ph[c, 10])
        print(f"class {c}: waves {waves}, mean wave lifetime {life / waves / 1e3:.0f} kcycles, accounted {100 * tot / max(life, 1):.0f} %")
        print("   " + ", ".join(f"{n} {100 * ph[c, i] / max(tot, 1):.1f}%" for i, n in enumerate(NAMES)))
        print("   " + ", ".join(f"{n} {int(ph[c, 11 + i])}" for i, n in enumerate(DBG)))


if __name__ == "__main__":
    main()
