#!/usr/bin/env python3
"""Per-phase cycle breakdown of the POA kernel (diagnostic library libhypo_gpu_prof.so, built with
-DHYPO_PHASE_TIMERS: s_memtime deltas accumulated per wave).  usage: phase_profile.py [n_windows]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402

NAMES = ["load_seq", "dp_rows", "traceback", "add_alignment", "toposort", "consensus", "output", "rowmeta", "wave_lifetime", "waves"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 97078
    gpu = capi.HypoGpu(0, path=os.path.join(ROOT, "hypo_amd", "_build", "libhypo_gpu_prof.so"))
    b = sim.window_batch(n, seed=1000)
    db = gpu.device_batch(b)
    for _ in range(2):
        db.run()
    torch.cuda.synchronize()
    st = db.stats()
    ph = db.workspace[512:512 + 8 * 24 * 8].cpu().numpy().view(np.uint64).reshape(8, 24)
    print("windows per class", st["n_class"], "escalated", st["n_escalated"])
    for c in range(8):
        if ph[c, 9] == 0 or st['n_class'][c] == 0:
            continue
        tot = float(ph[c, :8].sum())
        life = float(ph[c, 8])
        print(f"class {c}: waves={int(ph[c, 9])} mean wave lifetime={life / ph[c, 9] / 1e3:.1f} kcycles "
              f"accounted={100 * tot / life:.1f}%  cycles/window={life / max(st['n_class'][c], 1) / 1e3:.1f}k")
        nw = max(st['n_class'][c], 1)
        print(f"    per window: DP rows={ph[c, 10] / nw:.1f} real alignments={ph[c, 11] / nw:.2f} reused={ph[c, 12] / nw:.2f} "
              f"toposorts={ph[c, 13] / nw:.2f} serial consensus={ph[c, 14] / nw:.3f}; dp cycles/row={ph[c, 1] / max(ph[c, 10], 1):.0f}; "
              f"slow rows={100 * ph[c, 15] / max(ph[c, 10], 1):.1f}%; exact threading {ph[c, 17] / nw:.2f} of {ph[c, 16] / nw:.2f} tries")
        for i in range(8):
            print(f"    {NAMES[i]:14s} {100 * ph[c, i] / tot:6.2f}%   {ph[c, i] / max(st['n_class'][c], 1) / 1e3:9.2f} kcycles/window")


if __name__ == "__main__":
    main()
