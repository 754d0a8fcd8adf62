#!/usr/bin/env python3
"""Per-phase cycle breakdown of the POA kernel (diagnostic library libhypo_gpu_prof.so, built with
-DHYPO_PHASE_TIMERS: s_memtime deltas accumulated per wave).  usage: phase_profile.py [n_windows] [read_sub]      (HYPO_POA_SEQUENTIAL=1: one size class at a time)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hypo_amd import capi, sim  # noqa: E402

NAMES = ["load_seq", "score_rows", "traceback", "add_alignment", "toposort", "consensus", "output", "rowmeta", "exact_rows", "wave_lifetime", "waves"]
NP = 9      # phases; then lifetime, waves, 17 counters (poa_kernel.hip)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 97078
    gpu = capi.HypoGpu(0, path=os.path.join(ROOT, "hypo_amd", "_build", "libhypo_gpu_prof.so"))
    sub = float(sys.argv[2]) if len(sys.argv) > 2 else 0.002
    b = sim.window_batch(n, seed=1000, read_sub=sub)
    print(f"# {n} windows, read_sub={sub}, {'sequential' if os.environ.get('HYPO_POA_SEQUENTIAL') else 'concurrent'} schedule")
    db = gpu.device_batch(b)
    for _ in range(2):
        db.run()
    torch.cuda.synchronize()
    st = db.stats()
    ph = db.workspace[512:512 + 6 * 32 * 8].cpu().numpy().view(np.uint64).reshape(6, 32)      # poa_kernel.hip: [class][32]
    print("windows per class", st["n_class"], "escalated", st["n_escalated"])
    for c in range(6):
        if ph[c, NP + 1] == 0 or st['n_class'][c] == 0:
            continue
        tot = float(ph[c, :NP].sum())
        life = float(ph[c, NP])
        D = NP + 2
        print(f"class {c}: waves={int(ph[c, NP + 1])} mean wave lifetime={life / ph[c, NP + 1] / 1e3:.1f} kcycles "
              f"accounted={100 * tot / life:.1f}%  cycles/window={life / max(st['n_class'][c], 1) / 1e3:.1f}k")
        nw = max(st['n_class'][c], 1)
        print(f"    per window: rows={ph[c, D] / nw:.1f} real alignments={ph[c, D + 1] / nw:.2f} reused={ph[c, D + 2] / nw:.2f} "
              f"toposorts={ph[c, D + 3] / nw:.2f} serial consensus={ph[c, D + 4] / nw:.3f}; exact threading {ph[c, D + 7] / nw:.2f} of {ph[c, D + 6] / nw:.2f} tries")
        print(f"    score rows: {ph[c, D + 9] / nw:.1f} per window at {ph[c, 1] / max(ph[c, D + 9], 1):.0f} cycles/row ({100 * ph[c, D + 5] / max(ph[c, D + 9], 1):.1f}% slow-path rows)   "
              f"threading: {ph[c, D + 8] / nw:.2f} per window along the guide; {ph[c, NP - 1] / max(ph[c, D + 6], 1):.0f} cycles per attempt")
        for i in range(NP):
            print(f"    {NAMES[i]:14s} {100 * ph[c, i] / tot:6.2f}%   {ph[c, i] / max(st['n_class'][c], 1) / 1e3:9.2f} kcycles/window")


if __name__ == "__main__":
    main()
