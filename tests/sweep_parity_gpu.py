#!/usr/bin/env python3
"""Opt-in long parity sweep on a GPU box (not collected by pytest): many seeded batches through libhypo_gpu and through the
oracle, compared byte for byte.  Shapes: the C1-shaped simulator batch at several read error rates (re-queue paths, both
class-0 geometries), tiny-window batches (dense short reads, HiFi-like depth), wide windows, and the randomised windows of
test_gpu_fuzz.py.
usage: sweep_parity_gpu.py [minutes] [first_round]      exit code 1 on the first mismatch (rounds seed the data)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hypo_amd import capi, sim  # noqa: E402
from hypo_amd.batch import build_batch  # noqa: E402
import oracle  # noqa: E402
from test_gpu_fuzz import _window  # noqa: E402
from test_poa_emulator import _one_sub_windows  # noqa: E402


REF = oracle.Ref() if oracle.Ref.available() else None      # the real reference classes compiled in place (oracle/_ref/libhyporef.so)
REF_SLICE = 25000                                            # windows of every comparison that also go through it
REF_CHECKED = [0]


def compare(gpu, orc, b, scores, tag):
    bases, off, ln, st = gpu.poa_batch(b, scores)
    ob, ooff, oln, ost = orc.poa_batch_raw(b, scores=scores)[:4]
    if REF is not None:
        # a slice of every batch against hypo::Window::generate_consensus itself, not only against the restatement
        from hypo_amd import dist as hd
        import exhaustive_parity as ex
        n = min(b.n_windows, REF_SLICE)
        sb = hd.take_windows(b, 0, n, compact=True)
        soff = sb.slot_layout()
        rb, _, rln, rst, _ = REF.poa_batch_raw(sb, scores=scores, off=soff)
        gb2, _, gln2, gst2 = gpu.poa_batch(sb, scores, off=soff)
        w = ex.first_difference((gb2, gln2, gst2), (rb, rln, rst), soff)
        if w >= 0:
            print(f"MISMATCH vs the REAL reference {tag}: window {w}: {ex.describe(sb, w)}", flush=True)
            sys.exit(1)
        REF_CHECKED[0] += n
    bad = np.nonzero((ln != oln) | (st != ost))[0]
    if len(bad) == 0:
        for i in np.nonzero(st == 0)[0]:
            a, o = int(off[i]), int(ooff[i])
            if bases[a:a + int(ln[i])].tobytes() != ob[o:o + int(oln[i])].tobytes():
                bad = np.array([i])
                break
    if len(bad):
        print(f"MISMATCH {tag}: window {int(bad[0])} ({len(bad)} differ)", flush=True)
        sys.exit(1)
    return b.n_windows


def one_round(gpu, orc, rnd):
    """One round of the sweep (seeded by `rnd`): ~0.54 M windows through libhypo_gpu and the oracle; exits on the first mismatch."""
    total = 0
    default = (5, -4, -8, 3, -5, -4)
    for sub in (0.002, 0.01, 0.03):
        for lanes in ("16", "32"):
            os.environ["HYPO_POA_CLASS0"] = lanes
            total += compare(gpu, orc, sim.window_batch(60000, seed=5000 + rnd, read_sub=sub), default, f"sim sub={sub} lanes={lanes} round={rnd}")
    os.environ.pop("HYPO_POA_CLASS0", None)
    # tiny windows almost only (the rule picks four class-0 groups per wave), many arms per window, wide windows
    rs = np.random.default_rng(9000 + rnd)
    n = 80000
    wl = rs.choice([3, 5, 8, 12, 16, 24, 32, 48, 64, 99], size=n, p=[.15, .15, .15, .13, .13, .1, .1, .05, .03, .01])
    for lo, hi, tag in ((3, 45, "dense"), (44, 58, "hifi")):
        shapes = np.stack([wl, rs.integers(lo, hi, size=n), np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)], axis=1)
        total += compare(gpu, orc, sim.window_batch(n, seed=9100 + rnd, shapes=shapes, read_sub=0.004), default, f"{tag} round={rnd}")
    for length in (130, 160, 200):
        total += compare(gpu, orc, sim.grid_batch(length, 20, 1500, 0.01, seed=9200 + rnd), default, f"wide {length} round={rnd}")
    rng = np.random.default_rng(7000 + rnd)
    for scores in (default, (3, -6, -5, 3, -5, -4), (1, -1, -1, 1, -1, -1)):
        wins = [_window(rng, False) for _ in range(6000)] + [_window(rng, True) for _ in range(150)]
        total += compare(gpu, orc, build_batch(wins), scores, f"fuzz scores={scores} round={rnd}")
    # arms one base off the arm before them over low-complexity drafts (Poa::guided_one_sub, Poa::topo_insert)
    for scores in (default, (2, -1, -2, 3, -5, -4), (4, -3, -5, 3, -5, -4)):
        total += compare(gpu, orc, build_batch(_one_sub_windows(rng, 8000)), scores, f"one-sub scores={scores} round={rnd}")
    # ... and the same with prefix and suffix arms in every window (kLOV / kROV: free ends inside runs of one letter)
    total += compare(gpu, orc, build_batch(_one_sub_windows(rng, 30000, mixed_frac=1.0)), default, f"one-sub prefix/suffix arms round={rnd}")
    return total


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    gpu = capi.HypoGpu(0)
    orc = oracle.Oracle()
    t_end = time.time() + 60 * minutes
    total, rnd = 0, (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    while time.time() < t_end:
        rnd += 1
        total += one_round(gpu, orc, rnd)
        print(f"round {rnd}: {total} windows identical to the oracle so far, {REF_CHECKED[0]} of them also to oracle/_ref/libhyporef.so", flush=True)
    print(f"OK: {total} windows, 0 mismatches ({REF_CHECKED[0]} windows also compared with the real reference classes)")


if __name__ == "__main__":
    main()
