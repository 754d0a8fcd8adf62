#!/usr/bin/env python3
"""Opt-in CPU fuzz of the kernel's source in the lockstep emulator (not collected by pytest): seeded simulator batches at several read-error
rates and the randomised windows of test_gpu_fuzz.py start in size class 0, 1, 2 or 3, follow the kernel's re-queue chain
with their graphs (Emu.poa_chain) and are compared with the oracle string by string.
usage: emu_fuzz_cpu.py <seed> <minutes> [long]  exit code 1 on the first difference   (long: LONG windows through the LONG class and class 5)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hypo_amd import sim  # noqa: E402
from hypo_amd.batch import build_batch  # noqa: E402
import oracle  # noqa: E402
from emu_util import Emu  # noqa: E402
from test_gpu_fuzz import _window  # noqa: E402

SCORES = [(5, -4, -8, 3, -5, -4), (3, -5, -4, 3, -5, -4), (2, -3, -2, 5, -4, -8)]


def main():
    seed, minutes = int(sys.argv[1]), float(sys.argv[2])
    LONG = len(sys.argv) > 3 and sys.argv[3] == "long"
    emu, orc = Emu(), oracle.Oracle()
    rng = np.random.default_rng(seed)
    t_end = time.time() + 60 * minutes
    compared = rnd = 0
    while time.time() < t_end:
        rnd += 1
        sub = float(rng.choice([0.002, 0.005, 0.01, 0.02, 0.05]))
        if LONG:
            b = build_batch([_window(rng, True) for _ in range(12)])
            sc = tuple(int(x) for x in SCORES[rnd % 3])
            want = orc.poa_batch(b, scores=sc)[0]
            for cfg in (4, 5) if rnd % 4 == 0 else (4,):
                cons, st, res = emu.poa_batch(b, cfg, scores=sc)[:3]
                for i in range(b.n_windows):
                    if cons[i] is None:
                        continue
                    if cons[i] != want[i]:
                        print(f"DIFFERENCE seed {seed} round {rnd} class {cfg} LONG window {i} scores {sc}", flush=True)
                        sys.exit(1)
                    compared += 1
            if rnd % 5 == 0:
                print(f"seed {seed} round {rnd} compared {compared} bad 0", flush=True)
            continue
        if rnd % 3:
            b = sim.window_batch(300, seed=seed * 100003 + rnd, read_sub=sub)
        else:
            b = build_batch([_window(rng, False) for _ in range(150)])
        sc = SCORES[rnd % 3]
        sc = tuple(int(x) for x in sc)
        want = orc.poa_batch(b, scores=sc)[0]
        for cfg in (0, 1, 2, 3)[(rnd % 2):][:3]:
            cons, res, hops, carried = emu.poa_chain(b, cfg, scores=sc)
            for i in range(b.n_windows):
                if cons[i] is None:
                    continue                                   # (a status other than OK: compared by the status tests)
                if cons[i] != want[i]:
                    print(f"DIFFERENCE seed {seed} round {rnd} class {cfg} window {i} scores {sc}", flush=True)
                    sys.exit(1)
                compared += 1
        if rnd % 5 == 0:
            print(f"seed {seed} round {rnd} compared {compared} bad 0", flush=True)
    print(f"seed {seed} round {rnd} compared {compared} bad 0", flush=True)


if __name__ == "__main__":
    main()
