#!/usr/bin/env python3
"""Goldens of the opt-in native flavour (--native-klov): windows whose prefix arms are noisy, polished by the REAL reference
classes compiled with -march=native (oracle/_ref/libhyporef_native.so: spoa then uses its AVX2 / SSE4.1 engine, whose kLOV
end-row rule differs from the scalar engine's, external/spoa/src/simd_alignment_engine.cpp:803,834-840,859-861).
Every record also carries the scalar (default build) consensus, so the fixture shows where the two flavours part.

Run in the build container only (needs /root/reference and a CPU with AVX2 or SSE4.1):  python tests/golden/make_native_golden.py
Writes tests/golden/windows_native_klov.jsonl.gz."""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from hypo_amd.batch import TextWindow  # noqa: E402
import oracle  # noqa: E402


def mutate(rng, s, e):
    out = []
    for c in s:
        x = rng.random()
        if x < e / 3:
            continue
        out.append(rng.choice("ACGT") if x < 2 * e / 3 else c)
        if rng.random() < e / 3:
            out.append(rng.choice("ACGT"))
    return "".join(out) or "A"


def main():
    oracle.build(ref=True)
    native = oracle.Ref(os.path.join(os.path.dirname(oracle.REF_SO), "libhyporef_native.so"))
    scalar = oracle.Ref()
    rng = random.Random(20260930)
    recs, n_diff = [], 0
    tries = 0
    while len(recs) < 400 and tries < 40000:
        tries += 1
        L = rng.choice([20, 30, 45, 60, 80, 100, 150])
        truth = "".join(rng.choice("ACGT") for _ in range(L))
        w = TextWindow(mutate(rng, truth, 0.05))
        for _ in range(rng.choice([5, 8, 12, 20])):
            s = mutate(rng, truth, rng.choice([0.03, 0.08, 0.15]))
            k = rng.random()
            if k < 0.4:
                w.internal.append(s)
            elif k < 0.85:
                w.prefix.append(s[:rng.randint(1, len(s))])
            else:
                w.suffix.append(s[rng.randint(0, len(s) - 1):])
        cn, cs = native.window(w)[0], scalar.window(w)[0]
        differ = cn != cs
        if differ or len(recs) - n_diff < 150:            # every diverging window found, plus 150 on which the flavours agree
            recs.append({"draft": w.draft, "internal": w.internal, "prefix": w.prefix, "suffix": w.suffix, "n_empty": 0, "long": False,
                         "scores": [5, -4, -8, 3, -5, -4], "consensus_native": cn, "consensus_scalar": cs})
            n_diff += differ
    out = os.path.join(HERE, "windows_native_klov.jsonl.gz")
    with gzip.open(out, "wt") as f:
        for r in recs:
            f.write(json.dumps(r) + "\n")
    print(f"{len(recs)} windows, {n_diff} on which the native and the scalar flavour differ ({tries} tried); {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
