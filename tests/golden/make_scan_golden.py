#!/usr/bin/env python3
"""G3: scan fixtures {contig text, k, solid-kmer bit set} -> marked positions + k-mer ids + ranks, produced
by the REAL reference: hypo::Contig::find_solid_pos (src/Contig.cpp:40-74) over the real suk::SolidKmers / sdsl bit vector,
compiled in place by oracle/Makefile into oracle/_ref/libhyporef_scan.so (oracle/ref_scan_harness.cpp).

Run in the build container only (needs /root/reference):   python tests/golden/make_scan_golden.py
Writes tests/golden/scan_cases.json.gz.  Data only: inputs + the reference's outputs.

Cases: random contigs with N and homopolymer runs (both homopolymer-edge tests, Contig.cpp:59,63), lengths around the 64-bit
word boundaries, contigs shorter than k, k from 2 (the CLI's minimum, main.cpp:172) to 12, bit sets from the simulator
(unique canonical k-mers, both strands), an all-ones set and a random set, lower-case / IUPAC letters (-> N, PackedSeq.cpp:44-48).
"""
import base64
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from hypo_amd import sim  # noqa: E402
import oracle  # noqa: E402


def b64(a: np.ndarray) -> str:
    return base64.b64encode(gzip.compress(np.ascontiguousarray(a).tobytes(), 9)).decode()


def contig_text(rng, n, n_frac, hp_frac, odd_letters):
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    for _ in range(int(n * hp_frac)):
        if n < 4:
            break
        s = int(rng.integers(0, n - 2))
        codes[s:s + int(rng.integers(2, 9))] = rng.integers(0, 4)
    codes[rng.random(n) < n_frac] = 4
    text = np.frombuffer(b"ACGTN", dtype=np.uint8)[codes].copy()
    if odd_letters and n:
        pos = rng.integers(0, n, size=max(1, n // 300))
        text[pos] = rng.choice(np.frombuffer(b"acgtnRYKMSWxX-", dtype=np.uint8), size=pos.size)
    return text.tobytes()


def enc(text: bytes) -> np.ndarray:
    lut = np.full(256, 4, np.uint8)
    for c, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[c] = v
    return lut[np.frombuffer(text, dtype=np.uint8)]


def main():
    oracle.build(ref=True)
    ref = oracle.RefScan()
    rng = np.random.default_rng(20260929)
    cases = []
    plan = [  # (n, k, n_frac, hp_frac, bitset kind, odd letters)
        (3000, 5, 0.01, 0.02, "sim3", False), (2999, 7, 0.005, 0.02, "sim3", False), (20000, 9, 0.002, 0.01, "sim2", True),
        (12000, 11, 0.001, 0.005, "sim1", False), (12001, 11, 0.0, 0.0, "sim1", True), (10000, 12, 0.001, 0.01, "sim1", False),
        (64, 5, 0.0, 0.05, "ones", False), (65, 5, 0.02, 0.05, "ones", False), (63, 7, 0.0, 0.0, "ones", False),
        (128, 3, 0.02, 0.1, "ones", False), (129, 2, 0.0, 0.1, "ones", False), (4, 5, 0.0, 0.0, "ones", False),
        (11, 11, 0.0, 0.0, "ones", False), (12, 11, 0.0, 0.0, "ones", False), (1, 2, 0.0, 0.0, "ones", False),
        (5000, 8, 0.0, 0.3, "ones", False), (8191, 8, 0.01, 0.0, "rand", True), (8192, 9, 0.0, 0.02, "rand", False),
        (8193, 6, 0.3, 0.0, "ones", False), (16384 + 70, 11, 0.0005, 0.01, "sim1", False),
    ]
    for n, k, n_frac, hp_frac, kind, odd in plan:
        text = contig_text(rng, n, n_frac, hp_frac, odd)
        codes = enc(text)
        nw = max((1 << (2 * k)) // 64, 1)
        if kind.startswith("sim"):
            bits = sim.solid_bitset(codes, k, max_count=int(kind[3:]))
        elif kind == "ones":
            bits = np.full(nw, np.uint64(0xFFFFFFFFFFFFFFFF))
            if 4 ** k < 64:
                bits[0] = np.uint64((1 << (4 ** k)) - 1)
        else:
            bits = rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64) & \
                rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)
        words, kids, rank, ns = ref.solid_scan(text, k, bits)
        # the bit set travels as the sorted list of set k-mer ids when that compresses better than the words
        ids = np.flatnonzero(np.unpackbits(bits.view(np.uint8), bitorder="little")).astype(np.uint32)
        bs = {"bitset_ids": b64(ids)} if len(b64(ids)) < len(b64(bits)) else {"bitset": b64(bits)}
        cases.append({"n": n, "k": k, "contig": text.decode(), **bs, "words": b64(words), "kids": b64(kids),
                      "rank": b64(rank), "n_solid": ns})
        print(f"n={n} k={k} {kind}: {ns} solid positions")
    out = os.path.join(HERE, "scan_cases.json.gz")
    with gzip.open(out, "wt") as f:
        json.dump({"source": "hypo::Contig::find_solid_pos (src/Contig.cpp:40-74) via oracle/ref_scan_harness.cpp", "cases": cases}, f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
