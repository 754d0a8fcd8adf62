#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
It builds oracle/_ref/libhyporef.so (the reference's own Window / PackedSeq / spoa sources compiled
in place by oracle/Makefile, SISD flavour) and records {inputs -> outputs of the reference}:

  windows_synth.jsonl.gz     G1  seeded synthetic windows covering the edge cases of SURVEY.md §8c
  windows_real_c1.jsonl.gz   G1b stratified sample of REAL-pipeline windows (inputs harvested from the
                                 reference's inspect dump of the C1-shaped 5 Mbp run, see inspect_dump.py);
                                 expected = real Window class, cross-checked against the dump's own consensus
  windows_real_long.jsonl.gz G1b same from the `-B` (long-read) run: LONG windows + neighbours
  replay_cases.jsonl.gz      G2  sequence-level {ordered sequences, modes, scores} -> last alignment pairs,
                                 final rank order, heaviest-bundle consensus (kNW / kLOV / kROV)
  spoa_sample.json.gz        G5  spoa's own pin: test/data/sample.fastq reads + the GlobalConsensus
                                 known-answer string of external/spoa/test/spoa_test.cpp:220-239
  packedseq_cases.json.gz        PackedSeq<2|4> text -> unpack() round trips

Fixtures are data only (inputs + expected outputs); no reference source is stored.
"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import inspect_dump  # noqa: E402
from hypo_amd.batch import TextWindow  # noqa: E402
import oracle  # noqa: E402

REF = "/root/reference"
DUMP_C1 = os.environ.get("HYPO_INSPECT_C1", "/tmp/oracle/e2e5/aux/inspect_ctg1.txt")
DUMP_LONG = os.environ.get("HYPO_INSPECT_LONG", "/tmp/oracle/e2e3/aux/inspect_ctg1.txt")
DEFAULT = [5, -4, -8, 3, -5, -4]

# known answer held by the reference's own test (external/spoa/test/spoa_test.cpp:228-237)
SPOA_GLOBAL_CONSENSUS = (
    "ATGATGCGCTTTGTTGGCGCGGTGGCTTGATGCAGGGGCTAATCGAC"
    "CTCTGGCAACCACTTTTCCATGACAGGAGTTGAATATGGCATTCAGTAATCCCTTCGATGATCCGCAGGG"
    "AGCGTTTTACATATTGCGCAATGCGCAGGGGCAATTCAGTCTGTGGCCGCAACAATGCGTCTTACCGGCA"
    "GGCTGGGACATTGTGTGTCAGCCGCAGTCACAGGCGTCCTGCCAGCAGTGGCTGGAAGCCCACTGGCGTA"
    "CTCTGACACCGACGAATTTTACCCAGTTGCAGGAGGCACAATGAGCCAGCATTTACCTTTGGTCGCCGCA"
    "CAGCCCGGCATCTGGATGGCAGAAAAACTGTCAGAATTACCCTCCGCCTGGAGCGTGGCGCATTACGTTG"
    "AGTTAACCGGAGAGGTTGATTCGCCATTACTGGCCCGCGCGGTGGTTGCCGGACTAGCGCAAGCAGATAC"
    "GC")


def mutate(rng, s, err):
    out = []
    for c in s:
        x = rng.random()
        if x < err / 3:
            continue
        out.append(rng.choice("ACGT") if x < 2 * err / 3 else c)
        if rng.random() < err / 3:
            out.append(rng.choice("ACGT"))
    return "".join(out) or "A"


def rec(w, scores, cons, kept=None, tag=""):
    d = {"tag": tag, "long": int(w.is_long), "scores": list(scores), "draft": w.draft,
         "internal": w.internal, "prefix": w.prefix, "suffix": w.suffix, "n_empty": w.n_empty,
         "consensus": cons}
    if kept is not None:
        d["kept"] = kept
    return d


def dump(name, records):
    path = os.path.join(HERE, name)
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        for r in records:
            f.write((json.dumps(r, separators=(",", ":")) + "\n").encode())
    print(f"{name}: {len(records)} records, {os.path.getsize(path)} bytes")


def synth_windows(ref):
    rng = random.Random(20260928)
    out = []

    def make(L, narm, err, mix, draft_err=0.03, with_n=False, no_internal=False, n_empty=0, is_long=False,
             scores=DEFAULT, tag=""):
        truth = "".join(rng.choice("ACGT") for _ in range(L))
        draft = mutate(rng, truth, draft_err)
        if with_n:
            d = list(draft)
            for _ in range(1 + len(d) // 40):
                d[rng.randrange(len(d))] = "N"
            draft = "".join(d)
        w = TextWindow(draft, is_long=is_long, n_empty=n_empty)
        for _ in range(narm):
            s = mutate(rng, truth, err)
            k = rng.random()
            if k < mix[0] and not no_internal:
                w.internal.append(s)
            elif k < mix[0] + mix[1]:
                w.prefix.append(s[:rng.randint(max(1, len(s) // 10), len(s))])
            else:
                w.suffix.append(s[rng.randint(0, len(s) - max(1, len(s) // 10)):])
        cons, kept = ref.window(w, scores)
        if is_long:  # the reference filters long arms (Window.hpp:66-101): keep inputs + kept flags
            out.append(rec(w, scores, cons, kept, tag))
        else:
            out.append(rec(w, scores, cons, None, tag))

    # grid of SURVEY.md §8(d): length x arms x error, 60/20/20 mix
    for L in (8, 16, 32, 64, 100, 200):
        for narm in (6, 12, 30, 50):
            for err in (0.005, 0.08):
                for _ in range(3):
                    make(L, narm, err, (0.6, 0.2, 0.2), tag=f"grid L{L} a{narm} e{err}")
    for i in range(40):
        make(rng.choice((2, 3, 5, 8, 13)), rng.choice((2, 3, 5, 9)), 0.1, (0.6, 0.2, 0.2), tag="tiny")
    for i in range(40):
        make(rng.choice((30, 60, 100)), rng.choice((3, 8, 20)), 0.02, (0.0, 0.5, 0.5), no_internal=True,
             tag="draft-backbone prefix+suffix")
    for i in range(20):
        make(rng.choice((30, 60)), rng.choice((3, 8)), 0.05, (0.0, 1.0, 0.0), no_internal=True, tag="prefix only")
    for i in range(20):
        make(rng.choice((30, 60)), rng.choice((3, 8)), 0.05, (0.0, 0.0, 1.0), no_internal=True, tag="suffix only")
    for i in range(40):
        make(rng.choice((20, 50, 100)), rng.choice((4, 10, 25)), 0.03, (0.6, 0.2, 0.2), with_n=True, tag="N in draft")
    for i in range(30):
        make(rng.choice((20, 50)), rng.choice((0, 1, 2, 3)), 0.03, (0.7, 0.15, 0.15), n_empty=rng.choice((0, 1, 2, 5)),
             tag="few arms / empties")
    for i in range(60):
        make(rng.choice((16, 40, 90)), rng.choice((6, 15, 30)), 0.25, (0.5, 0.25, 0.25), tag="very noisy (kLOV/kROV ties)")
    for sc in ([2, -3, -1, 3, -5, -4], [1, -1, -1, 1, -1, -1], [10, -10, 0, 3, -5, -4], [5, -4, -8, 5, -4, -8]):
        for i in range(20):
            make(rng.choice((16, 40, 90)), rng.choice((6, 15)), 0.08, (0.6, 0.2, 0.2), scores=sc, tag=f"scores {sc[:3]}")
    # zero-length arms are skipped by the window loops (Window.cpp:100,113,124)
    for i in range(10):
        truth = "".join(rng.choice("ACGT") for _ in range(30))
        w = TextWindow(mutate(rng, truth, 0.03), [mutate(rng, truth, 0.02) for _ in range(4)] + [""],
                       ["", truth[:12]], [truth[20:], ""], 0)
        cons, _ = ref.window(w, DEFAULT)
        out.append(rec(w, DEFAULT, cons, None, "zero-length arms"))
    # long windows, 2-round curate
    for L, narm in ((200, 10), (300, 25), (500, 40), (500, 12), (350, 3)):
        for _ in range(2):
            make(L, narm, 0.10, (0.7, 0.15, 0.15), is_long=True, tag=f"long L{L} a{narm}")
    return out


def real_windows(ref, path, n_plain, n_presuf, n_long, seed):
    rng = random.Random(seed)
    recs = list(inspect_dump.parse(path))
    plain = [r for r in recs if not r[2].is_long and not (r[2].prefix or r[2].suffix)]
    presuf = [r for r in recs if not r[2].is_long and (r[2].prefix or r[2].suffix)]
    longs = [r for r in recs if r[2].is_long]
    pick = rng.sample(plain, min(n_plain, len(plain))) + rng.sample(presuf, min(n_presuf, len(presuf))) \
        + sorted(longs, key=lambda r: len(r[2].draft) * (1 + len(r[2].internal)))[:n_long]
    out = []
    for hdr, typ, w, cons_dump in pick:
        cons, kept = ref.window(w, DEFAULT)
        # a dumped window is post-filter already: the real class must keep every arm and agree with the pipeline
        assert all(kept), hdr
        assert cons == cons_dump, ("real Window class disagrees with the pipeline dump", hdr)
        out.append(rec(w, DEFAULT, cons, None, f"{typ} {hdr.strip('=')}"))
    return out


def replay_cases(ref):
    rng = random.Random(424242)
    out = []
    for i in range(400):
        L = rng.choice((6, 12, 25, 50, 100))
        truth = "".join(rng.choice("ACGT") for _ in range(L))
        n = rng.choice((2, 3, 5, 9, 15))
        err = rng.choice((0.02, 0.1, 0.3))
        seqs, modes = [], []
        for k in range(n):
            s = mutate(rng, truth, err)
            mode = oracle.NW if k == 0 else rng.choice((oracle.NW, oracle.NW, oracle.LOV, oracle.ROV))
            if mode == oracle.LOV:
                s = "J" + s[:rng.randint(1, len(s))]
            elif mode == oracle.ROV:
                s = s[rng.randint(0, len(s) - 1):] + "O"
            else:
                s = "J" + s + "O"
            seqs.append(s)
            modes.append(mode)
        sc = rng.choice(([5, -4, -8], [5, -4, -8], [3, -5, -4], [1, -1, -1], [2, -6, -2]))
        rc, pairs, rank, cons = ref.replay(seqs, modes, sc)
        assert rc == 0
        out.append({"scores": sc, "seqs": seqs, "modes": modes, "pairs": pairs.reshape(-1).tolist(),
                    "rank": rank.tolist(), "consensus": cons})
    return out


def spoa_sample(ref):
    reads = []
    with open(os.path.join(REF, "external/spoa/test/data/sample.fastq")) as f:
        lines = f.read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        reads.append(lines[i + 1])
    rc, pairs, rank, cons = ref.replay(reads, [oracle.NW] * len(reads), [5, -4, -8])
    assert cons == SPOA_GLOBAL_CONSENSUS, "reference build does not reproduce spoa's own GlobalConsensus pin"
    path = os.path.join(HERE, "spoa_sample.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps({"scores": [5, -4, -8], "reads": reads, "consensus": cons,
                            "rank_len": int(rank.size)}).encode())
    print("spoa_sample.json.gz:", len(reads), "reads,", os.path.getsize(path), "bytes")


def packedseq_cases(ref):
    rng = random.Random(7)
    cases = []
    for L in list(range(0, 12)) + [31, 32, 33, 100, 257]:
        t2 = "".join(rng.choice("ACGT") for _ in range(L))
        t4 = "".join(rng.choice("ACGTNRYacgtn") for _ in range(L))
        cases.append({"nb": 2, "text": t2, "unpacked": ref.pack_roundtrip(2, t2)})
        cases.append({"nb": 4, "text": t4, "unpacked": ref.pack_roundtrip(4, t4)})
    path = os.path.join(HERE, "packedseq_cases.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(cases).encode())
    print("packedseq_cases.json.gz:", len(cases))


def main():
    oracle.build(ref=True)
    ref = oracle.Ref()
    dump("windows_synth.jsonl.gz", synth_windows(ref))
    dump("replay_cases.jsonl.gz", replay_cases(ref))
    spoa_sample(ref)
    packedseq_cases(ref)
    if os.path.exists(DUMP_C1):
        dump("windows_real_c1.jsonl.gz", real_windows(ref, DUMP_C1, 900, 500, 0, 11))
    else:
        print("skip windows_real_c1: no inspect dump at", DUMP_C1)
    if os.path.exists(DUMP_LONG):
        dump("windows_real_long.jsonl.gz", real_windows(ref, DUMP_LONG, 150, 60, 12, 12))
    else:
        print("skip windows_real_long: no inspect dump at", DUMP_LONG)


if __name__ == "__main__":
    main()
