#!/usr/bin/env python3
"""Bounded-exhaustive parity of the POA path against the REAL reference (oracle/_ref/libhyporef.so = hypo::Window + spoa compiled in
place): EVERY short window of a bounded shape, not a random sample of them.

What is enumerated.  A window = a draft and n arms over a small alphabet, every arm internal, prefix or suffix (a window keeps its arms
as internal | prefix | suffix, so the kinds are a multiset).  For an alphabet of s letters, drafts of 1..Ld bases and arms of 1..La
bases the space holds  (sum_{l<=Ld} s^l) * (sum_{l<=La} s^l)^n * C(n+2, 2)  windows:

    space   alphabet  arms  draft   arm     windows
    a2n2    {A,C}     2     1..6    1..6     12.0 M
    a2n3    {A,C}     3     1..5    1..5    147.8 M
    a2n4    {A,C}     4     1..4    1..3     17.3 M
    a3n2    {A,C,G}   2     1..4    1..4     10.4 M
    a3n3    {A,C,G}   3     1..3    1..3     23.1 M
    a4n2    {A,C,G,T} 2     1..3    1..3      3.6 M
    a4n3    {A,C,G,T} 3     1..2    1..2      1.6 M
    a2n5    {A,C}     5     1..3    1..2      2.3 M

Low-complexity two- and three-letter sequences are where the score-free shortcuts of hypo_amd/csrc/poa_core.hpp (Poa::thread_guided,
thread_cols, guided_one_sub, topo_insert, the lazy rank order) meet ties, runs of one letter, second ends and side entrances — the
shapes uniform random genomes produce once in 10^7 windows (DESIGN.md 3.1's incident) are all in here.  Every window runs through the
device in EVERY short size class (hypo_gpu_set_option("poa_min_class", c): class 0 in both geometries, 1, 2, 3 — each class has its own
code paths: class 0's four-group geometry threads along the guide only, class 3 keeps its order lazily) and the consensus must be the
bytes hypo::Window::generate_consensus returned (external/spoa/src/sisd_alignment_engine.cpp:279-288,338-339,370-428;
graph.cpp:154-271,293-353), under the default scores and the two alternative margin sets of test_one_substitution_shapes_vs_oracle.

usage: exhaustive_parity.py [--spaces a2n2,a3n2,...] [--scores all|default] [--classes 0,0w,1,2,3] [--stride K] [--lib prof]
       [--chunk N] [--check-oracle] [--budget-seconds S]
exit code 1 on the first difference (the window is printed).  --lib prof uses the diagnostic library (libhypo_gpu_prof.so) and prints how
often each shortcut answered.  Test infrastructure: nothing in the product imports this."""
import argparse
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hypo_amd import abi  # noqa: E402
from hypo_amd.batch import HostBatch  # noqa: E402

SPACES = {            # name: (alphabet size, arms, max draft, max arm)
    "a2n2": (2, 2, 6, 6),
    "a2n3": (2, 3, 5, 5),
    "a2n4": (2, 4, 4, 3),
    "a3n2": (3, 2, 4, 4),
    "a3n3": (3, 3, 3, 3),
    "a4n2": (4, 2, 3, 3),      # four letters: cliques of four aligned nodes (3.6 M windows)
    "a4n3": (4, 3, 2, 2),      # (1.6 M)
    "a2n5": (2, 5, 3, 2),      # five arms (2.3 M)
}
SCORE_SETS = [(5, -4, -8, 3, -5, -4), (2, -1, -2, 3, -5, -4), (4, -3, -5, 3, -5, -4)]
LETTERS = "ACGT"


def kind_multisets(n):
    """(n_internal, n_prefix, n_suffix) with sum n."""
    return [(i, p, n - i - p) for i in range(n, -1, -1) for p in range(n - i, -1, -1)]


def space_size(name):
    s, n, ld, la = SPACES[name]
    return sum(s ** l for l in range(1, ld + 1)) * sum(s ** l for l in range(1, la + 1)) ** n * len(kind_multisets(n))


def configs(name):
    """Every (draft length, arm lengths, kinds) of a space with the number of windows it holds."""
    s, n, ld, la = SPACES[name]
    for dl in range(1, ld + 1):
        for lens in itertools.product(range(1, la + 1), repeat=n):
            for kinds in kind_multisets(n):
                yield s, dl, lens, kinds, s ** (dl + sum(lens))


def _digits(idx, s, count):
    """Base-s digits of idx, least significant first: [len(idx), count] uint8."""
    out = np.empty((idx.size, count), dtype=np.uint8)
    x = idx.copy()
    for j in range(count):
        out[:, j] = x % s
        x //= s
    return out


def build_config(s, dl, lens, kinds, first, count):
    """HostBatch of windows [first, first + count) of one configuration (all windows share every length: regular offsets)."""
    idx = np.arange(first, first + count, dtype=np.int64)
    dig = _digits(idx, s, dl + sum(lens))
    n = count
    na = len(lens)
    # draft: PackedSeq<4>, two bases per byte, first base in the high nibble
    db = (dl + 1) // 2
    dc = np.zeros((n, db * 2), dtype=np.uint8)
    dc[:, :dl] = dig[:, :dl]
    draft4 = ((dc[:, 0::2] << 4) | dc[:, 1::2]).astype(np.uint8).reshape(-1)
    # arms: PackedSeq<2>, four bases per byte, first base in the two highest bits; a window's arms lie back to back
    ab = [(l + 3) // 4 for l in lens]
    B = sum(ab)
    arms2 = np.zeros((n, B), dtype=np.uint8)
    col, o = dl, 0
    for j, l in enumerate(lens):
        ac = np.zeros((n, ab[j] * 4), dtype=np.uint8)
        ac[:, :l] = dig[:, col:col + l]
        arms2[:, o:o + ab[j]] = (ac[:, 0::4] << 6) | (ac[:, 1::4] << 4) | (ac[:, 2::4] << 2) | ac[:, 3::4]
        col += l
        o += ab[j]
    wd = np.zeros(n, dtype=abi.WINDOW_DTYPE)
    wd["type"] = abi.WIN_SHORT
    wd["draft_len"] = dl
    wd["draft_off"] = np.arange(n, dtype=np.uint64) * db
    wd["first_arm"] = np.arange(n, dtype=np.uint32) * na
    wd["n_internal"], wd["n_prefix"], wd["n_suffix"] = kinds
    pre = np.concatenate([[0], np.cumsum(ab)[:-1]]).astype(np.uint64)
    arm_off = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(B) + pre[None, :]).reshape(-1)
    arm_len = np.tile(np.asarray(lens, dtype=np.uint32), n)
    return HostBatch(wd, draft4, arm_off, arm_len, arms2.reshape(-1))


def concat(batches):
    """Several HostBatches as one (offsets shifted)."""
    if len(batches) == 1:
        return batches[0]
    wd = np.zeros(sum(b.n_windows for b in batches), dtype=abi.WINDOW_DTYPE)      # (np.concatenate would drop the padding of the 40-byte descriptor)
    pos = 0
    for b in batches:
        wd[pos:pos + b.n_windows] = b.windows
        pos += b.n_windows
    doff = np.cumsum([0] + [b.draft4.size for b in batches[:-1]])
    aoff = np.cumsum([0] + [b.arms2.size for b in batches[:-1]])
    farm = np.cumsum([0] + [b.n_arms for b in batches[:-1]])
    pos = 0
    for b, d, f in zip(batches, doff, farm):
        wd["draft_off"][pos:pos + b.n_windows] += np.uint64(d)
        wd["first_arm"][pos:pos + b.n_windows] += np.uint32(f)
        pos += b.n_windows
    return HostBatch(wd, np.concatenate([b.draft4 for b in batches]),
                     np.concatenate([b.arm_off + np.uint64(a) for b, a in zip(batches, aoff)]),
                     np.concatenate([b.arm_len for b in batches]), np.concatenate([b.arms2 for b in batches]))


def chunks(name, chunk, stride=1, offset=0):
    """HostBatches of about `chunk` windows covering the space (every `stride`-th block of a configuration when stride > 1)."""
    pend, have = [], 0
    blk = 0
    for s, dl, lens, kinds, size in configs(name):
        first = 0
        while first < size:
            take = min(size - first, max(chunk - have, 1))
            if stride == 1 or (blk % stride) == offset % stride:
                pend.append(build_config(s, dl, lens, kinds, first, take))
                have += take
            blk += 1
            first += take
            if have >= chunk:
                yield concat(pend)
                pend, have = [], 0
    if pend:
        yield concat(pend)


def describe(b, w):
    from hypo_amd.batch import unpack2, unpack4
    W = b.windows[w]
    d = unpack4(b.draft4[int(W["draft_off"]):int(W["draft_off"]) + (int(W["draft_len"]) + 1) // 2], int(W["draft_len"]))
    arms = []
    for a in range(int(W["first_arm"]), int(W["first_arm"]) + int(W["n_internal"]) + int(W["n_prefix"]) + int(W["n_suffix"])):
        o, l = int(b.arm_off[a]), int(b.arm_len[a])
        arms.append(unpack2(b.arms2[o:o + (l + 3) // 4], l))
    return f"draft {d} arms {arms} internal/prefix/suffix {int(W['n_internal'])}/{int(W['n_prefix'])}/{int(W['n_suffix'])}"


def first_difference(a, b, off):
    """a, b = (bases, len, status) laid out on the same slots `off`; index of the first window that differs, or -1."""
    (ab, al, ast), (bb, bl, bst) = a, b
    skip = (ast == 0xF0) | (bst == 0xF0)      # oracle.REF_ST_FILTERED: a LONG window whose arms the reference's own Window filter would not all keep
    if skip.any():
        al, bl, ast, bst = al.copy(), bl.copy(), ast.copy(), bst.copy()
        al[skip] = bl[skip] = 0
        ast[skip] = bst[skip] = 0
    bad = np.nonzero((al != bl) | (ast != bst))[0]
    if bad.size:
        return int(bad[0])
    ok = ast == 0
    o64, l64 = off[:-1].astype(np.int64)[ok], al.astype(np.int64)[ok]
    tot = int(l64.sum())
    if tot == 0:
        return -1
    idx = np.repeat(o64, l64) + (np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(l64) - l64, l64))
    ne = np.nonzero(ab[idx] != bb[idx])[0]
    if ne.size == 0:
        return -1
    starts = np.cumsum(l64) - l64
    return int(np.nonzero(ok)[0][np.searchsorted(starts, ne[0], side="right") - 1])


CLASS_VARIANTS = {"0": (0, "16"), "0w": (0, "32"), "1": (1, None), "2": (2, None), "3": (3, None)}
DBG = ["rows", "real_alignments", "reused", "toposorts", "serial_consensus", "slow_rows", "threading_attempts", "threading_hits(all)", "thread_guided_hits",
       "score_rows", "topo_dfs_steps", "topo_run_steps", "guided_one_sub_hits", "thread_cols_hits", "topo_insert_hits", "lazy_updates", "tied_end_rows_sorted"]


def run(spaces, score_sets, variants, chunk=2_000_000, stride=1, lib=None, check_oracle=False, budget=None, log=print, threads=0):
    import ctypes as C
    import torch
    import oracle
    from hypo_amd import capi
    gpu = capi.HypoGpu(0, path=lib) if lib else capi.HypoGpu(0)
    prof = bool(lib and "prof" in lib)
    ref = oracle.Ref()
    orc = oracle.Oracle() if check_oracle else None
    t0 = time.time()
    total, t_ref, t_gpu = 0, 0.0, 0.0
    hits = {v: np.zeros(len(DBG), dtype=np.int64) for v in variants}
    stats = {v: {"n_alignments": 0, "n_reused": 0, "n_threaded": 0, "n_escalated": 0, "cells_scored": 0} for v in variants}
    done_spaces = {}
    try:
        for name in spaces:
            n_space = 0
            for b in chunks(name, chunk, stride):
                off = b.slot_layout()
                for sc in score_sets:
                    tr = time.time()
                    rb, _, rln, rst, _ = ref.poa_batch_raw(b, scores=sc, off=off, n_threads=threads)
                    t_ref += time.time() - tr
                    want = (rb, rln, rst)
                    if orc is not None:
                        ob, _, oln, ost = orc.poa_batch_raw(b, scores=sc, off=off)[:4]
                        w = first_difference((ob, oln, ost), want, off)
                        if w >= 0:
                            log(f"ORACLE != REFERENCE space {name} scores {sc}: {describe(b, w)}")
                            return None
                    db = gpu.device_batch(b, off=off)
                    for v in variants:
                        mc, lanes = CLASS_VARIANTS[v]
                        if lanes:
                            os.environ["HYPO_POA_CLASS0"] = lanes
                        else:
                            os.environ.pop("HYPO_POA_CLASS0", None)
                        assert gpu.lib.hypo_gpu_set_option(b"poa_min_class", C.c_int(mc)) == 0
                        tg = time.time()
                        db.run(scores=sc)
                        gb, _, gln, gst = db.results()
                        t_gpu += time.time() - tg
                        w = first_difference((gb, gln, gst), want, off)
                        if w >= 0:
                            o = int(off[w])
                            log(f"MISMATCH space {name} class variant {v} scores {sc}: {describe(b, w)}\n  device status {int(gst[w])} "
                                f"'{gb[o:o + int(gln[w])].tobytes().decode()}' reference status {int(rst[w])} '{rb[o:o + int(rln[w])].tobytes().decode()}'")
                            return None
                        st = db.stats()
                        for k in stats[v]:
                            stats[v][k] += st[k]
                        if prof:
                            ph = db.workspace[512:512 + 6 * 32 * 8].cpu().numpy().view(np.uint64).reshape(6, 32)
                            hits[v] += ph[:, 11:11 + len(DBG)].sum(axis=0).astype(np.int64)
                    del db
                n_space += b.n_windows
                total += b.n_windows
                if budget and time.time() - t0 > budget:
                    break
            done_spaces[name] = (n_space, space_size(name))
            log(f"space {name}: {n_space} of {space_size(name)} windows x {len(score_sets)} score sets x {len(variants)} class variants identical to the reference "
                f"({time.time() - t0:.0f} s so far; reference {t_ref:.0f} s, device {t_gpu:.0f} s)")
            if budget and time.time() - t0 > budget:
                break
    finally:
        os.environ.pop("HYPO_POA_CLASS0", None)
        gpu.lib.hypo_gpu_set_option(b"poa_min_class", C.c_int(0))
    for v in variants:
        s = stats[v]
        na = max(s["n_alignments"], 1)
        log(f"class variant {v}: alignments {s['n_alignments']} reused {s['n_reused'] / na:.4f} threaded {s['n_threaded'] / na:.4f} "
            f"scored {1 - (s['n_reused'] + s['n_threaded']) / na:.4f}; re-queued windows {s['n_escalated']}")
        if prof:
            log("    " + ", ".join(f"{n} {int(x)}" for n, x in zip(DBG, hits[v])))
    return {"windows": total, "comparisons": total * len(score_sets) * len(variants), "spaces": done_spaces, "seconds": time.time() - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spaces", default="a2n2,a3n2,a2n4,a3n3,a2n3")
    ap.add_argument("--scores", default="all", choices=["all", "default"])
    ap.add_argument("--classes", default="0,0w,1,2,3")
    ap.add_argument("--stride", type=int, default=1, help="take every K-th block of `chunk` windows only (1 = exhaustive)")
    ap.add_argument("--chunk", type=int, default=2_000_000)
    ap.add_argument("--lib", default=None, help="'prof' = hypo_amd/_build/libhypo_gpu_prof.so (prints how often every shortcut answered)")
    ap.add_argument("--check-oracle", action="store_true", help="also compare oracle/hypo_oracle.c with the reference on every window")
    ap.add_argument("--budget-seconds", type=float, default=None)
    a = ap.parse_args()
    lib = os.path.join(ROOT, "hypo_amd", "_build", "libhypo_gpu_prof.so") if a.lib == "prof" else a.lib
    spaces = [s for s in a.spaces.split(",") if s]
    for s in spaces:
        print(f"# space {s}: alphabet {SPACES[s][0]}, {SPACES[s][1]} arms, draft 1..{SPACES[s][2]}, arms 1..{SPACES[s][3]}: {space_size(s)} windows", flush=True)
    r = run(spaces, SCORE_SETS if a.scores == "all" else SCORE_SETS[:1], a.classes.split(","), a.chunk, a.stride, lib, a.check_oracle, a.budget_seconds,
            log=lambda m: print(m, flush=True))
    if r is None:
        sys.exit(1)
    print(f"OK: {r['windows']} windows ({r['comparisons']} device results) identical to hypo::Window::generate_consensus, 0 mismatches, {r['seconds']:.0f} s")


if __name__ == "__main__":
    main()
