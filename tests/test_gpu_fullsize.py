"""GPU, BASELINE.json's full sizes.  The oracle still finishes in seconds on the host cores for the C2 batch and
for 100-250 Mbp scans, so those are compared directly; the C3-sized POA call (~1.9 M windows in ONE batch) is
checked through size-independent properties: replicas of the same window must agree wherever they land in the
queues, waves and size classes, and the base windows must equal the oracle."""
import numpy as np
import pytest

from hypo_amd import abi, capi, sim
from hypo_amd.batch import HostBatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    return capi.HypoGpu(0)


def _cons_list(bases, off, ln):
    return [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes() for i in range(ln.size)]


def test_c2_full_batch_vs_oracle(gpu, oracle_lib):
    b = sim.window_batch(97078, seed=2024)
    off = b.slot_layout()
    db = gpu.device_batch(b, off=off)
    db.run()
    bases, _, ln, st = db.results()
    ob, _, oln, ost, cells, aligns = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == 0).all() and (ost == 0).all() and (ln == oln).all()
    idx = np.repeat(off[:-1].astype(np.int64), ln.astype(np.int64)) + \
        (np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln.astype(np.int64)) - ln, ln.astype(np.int64)))
    assert (bases[idx] == ob[idx]).all()
    s = db.stats()
    assert s["dp_cells"] == cells and s["n_alignments"] == aligns and s["n_failed"] == 0
    # determinism: a second run of the same resident batch gives identical bytes
    db.run()
    bases2, _, ln2, st2 = db.results()
    assert (ln2 == ln).all() and (bases2[idx] == bases[idx]).all()


def test_c3_sized_batch_replica_property(gpu, oracle_lib):
    """~1.94 M windows (the C3 configuration's count) in one call: 20 shuffled replicas of a C1-shaped batch that
    share the packed buffers.  Every replica must produce the consensus of its original, and the originals must
    equal the oracle."""
    base = sim.window_batch(97078, seed=7)
    reps = 20
    rng = np.random.default_rng(1)
    order = rng.permutation(base.n_windows * reps)
    src = order % base.n_windows                         # window i of the big batch is a copy of base window src[i]
    big = HostBatch(base.windows[src].copy(), base.draft4, base.arm_off, base.arm_len, base.arms2)
    off = big.slot_layout()
    bases, _, ln, st = gpu.poa_batch(big, off=off)
    assert (st == 0).all()
    s = gpu.last_stats()
    assert s["n_failed"] == 0 and sum(s["n_class"]) == big.n_windows
    ob, ooff, oln, ost, _, _ = oracle_lib.poa_batch_raw(base)
    want = _cons_list(ob, ooff, oln)
    assert (ln == oln[src]).all()
    got = _cons_list(bases, off, ln)
    bad = [i for i in range(big.n_windows) if got[i] != want[src[i]]]
    assert not bad, f"{len(bad)} of {big.n_windows} windows differ, first {bad[:3]}"


def test_c2_full_batch_and_long_windows_vs_real_reference(gpu, ref_lib):
    """The prebuilt real reference classes (oracle/_ref, built where /root/reference exists) travel to the GPU box: the
    full C2 batch and a batch of LONG windows are compared with hypo::Window::generate_consensus itself."""
    if not hasattr(ref_lib.lib, "hyporef_batch"):
        pytest.skip("oracle/_ref/libhyporef.so predates hyporef_batch")
    import golden_util as gu
    from hypo_amd.batch import build_batch
    longs = [gu.to_window(r) for r in gu.load_jsonl("windows_real_long.jsonl.gz") if r["long"]]
    for b in (sim.window_batch(97078, seed=77), build_batch(longs * 4)):
        off = b.slot_layout()
        db = gpu.device_batch(b, off=off)
        db.run()
        bases, _, ln, st = db.results()
        rb, _, rln, rst, _ = ref_lib.poa_batch_raw(b, off=off)
        assert (st == rst).all() and (ln == rln).all()
        l64 = ln.astype(np.int64)
        idx = np.repeat(off[:-1].astype(np.int64), l64) + (np.arange(int(l64.sum()), dtype=np.int64) - np.repeat(np.cumsum(l64) - l64, l64))
        assert (bases[idx] == rb[idx]).all()


def test_dense_sr_shape_vs_oracle(gpu, oracle_lib):
    """The dense-SR shape of C4/C5: tiny windows (45 % <= 8 bp), ~22 arms of ~12 bases."""
    rng = np.random.default_rng(3)
    n = 60000
    wl = rng.choice([3, 5, 8, 12, 16, 24, 32, 48, 64, 99], size=n, p=[.15, .15, .15, .13, .13, .1, .1, .05, .03, .01])
    shapes = np.stack([wl, rng.integers(3, 45, size=n), np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)], axis=1)
    b = sim.window_batch(n, seed=9, shapes=shapes, read_sub=0.01)
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, off=off)
    ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == ost).all() and (ln == oln).all()
    assert _cons_list(bases, off, ln) == _cons_list(ob, off, oln)


@pytest.mark.parametrize("n_bases,k", [(100_000_000, 13), (250_000_000, 15)])
def test_scan_c3_c4_sizes_vs_oracle(gpu, oracle_lib, n_bases, k):
    """100 Mbp / k=13 (8 MiB set, L2-resident) and 250 Mbp / k=15 (128 MiB set, Infinity-Cache-resident)."""
    rng = np.random.default_rng(k)
    codes = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    codes[rng.integers(0, n_bases, size=n_bases // 5000)] = 4
    pad = codes.reshape(-1, 2)
    p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
    bits = rng.integers(0, 1 << 63, size=(1 << (2 * k)) // 64, dtype=np.int64).view(np.uint64)
    bits &= rng.integers(0, 1 << 63, size=bits.size, dtype=np.int64).view(np.uint64)       # ~25 % of k-mers solid
    cap = n_bases // 4
    ds = gpu.device_scan(p4, n_bases, k, bits, kids_cap=cap)
    ds.run()
    w, kids, rank, ns = ds.results()
    ow, okids, orank, ons = oracle_lib.solid_scan(p4, n_bases, k, bits, kids_cap=cap)
    assert ns == ons and (w == ow).all() and (rank == orank).all() and (kids == okids).all()
    assert int(rank[-1]) == ns == int(np.unpackbits(w.view(np.uint8)).sum())


# ---- C5 (BASELINE.json configs[4]): 3 Gbp draft => k = 17 (2 GiB bit set, src/main.cpp:490-528), 50x HiFi reads passed as
# ---- "short" (44-57 internal arms per window), wide windows ----------------------------------------------------------------
def test_scan_c5_k17_vs_oracle(gpu, oracle_lib):
    """k = 17: the 4^17-bit set is 2 GiB (far beyond L2 and the Infinity Cache); 512 Mbp of contig, every probe a random
    HBM sector."""
    k, n_bases = 17, 512_000_000
    rng = np.random.default_rng(k)
    codes = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    codes[rng.integers(0, n_bases, size=n_bases // 5000)] = 4
    pad = codes.reshape(-1, 2)
    p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
    del codes, pad
    nw = (1 << (2 * k)) // 64
    bits = rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)
    bits &= rng.integers(0, 1 << 63, size=nw, dtype=np.int64).view(np.uint64)        # ~25 % of all 17-mers solid
    cap = n_bases // 4
    ds = gpu.device_scan(p4, n_bases, k, bits, kids_cap=cap)
    ds.run()
    w, kids, rank, ns = ds.results()
    del ds
    ow, okids, orank, ons = oracle_lib.solid_scan(p4, n_bases, k, bits, kids_cap=cap)
    assert ns == ons and (w == ow).all() and (rank == orank).all() and (kids == okids).all()
    assert int(rank[-1]) == ns


def _tiny_lengths(rng, n):
    return rng.choice([3, 5, 8, 12, 16, 24, 32, 48, 64, 99], size=n, p=[.15, .15, .15, .13, .13, .1, .1, .05, .03, .01])


def test_c5_hifi_shape_million_windows_vs_oracle(gpu, oracle_lib):
    """1.05 M windows of the dense shape with 44-57 internal arms each (50x HiFi reads given as -b): 53 M alignments in one
    call, compared window by window with the oracle."""
    n = 1_050_000
    rng = np.random.default_rng(55)
    shapes = np.stack([_tiny_lengths(rng, n), rng.integers(44, 58, size=n), np.zeros(n, np.int64), np.zeros(n, np.int64),
                       np.zeros(n, np.int64)], axis=1)
    b = sim.window_batch(n, seed=56, shapes=shapes, read_sub=0.001)
    off = b.slot_layout()
    db = gpu.device_batch(b, off=off)
    db.run()
    bases, _, ln, st = db.results()
    s = db.stats()
    ob, _, oln, ost, cells, aligns = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == ost).all() and (ln == oln).all() and s["n_failed"] == 0
    assert s["dp_cells"] == cells and s["n_alignments"] == aligns
    l64 = ln.astype(np.int64)
    idx = np.repeat(off[:-1].astype(np.int64), l64) + (np.arange(int(l64.sum()), dtype=np.int64) - np.repeat(np.cumsum(l64) - l64, l64))
    assert (bases[idx] == ob[idx]).all()


def test_c5_wide_windows_full_batch_vs_oracle(gpu, oracle_lib):
    """The LDS-pressure case: 60 000 windows of 160-200 bp (the reference cuts a weak region only above 2 x 100 bp,
    src/Contig.cpp:526-711) with 30-56 arms, all in the 40 KB size class."""
    n = 60_000
    rng = np.random.default_rng(77)
    ni = rng.integers(24, 44, size=n)
    shapes = np.stack([rng.integers(160, 201, size=n), ni, rng.integers(3, 7, size=n), rng.integers(3, 7, size=n),
                       np.zeros(n, np.int64)], axis=1)
    b = sim.window_batch(n, seed=78, shapes=shapes, read_sub=0.003)
    off = b.slot_layout()
    db = gpu.device_batch(b, off=off)
    db.run()
    bases, _, ln, st = db.results()
    s = db.stats()
    ob, _, oln, ost, cells, aligns = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == ost).all() and (ln == oln).all() and s["n_failed"] == 0
    assert s["dp_cells"] == cells and s["n_alignments"] == aligns
    assert _cons_list(bases, off, ln) == _cons_list(ob, off, oln)


def _fresh_contexts(devices, env):
    """hypo_gpu_init for a device list, in a child process (the test process keeps its own single context)."""
    import subprocess
    import sys
    import os
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from hypo_amd import capi, sim
import oracle
gpu = capi.HypoGpu(0, devices=%r)
assert gpu.lib.hypo_gpu_num_devices() == %d
for seed, n in ((41, 30000), (42, 7)):
    b = sim.window_batch(n, seed=seed)
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch_sharded(b, off=off)
    s = gpu.last_stats()
    ob, _, oln, ost, cells, aligns = oracle.Oracle().poa_batch_raw(b, off=off)
    assert (st == ost).all() and (ln == oln).all() and s["dp_cells"] == cells and s["n_alignments"] == aligns and sum(s["n_class"]) + s["n_trivial"] >= n - 0
    l64 = ln.astype(np.int64)
    idx = np.repeat(off[:-1].astype(np.int64), l64) + (np.arange(int(l64.sum()), dtype=np.int64) - np.repeat(np.cumsum(l64) - l64, l64))
    assert (bases[idx] == ob[idx]).all()
print("SHARDED-OK")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), devices, len(devices))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SHARDED-OK" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
    return p


def test_sharded_batch_over_three_contexts_direct_gather(gpu):
    """hypo_gpu_poa_batch_sharded: cost-balanced contiguous ranges, rebased descriptors, per-device upload and run, results
    written into the caller's slots.  On a one-GPU box three contexts share device 0 (HYPO_ALLOW_DUP_DEVICES), which RCCL
    refuses, so the ranges come back per device."""
    _fresh_contexts([0, 0, 0], {"HYPO_ALLOW_DUP_DEVICES": "1", "HYPO_MULTI_GATHER": "direct"})


def test_sharded_batch_rccl_gather_single_rank(gpu):
    """The RCCL leg of the same call (library loaded on demand, ncclCommInitAll, grouped broadcasts of bases / lengths /
    status, copy back from device 0) with a communicator of one rank: everything but a second GPU."""
    _fresh_contexts([0], {"HYPO_MULTI_GATHER": "rccl"})


def test_parity_sweep_one_round(gpu, oracle_lib):
    """One seeded round of tests/sweep_parity_gpu.py inside -m gpu (~0.54 M windows: simulator batches at 0.2 / 1 / 3 % read error
    in both class-0 geometries, dense and HiFi-depth tiny windows, wide windows, fuzzed windows under three score sets), every
    consensus compared byte for byte with the oracle.  The open-ended sweep stays opt-in (110 M windows in round 1)."""
    import sweep_parity_gpu as sweep
    try:
        n = sweep.one_round(gpu, oracle_lib, 424242)
    except SystemExit as e:                         # compare() exits on the first mismatch after printing it
        pytest.fail(f"parity sweep mismatch (exit {e.code})")
    assert n > 500000


def test_c4_mixed_short_and_long_batch_vs_oracle(gpu, oracle_lib):
    """BASELINE config C4's window mix at size: 150 000 C1-shaped SHORT windows and 4 000 LONG windows (120-500 bp, 12-45 arms with
    10 % errors incl. indels, two rounds + curate) shuffled into ONE batch; every consensus equals the oracle's."""
    b = sim.c4_batch(150000, 4000, seed=404)
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, off=off)
    ob, _, oln, ost = oracle_lib.poa_batch_raw(b, off=off)[:4]
    assert (st == ost).all() and (ln == oln).all() and (st == 0).all()
    want, got = _cons_list(ob, off, oln), _cons_list(bases, off, ln)
    bad = [i for i in range(len(want)) if want[i] != got[i]]
    assert not bad, f"{len(bad)} windows differ, first {bad[0]} (type {int(b.windows['type'][bad[0]])})"
    is_long = b.windows["type"] == abi.WIN_LONG
    assert int(is_long.sum()) == 4000 and (ln[is_long] > 60).all()


def _deep_windows():
    """Windows beyond what round 2's largest class held (1 023 sequences, 4 000 nodes, 16 in-edges): a SHORT window of a deep
    repeat (1 300 arms of ~90 bp at 3 % error), one with 2 200 noisy arms (~6 000 nodes), and LONG windows whose graphs pass
    what the LONG class holds (500 bp, 300-420 arms at 8 % errors incl. indels: more than the 255 sequences of the LONG class; noisier arms would be dropped by
    Filter::is_good inside the real Window class, which the reference harness goes through)."""
    from hypo_amd.batch import TextWindow, build_batch
    from test_gpu_fuzz import _mutate
    rng = np.random.default_rng(2026)
    A = "ACGT"
    wins = []
    for n_arms, L, err in ((1300, 90, 0.03), (2200, 100, 0.10)):
        truth = "".join(A[i] for i in rng.integers(0, 4, size=L))
        wins.append(TextWindow(_mutate(rng, truth, 0.01), [_mutate(rng, truth, err) for _ in range(n_arms)], [], [], n_empty=0, is_long=False))
    for n_arms in (300, 420):
        truth = "".join(A[i] for i in rng.integers(0, 4, size=500))
        wins.append(TextWindow(_mutate(rng, truth, 0.02), [_mutate(rng, truth, 0.08) for _ in range(n_arms)], [], [], n_empty=0, is_long=True))
    return build_batch(wins)


def test_last_resort_class_vs_oracle_and_reference(gpu, oracle_lib):
    """No window 'keeps its draft' because it is big: the last class holds 16 382 sequences / 32 767 nodes / 58 in-edges per
    window (the reference has no limit, external/spoa/src/graph.cpp:99-128).  Deep SHORT windows and LONG windows with graphs of
    more than 4 000 nodes against the oracle and, where the prebuilt real reference classes are there, against those."""
    import oracle
    b = _deep_windows()
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, off=off)
    s = gpu.last_stats()
    assert (st == 0).all() and s["n_failed"] == 0
    assert s["n_class"][5] == b.n_windows, s["n_class"]        # they really ran in the last class
    ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(b, off=off)
    assert (ost == 0).all() and (ln == oln).all()
    assert _cons_list(bases, off, ln) == _cons_list(ob, off, oln)
    if oracle.Ref.available():
        ref = oracle.Ref()
        if hasattr(ref.lib, "hyporef_batch"):
            rb, _, rln, rst, _ = ref.poa_batch_raw(b, off=off)
            assert (rst == 0).all() and (rln == ln).all()
            assert _cons_list(bases, off, ln) == _cons_list(rb, off, rln)


@pytest.mark.parametrize("name", ["e2e_real_5m_s131", "e2e_real_long_s137"])
def test_real_windows_of_non_iid_sets_vs_the_reference_own_consensus(gpu, name, tmp_path):
    """Round 6: the REAL windows of the non-i.i.d. goldens — cut by the reference compiled in place from reads with indel errors, a second
    haplotype and mis-placed reads over a genome with tandem repeats, homopolymer runs and dispersed copies (tests/e2e_util.py:
    realistic_window_batch) — through the device in every short size class; the consensus of every window must be the string the reference's
    own Window::generate_consensus left in its per-region dump (no oracle in between).  ~96 k SHORT windows (5 Mbp set), SHORT + LONG windows
    with real long-read arms (the `-B` set)."""
    import ctypes as C
    import os
    import e2e_util as eu
    import oracle
    if not oracle.RefArms.available():
        pytest.skip("oracle/_ref/libhyporef_arms.so not built (the real reference only exists in the build container)")
    b, cons, man, rr = eu.realistic_window_batch(tmp_path, name)
    # (the reference counts a LONG window without a single arm as valid; its dump record has nothing to polish and is not part of the batch)
    assert man["reference_counts"]["windows"] - 50 <= b.n_windows <= man["reference_counts"]["windows"] and b.n_windows > 50000
    want = [c.encode() for c in cons]
    off = b.slot_layout()
    db = gpu.device_batch(b, off=off)
    try:
        for variant, (mc, lanes) in {"0": (0, "16"), "0w": (0, "32"), "1": (1, None), "2": (2, None), "3": (3, None)}.items():
            if lanes:
                os.environ["HYPO_POA_CLASS0"] = lanes
            else:
                os.environ.pop("HYPO_POA_CLASS0", None)
            assert gpu.lib.hypo_gpu_set_option(b"poa_min_class", C.c_int(mc)) == 0
            db.run()
            bases, _, ln, st = db.results()
            assert (st == 0).all(), (variant, np.nonzero(st)[0][:5])
            got = _cons_list(bases, off, ln)
            bad = [i for i in range(b.n_windows) if got[i] != want[i]]
            assert not bad, f"{name} class variant {variant}: {len(bad)} windows differ from the reference's consensus, first {bad[0]}"
            s = db.stats()
            na = max(s["n_alignments"], 1)
            print(f"{name} variant {variant}: windows per class {s['n_class'][:6]} alignments reused {s['n_reused'] / na:.3f} threaded {s['n_threaded'] / na:.3f} "
                  f"scored {1 - (s['n_reused'] + s['n_threaded']) / na:.3f} re-queued {s['n_escalated']}")
    finally:
        os.environ.pop("HYPO_POA_CLASS0", None)
        gpu.lib.hypo_gpu_set_option(b"poa_min_class", C.c_int(0))
