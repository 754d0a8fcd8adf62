"""GPU: parity of the HIP POA path (through the C-ABI of libhypo_gpu.so) with the committed goldens
of the real reference and with the oracle on seeded inputs.  Bit-exact: consensus strings must be equal."""
import ctypes as C

import numpy as np
import pytest

from hypo_amd import abi, capi, sim
from hypo_amd.batch import TextWindow, build_batch
import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    return capi.HypoGpu(0)


def _same_consensus(bases, ob, off, ln):
    """Compares only the len[w] valid bytes of every slot."""
    n = ln.size
    idx = np.repeat(off[:-1].astype(np.int64), ln.astype(np.int64)) + \
        (np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln.astype(np.int64)) - ln, ln.astype(np.int64)))
    return bool((bases[idx] == ob[idx]).all())


def _short_only(items):
    return [it for it in items if not it[0].is_long]


@pytest.mark.parametrize("name", gu.WINDOW_FILES)
def test_goldens_host_api(gpu, name):
    total = 0
    for scores, items in gu.windows_by_scores(name).items():
        if not items:
            continue
        cons, st = gpu.poa_consensus(build_batch([w for w, _, _ in items]), scores)
        for (w, want, tag), got, s in zip(items, cons, st):
            assert s == 0, (tag, s)
            assert got == want, (tag, scores)
            total += 1
    assert total > 100


def test_goldens_device_api(gpu):
    items = gu.windows_by_scores("windows_real_c1.jsonl.gz")[abi.DEFAULT_SCORES] + \
        gu.windows_by_scores("windows_real_long.jsonl.gz")[abi.DEFAULT_SCORES]
    db = gpu.device_batch(build_batch([w for w, _, _ in items]))
    db.run()
    cons, st = db.consensus()
    assert [c for c in cons] == [want for _, want, _ in items]
    stats = db.stats()
    assert stats["n_failed"] == 0 and sum(stats["n_class"]) == len(items)


def test_vs_oracle_c1_shape(gpu, oracle_lib):
    b = sim.window_batch(20000, seed=11)
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, off=off)
    ob, _, oln, ost, cells, aligns = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == 0).all() and (ost == 0).all()
    assert (ln == oln).all()
    assert _same_consensus(bases, ob, off, ln)
    s = gpu.last_stats()
    assert s["dp_cells"] == cells and s["n_alignments"] == aligns


@pytest.mark.parametrize("length,arms,err", [(8, 6, 0.08), (32, 30, 0.005), (64, 50, 0.08), (100, 30, 0.08),
                                             (100, 50, 0.005), (200, 30, 0.08)])
def test_vs_oracle_grid(gpu, oracle_lib, length, arms, err):
    b = sim.grid_batch(length, arms, 300, err, seed=length * 1000 + arms)
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, off=off)
    ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == ost).all() and (ln == oln).all()
    assert _same_consensus(bases, ob, off, ln)


@pytest.mark.parametrize("scores", [(2, -3, -1, 3, -5, -4), (1, -1, -1, 1, -1, -1), (10, -10, 0, 3, -5, -4),
                                    (127, -128, -128, 3, -5, -4)])
def test_vs_oracle_scores(gpu, oracle_lib, scores):
    b = sim.grid_batch(60, 12, 300, 0.1, seed=77)
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, scores=scores, off=off)
    ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(b, scores=scores, off=off)
    assert (st == ost).all() and (ln == oln).all()
    assert _same_consensus(bases, ob, off, ln)


def test_dispatch_and_edge_cases(gpu, oracle_lib):
    ws = [
        TextWindow("ACGTNACGT", ["ACGTACGT"], [], [], 0),
        TextWindow("ACGTACGT", ["ACGTACGT", "ACGTACGT"], [], [], 3),
        TextWindow("ACGTACGT", [], [], [], 0),
        TextWindow("ACGTACGT", ["", ""], [], [], 0),
        TextWindow("ACGAACGT", ["ACGTACGT", "ACGTACGT", "ACGTACGT"], [], [], 3),
        TextWindow("A", ["A", "C", "A"], [], [], 0),
        TextWindow("ACGTNNACGT", [], ["ACG", "ACGTA", "ACGTAC"], ["CGT", "ACGT", "TACGT"], 0),
    ]
    b = build_batch(ws)
    cons, st = gpu.poa_consensus(b)
    ocons, ost, _, _ = oracle_lib.poa_batch(b)
    assert list(st) == list(ost)
    assert cons == ocons
    assert cons[:5] == ["ACGTNACGT", "", "ACGTACGT", "ACGTACGT", "ACGTACGT"]


def test_slot_overflow_and_empty_batch(gpu):
    b = build_batch([TextWindow("ACGTACGTAC", ["ACGTACGTAC"] * 3)])
    bases, off, ln, st = gpu.poa_batch(b, off=np.array([0, 4], dtype=np.uint64))
    assert st[0] == abi.ST_CONS_OVERFLOW and ln[0] == 10
    e = build_batch([])
    bases, off, ln, st = gpu.poa_batch(e)
    assert ln.size == 0


def test_invalid_scores_rejected(gpu):
    b = build_batch([TextWindow("ACGT", ["ACGT", "ACGT"])])
    with pytest.raises(capi.HypoGpuError):
        gpu.poa_batch(b, scores=(5, -4, 1, 3, -5, -4))


def test_escalation_between_classes(gpu, oracle_lib):
    """Windows that overflow a size class are re-queued on the device and still come out bit-exact."""
    b = sim.grid_batch(100, 50, 200, 0.2, seed=5)       # very noisy: graphs outgrow the plan's estimate
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, off=off)
    ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == ost).all() and (ln == oln).all() and _same_consensus(bases, ob, off, ln)
    s = gpu.last_stats()
    assert s["n_failed"] == 0
    assert sum(s["n_class"]) == 200


def test_long_windows_vs_oracle(gpu, oracle_lib):
    """LONG windows (two-round curate) of random noisy arms, incl. arms longer than the window."""
    import random
    rng = random.Random(3)

    def mut(s, e):
        out = []
        for c in s:
            x = rng.random()
            if x < e / 3:
                continue
            out.append(rng.choice("ACGT") if x < 2 * e / 3 else c)
            if rng.random() < e / 3:
                out.append(rng.choice("ACGT"))
        return "".join(out) or "A"
    ws = []
    for _ in range(24):
        L = rng.choice([200, 350, 500])
        truth = "".join(rng.choice("ACGT") for _ in range(L))
        w = TextWindow(mut(truth, 0.03), is_long=True)
        for _ in range(rng.choice([3, 12, 30])):
            s = mut(truth, 0.1)
            k = rng.random()
            (w.internal if k < 0.7 else w.prefix if k < 0.85 else w.suffix).append(
                s if k < 0.7 else (s[:rng.randint(len(s) // 2, len(s))] if k < 0.85 else s[rng.randint(0, len(s) // 2):]))
        ws.append(w)
    b = build_batch(ws)
    cons, st = gpu.poa_consensus(b)
    ocons, ost, _, _ = oracle_lib.poa_batch(b)
    assert list(st) == list(ost) == [0] * len(ws)
    assert cons == ocons


def test_branch_completion_tie(gpu):
    """LONG window whose heaviest-bundle end node is not a sink: branch completion (graph.cpp:660-705) meets two in-edges of
    equal weight whose sources score the same, and the later one must win.  An earlier form of that loop was compiled wrongly
    for the device (right in the CPU emulator); found by the messy end-to-end seeds."""
    doc = gu.load_json("windows_branch_completion.json.gz")
    for c in doc["windows"]:
        w = TextWindow(c["draft"], c["internal"], c["prefix"], c["suffix"], n_empty=c["n_empty"], is_long=c["long"])
        cons, st = gpu.poa_consensus(build_batch([w]), tuple(c["scores"]))
        assert st[0] == 0 and cons[0] == c["consensus"], c["tag"]


def test_smaller_workspace_limits_only_residency(gpu, oracle_lib):
    """The HBM scratch follows the workspace the caller gives: the 16-group minimum still polishes 150 LONG windows correctly,
    anything smaller is refused with HYPO_E_WORKSPACE (include/hypo_gpu.h)."""
    longs = [gu.to_window(r) for r in gu.load_jsonl("windows_real_long.jsonl.gz") if r["long"]]
    b = build_batch((longs * 13)[:150])
    n = b.n_windows
    rec = int(gpu.lib.hypo_gpu_poa_workspace_bytes(C.c_uint32(n), C.c_uint32(0)))
    prefix = 8192 + (6 * n * 4 + 255) // 256 * 256 + (2 * n + 255) // 256 * 256
    # the error message of a hopelessly small workspace names the minimum
    with pytest.raises(capi.HypoGpuError) as ei:
        gpu.device_batch(b, workspace_bytes=prefix).run()
    minimal = int(str(ei.value).split("minimum ")[1].split(" ")[0])
    assert prefix < minimal < rec
    db = gpu.device_batch(b, workspace_bytes=minimal)
    db.run()
    bases, off, ln, st = db.results()
    ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(b, off=off)
    assert (st == ost).all() and (ln == oln).all() and _same_consensus(bases, ob, off, ln)
    with pytest.raises(capi.HypoGpuError):
        gpu.device_batch(b, workspace_bytes=minimal - 4096).run()


def test_concurrent_host_threads(gpu, oracle_lib):
    """Four host threads, each with its own batch, workspace and stream, call the device entry point at the same time
    (ctypes releases the GIL): the context's shared side streams and plan read-back are serialised inside the library."""
    import threading
    import torch
    batches = [sim.window_batch(3000, seed=300 + t, read_sub=0.002 + 0.004 * t) for t in range(4)]
    dbs = [gpu.device_batch(b) for b in batches]
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    errs = []

    def work(t):
        try:
            for _ in range(6):
                dbs[t].run(stream=streams[t])
            streams[t].synchronize()
        except Exception as e:        # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(4):
        bases, off, ln, st = dbs[t].results()
        ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(batches[t], off=off)
        assert (st == ost).all() and (ln == oln).all() and _same_consensus(bases, ob, off, ln), t


def test_bad_descriptors_are_answered_not_followed(gpu, oracle_lib):
    """Windows whose descriptor points outside the batch's buffers get HYPO_ST_INVALID; their neighbours are unaffected."""
    b = sim.window_batch(300, seed=31)
    good = oracle_lib.poa_batch(b)[0]
    w = b.windows.copy()
    w["first_arm"][5] = b.n_arms - 1                      # arms run past n_arms
    w["draft_off"][9] = b.draft4.size + 100               # draft outside draft4
    w["n_internal"][13] = 0xfffffff0                      # count overflow
    ao = b.arm_off.copy()
    a17 = int(w["first_arm"][17])
    ao[a17] = np.uint64(b.arms2.size + 5)                 # one arm's bytes outside arms2
    from hypo_amd.batch import HostBatch
    bad = HostBatch(w, b.draft4, ao, b.arm_len, b.arms2)
    off = b.slot_layout()
    db = gpu.device_batch(bad, off=off)
    db.run()
    bases, _, ln, st = db.results()
    for i in (5, 9, 13, 17):
        assert st[i] == abi.ST_INVALID and ln[i] == 0, (i, st[i])
    for i in range(300):
        if i not in (5, 9, 13, 17):
            assert st[i] == 0 and bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes().decode() == good[i]
    with pytest.raises(capi.HypoGpuError):                # the host-side helper refuses to index past the arm table
        gpu.poa_batch(bad)


def test_build_id_and_contexts(gpu):
    import test_abi
    test_abi.test_build_id_matches_sources(gpu.lib)
    assert gpu.lib.hypo_gpu_num_devices() == 1
    assert gpu.lib.hypo_gpu_use_device(0) == 0 and gpu.lib.hypo_gpu_use_device(1) == abi.HYPO_E_INVALID


def test_two_batches_in_flight_and_computed_arm_offsets(gpu, oracle_lib):
    """hypo_gpu_poa_batch_begin / _end: two batches queued back to back on one context (a third is refused), arm offsets
    computed on the device (arm_off = NULL: the simulator lays arms back to back), results equal to the oracle's."""
    bs = [sim.window_batch(3000, seed=61), sim.window_batch(2500, seed=62, read_sub=0.01)]
    offs = [b.slot_layout() for b in bs]
    outs = [(np.zeros(int(o[-1]) + 1, np.uint8), np.zeros(b.n_windows, np.uint32), np.zeros(b.n_windows, np.uint8)) for b, o in zip(bs, offs)]
    t0, k0 = gpu.poa_batch_begin(bs[0], offs[0], *outs[0], no_arm_off=True)
    t1, k1 = gpu.poa_batch_begin(bs[1], offs[1], *outs[1], no_arm_off=True)
    assert {t0, t1} == {0, 1}
    with pytest.raises(capi.HypoGpuError):
        gpu.poa_batch_begin(bs[0], offs[0], *outs[0])
    gpu.poa_batch_end(t1)
    gpu.poa_batch_end(t0)
    with pytest.raises(capi.HypoGpuError):
        gpu.poa_batch_end(t0)
    for b, off, (bases, ln, st) in zip(bs, offs, outs):
        ob, _, oln, ost, _, _ = oracle_lib.poa_batch_raw(b, off=off)
        assert (st == ost).all() and (ln == oln).all() and _same_consensus(bases, ob, off, ln)
