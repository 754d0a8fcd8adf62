"""Randomised windows far from the benchmark shapes, device against the oracle: arms much shorter or longer than the window,
0-15 % error, no internal arms, prefix/suffix only, N in the draft, LONG windows of every flavour, odd score sets.  These
shapes reach the rarely taken paths (branch completion, weight ties, literal toposort, class escalation, curate)."""
import numpy as np
import pytest

from hypo_amd import capi
from hypo_amd.batch import TextWindow, build_batch

pytestmark = pytest.mark.gpu
A = "ACGT"


@pytest.fixture(scope="module")
def gpu():
    import os
    return capi.HypoGpu(0, path=os.environ["HYPO_GPU_LIB"]) if os.environ.get("HYPO_GPU_LIB") else capi.HypoGpu(0)


def _mutate(rng, s, err):
    out = []
    for ch in s:
        r = rng.random()
        if r < err / 3:
            continue                                            # deletion
        if r < 2 * err / 3:
            out.append(A[rng.integers(4)])                      # substitution (may be silent)
        else:
            out.append(ch)
        if rng.random() < err / 3:
            out.append(A[rng.integers(4)])                      # insertion
    return "".join(out) or "A"


def _window(rng, is_long, huge=False):
    L = int(rng.choice([6, 12, 25, 40, 70, 100, 140, 190])) if not is_long else int(rng.choice([700, 900] if huge else [120, 200, 330, 480]))
    truth = "".join(A[i] for i in rng.integers(0, 4, size=L))
    draft = list(_mutate(rng, truth, 0.02))
    if rng.random() < 0.15 and not is_long:
        draft[rng.integers(len(draft))] = "N"
    draft = "".join(draft)
    err = float(rng.choice([0.0, 0.01, 0.05, 0.15]))
    narm = int(rng.integers(2, 14 if is_long else 40))
    kind = rng.choice(["internal", "mixed", "prefix", "suffix", "presuf"])
    internal, prefix, suffix = [], [], []
    for _ in range(narm):
        k = kind if kind != "mixed" else rng.choice(["internal", "prefix", "suffix"])
        if kind == "presuf":
            k = rng.choice(["prefix", "suffix"])
        if k == "internal":
            internal.append(_mutate(rng, truth, err))
        elif k == "prefix":
            cut = int(rng.integers(max(1, L // 10), L + 1))
            prefix.append(_mutate(rng, truth[:cut], err))
        else:
            cut = int(rng.integers(0, L - max(1, L // 10) + 1))
            suffix.append(_mutate(rng, truth[cut:], err))
    n_empty = int(rng.integers(0, 3)) if rng.random() < 0.1 else 0
    return TextWindow(draft, internal, prefix, suffix, n_empty=n_empty, is_long=is_long)


@pytest.mark.parametrize("seed,is_long,n,scores", [
    (1, False, 3000, (5, -4, -8, 3, -5, -4)),
    (2, False, 1500, (3, -6, -5, 3, -5, -4)),
    (3, False, 800, (1, -1, -1, 1, -1, -1)),
    (4, True, 400, (5, -4, -8, 3, -5, -4)),
    (5, True, 300, (5, -4, -8, 2, -3, -2)),
    (6, True, 200, (5, -4, -8, 1, -1, -1)),
])
def test_random_windows_vs_oracle(gpu, oracle_lib, seed, is_long, n, scores):
    rng = np.random.default_rng(seed)
    wins = [_window(rng, is_long) for _ in range(n)]
    b = build_batch(wins)
    cons, st = gpu.poa_consensus(b, scores)
    ocons, ost = oracle_lib.poa_batch(b, scores=scores)[:2]
    bad = [i for i in range(n) if cons[i] != ocons[i] or st[i] != ost[i]]
    assert not bad, (f"{len(bad)} of {n} windows differ; first: window {bad[0]} draft {len(wins[bad[0]].draft)} bp, "
                     f"{len(wins[bad[0]].internal)}/{len(wins[bad[0]].prefix)}/{len(wins[bad[0]].suffix)} arms, long={is_long}")
    s = gpu.last_stats()
    assert s["n_failed"] == 0


def test_windows_beyond_the_long_class(gpu, oracle_lib):
    """LONG windows of 700-900 bp (more than the LONG class's 639 columns) and short windows with 300 arms: the catch-all class."""
    rng = np.random.default_rng(77)
    wins = [_window(rng, True, huge=True) for _ in range(48)]
    truth = "".join(A[i] for i in rng.integers(0, 4, size=60))
    wins += [TextWindow(truth, [_mutate(rng, truth, 0.05) for _ in range(300)], [], []) for _ in range(4)]
    b = build_batch(wins)
    cons, st = gpu.poa_consensus(b, (5, -4, -8, 3, -5, -4))
    ocons, ost = oracle_lib.poa_batch(b)[:2]
    assert [i for i in range(len(wins)) if cons[i] != ocons[i] or st[i] != ost[i]] == []
    assert gpu.last_stats()["n_class"][5] >= 48


@pytest.mark.parametrize("lanes", ["16", "32"])
def test_both_class0_geometries(gpu, oracle_lib, lanes, monkeypatch):
    """Class 0 runs as four 16-lane or two 32-lane groups per wave, picked per call from the batch's mix (poa_run);
    HYPO_POA_CLASS0 forces one.  Same windows, both geometries, against the oracle."""
    monkeypatch.setenv("HYPO_POA_CLASS0", lanes)
    rng = np.random.default_rng(900)
    wins = []
    while len(wins) < 4000:
        w = _window(rng, False)
        if len(w.draft) <= 44:
            wins.append(w)
    b = build_batch(wins)
    scores = (5, -4, -8, 3, -5, -4)
    cons, st = gpu.poa_consensus(b, scores)
    ocons, ost = oracle_lib.poa_batch(b, scores=scores)[:2]
    assert [i for i in range(len(wins)) if cons[i] != ocons[i] or st[i] != ost[i]] == []
    s = gpu.last_stats()
    assert s["n_failed"] == 0 and s["n_class"][0] > 2000


@pytest.mark.parametrize("seed,scores", [(11, (5, -4, -8, 3, -5, -4)), (12, (2, -1, -2, 3, -5, -4)), (13, (4, -3, -5, 3, -5, -4)),
                                         (14, (3, -6, -5, 3, -5, -4))])
def test_one_substitution_shapes_vs_oracle(gpu, oracle_lib, seed, scores):
    """The windows of tests/test_poa_emulator.py::_one_sub_windows (arms one base off the arm before them over two- and
    three-letter drafts, arms a base short or long, repeated substitutions) on the device: Poa::guided_one_sub and
    Poa::topo_insert under three score sets that allow the shortcut (other margins between mismatch, gap and two gaps) and one
    that must refuse it."""
    from test_poa_emulator import _one_sub_windows
    rng = np.random.default_rng(8800 + seed)
    wins = _one_sub_windows(rng, 6000)
    b = build_batch(wins)
    cons, st = gpu.poa_consensus(b, scores)
    ocons, ost = oracle_lib.poa_batch(b, scores=scores)[:2]
    bad = [i for i in range(len(wins)) if cons[i] != ocons[i] or st[i] != ost[i]]
    assert not bad, f"{len(bad)} of {len(wins)} windows differ; first: window {bad[0]}"
    assert gpu.last_stats()["n_failed"] == 0


def test_klov_one_substitution_with_a_second_end(gpu, oracle_lib):
    """The window of tests/test_poa_emulator.py::test_klov_one_substitution_with_a_second_end on the device, alone and among
    copies (every class-0 / class-1 lane group of a wave busy with it)."""
    from test_poa_emulator import _second_end_window
    b = build_batch([_second_end_window() for _ in range(300)])
    cons, st = gpu.poa_consensus(b, (5, -4, -8, 3, -5, -4))
    ocons, ost = oracle_lib.poa_batch(b)[:2]
    assert [i for i in range(300) if cons[i] != ocons[i] or st[i] != ost[i]] == []


def test_windows_beyond_the_table_driven_classes_run_in_class_6(gpu, oracle_lib):
    """What used to keep its draft with HYPO_ST_CAPACITY: a LONG window in which two long-read arms carry a 1 500-base insertion (sequences
    beyond class 5's 1 021 bases), a SHORT window with a 1 300-base draft, and the 20 000 short arms of a collapsed repeat (class 5 holds
    16 382 sequences) run in size class 6 (hypo_amd/csrc/poa_giant.hpp) and answer what hypo::Window::generate_consensus answers
    (oracle/_ref/libhyporef.so; the reference has no size limit: external/spoa/src/sisd_alignment_engine.cpp:60-93, graph.cpp:99-128)."""
    import oracle
    from test_giant import giant_windows
    rng = np.random.default_rng(606)
    wins = giant_windows(rng, deep_arms=20000) + [_window(rng, False) for _ in range(200)]
    b = build_batch(wins)
    off = b.slot_layout()
    bases, _, ln, st = gpu.poa_batch(b, off=off)
    s = gpu.last_stats()
    assert (st == 0).all() and s["n_failed"] == 0, (st[:3], s)
    assert s["n_class"][6] == 3, s["n_class"]                    # they really ran there
    ob, _, oln, ost, cells, aligns = oracle_lib.poa_batch_raw(b, off=off)
    assert (ost == 0).all() and (ln == oln).all()
    got = [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes() for i in range(b.n_windows)]
    assert got == [ob[int(off[i]):int(off[i]) + int(oln[i])].tobytes() for i in range(b.n_windows)]
    assert s["dp_cells"] == cells and s["n_alignments"] == aligns
    if oracle.Ref.available():
        rb, _, rln, rst, _ = oracle.Ref().poa_batch_raw(b, off=off)
        assert (rst == 0).all() and got == [rb[int(off[i]):int(off[i]) + int(rln[i])].tobytes() for i in range(b.n_windows)]
    # a context without the arena answers HYPO_ST_CAPACITY for them (and only for them)
    import ctypes as C
    assert gpu.lib.hypo_gpu_set_option(b"giant_arena_mb", C.c_int(1)) == 0          # 1 MB: a slice of 512 KB holds the deep window's small graph, not the score matrices of the other two
    try:
        g2 = type(gpu)(0)                                        # (re-initialises the library's context: its POA state is created anew)
        bases2, _, ln2, st2 = g2.poa_batch(b, off=off)
        assert list(st2[:3]) == [2, 2, 0] and (st2[3:] == 0).all() and g2.last_stats()["n_failed"] == 2
        assert gpu.lib.hypo_gpu_set_option(b"giant_arena_mb", C.c_int(0)) == 0      # no class 6 at all
        g3 = type(gpu)(0)
        bases3, _, ln3, st3 = g3.poa_batch(b, off=off)
        assert list(st3[:3]) == [2, 2, 2] and (st3[3:] == 0).all()
    finally:
        assert gpu.lib.hypo_gpu_set_option(b"giant_arena_mb", C.c_int(1024)) == 0
        type(gpu)(0)
