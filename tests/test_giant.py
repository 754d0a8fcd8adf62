"""Size class 6 of the POA kernel (hypo_amd/csrc/poa_giant.hpp: the windows beyond the table-driven classes, the reference's procedure carried
out literally with the window's state in a slice of HBM) in the lockstep emulator on the CPU: against the REAL reference on slices of the
bounded-exhaustive spaces and on windows only this class can hold (an arm with a 1 500-base insertion in a LONG window, a draft of 1 300
bases, many sequences), against the oracle on simulator batches, and its behaviour when the slice is too small."""
import numpy as np
import pytest

import emu_util
import exhaustive_parity as ex
from hypo_amd import sim
from hypo_amd.batch import TextWindow, build_batch


@pytest.fixture(scope="module")
def emu():
    return emu_util.Emu()


def _mutate(rng, s, err):
    out = []
    for ch in s:
        r = rng.random()
        if r < err / 3:
            continue
        if r < 2 * err / 3:
            out.append("ACGT"[rng.integers(0, 4)])
        else:
            out.append(ch)
        if rng.random() < err / 3:
            out.append("ACGT"[rng.integers(0, 4)])
    return "".join(out) or "A"


def giant_windows(rng, deep_arms=400):
    """Windows beyond class 5's tables: (1) LONG, two arms carry the same 1 500-base insertion; (2) SHORT with a 1 300-base draft and
    arms of that length; (3) SHORT with `deep_arms` short arms (the GPU test uses 20 000: more than class 5's 16 382 sequences)."""
    A = "ACGT"
    rnd = lambda n: "".join(A[i] for i in rng.integers(0, 4, size=n))
    t1 = rnd(500)
    ins = rnd(1500)
    arms1 = [_mutate(rng, t1, 0.04) for _ in range(7)]
    for k in (2, 5):
        a = _mutate(rng, t1, 0.04)
        arms1.append(a[:250] + ins + a[250:])
    w1 = TextWindow(_mutate(rng, t1, 0.02), arms1, [], [], 0, True)
    t2 = rnd(1300)
    w2 = TextWindow(_mutate(rng, t2, 0.01), [_mutate(rng, t2, 0.01) for _ in range(5)], [t2[:700]], [t2[500:]], 0, False)
    t3 = rnd(50)
    w3 = TextWindow(_mutate(rng, t3, 0.02), [_mutate(rng, t3, 0.02) for _ in range(deep_arms)], [t3[:30], t3[:41]], [t3[20:], t3[33:]], 0, False)
    return [w1, w2, w3]


def test_giant_vs_oracle_on_simulator_batches(emu, oracle_lib):
    for b, scores in ((sim.window_batch(200, seed=21), (5, -4, -8, 3, -5, -4)), (sim.window_batch(120, seed=22, read_sub=0.03), (3, -6, -5, 3, -5, -4)),
                      (sim.c4_batch(30, 6, seed=23), (5, -4, -8, 3, -5, -4))):
        off = b.slot_layout()
        cons, st, res, cells, aligns = emu.poa_giant(b, scores=scores, off=off)
        want, wst, wc, wa = oracle_lib.poa_batch(b, scores=scores, off=off)
        assert (res == emu_util.RES_OK).all() and list(st) == list(wst)
        assert cons == want and cells == wc and aligns == wa


@pytest.mark.parametrize("name,stride", [("a2n3", 9000), ("a3n3", 1500), ("a2n4", 2300)])
def test_giant_vs_real_reference_on_exhaustive_slices(emu, name, stride):
    import oracle
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref/libhyporef.so not built (the real reference only exists in the build container)")
    ref = oracle.Ref()
    b = next(ex.chunks(name, 1500, stride=stride, offset=2))
    off = b.slot_layout()
    rb, _, rln, rst, _ = ref.poa_batch_raw(b, off=off)
    cons, st, res, _, _ = emu.poa_giant(b, off=off)
    want = [rb[int(off[i]):int(off[i]) + int(rln[i])].tobytes().decode() for i in range(b.n_windows)]
    bad = [i for i in range(b.n_windows) if res[i] != emu_util.RES_OK or cons[i] != want[i]]
    assert not bad, f"{name}: {ex.describe(b, bad[0])}: class 6 {cons[bad[0]]!r} reference {want[bad[0]]!r}"


def test_windows_only_class_6_holds_vs_real_reference(emu, oracle_lib):
    import oracle
    rng = np.random.default_rng(606)
    wins = giant_windows(rng)
    b = build_batch(wins)
    off = b.slot_layout()
    cons, st, res, cells, aligns = emu.poa_giant(b, off=off, slice_bytes=96 << 20)
    assert (res == emu_util.RES_OK).all() and (st == 0).all()
    want, wst, wc, wa = oracle_lib.poa_batch(b, off=off)
    assert cons == want and cells == wc and aligns == wa
    if oracle.Ref.available():
        ref = oracle.Ref()
        rb, _, rln, rst, _ = ref.poa_batch_raw(b, off=off)
        assert (rst == 0).all(), "the LONG window's arms must pass the reference's own filter (one shared minimizer per 50 bases)"
        assert cons == [rb[int(off[i]):int(off[i]) + int(rln[i])].tobytes().decode() for i in range(b.n_windows)]
    # the classes below cannot take them: class 5 answers "overflow" (the kernel then queues the window for class 6)
    c5, st5, res5, _, _ = emu.poa_batch(build_batch(wins[:2]), 5)
    assert all(r in (emu_util.RES_OVERFLOW, emu_util.RES_UNSUPPORTED) for r in res5)


def test_a_slice_that_is_too_small_is_reported_not_overrun(emu):
    rng = np.random.default_rng(607)
    b = build_batch(giant_windows(rng, deep_arms=50)[:2])
    cons, st, res, _, _ = emu.poa_giant(b, slice_bytes=1 << 20)           # 1 MB: the score matrix of neither window fits
    assert all(r == emu_util.RES_OVERFLOW for r in res) and cons == [None, None]


def _asan_check():
    """(child process of the test below) slices of exactly the size handed over, garbage-filled, under AddressSanitizer"""
    import oracle
    e = emu_util.Emu(asan=True)
    orc = oracle.Oracle()
    small = sim.window_batch(16, seed=3)
    want = orc.poa_batch(small)[0]
    cons, st, res, _, _ = e.poa_giant(small, slice_bytes=1 << 20)
    assert all(r == emu_util.RES_OK for r in res) and cons == want
    n_over = 0
    for kb in (32, 96):                                             # slices at the edge of what a window needs: answered or refused, never overrun
        cons, st, res, _, _ = e.poa_giant(small, slice_bytes=kb << 10)
        for i in range(small.n_windows):
            assert res[i] in (emu_util.RES_OK, emu_util.RES_OVERFLOW)
            assert res[i] != emu_util.RES_OK or cons[i] == want[i]
            n_over += res[i] == emu_util.RES_OVERFLOW
    lng = sim.c4_batch(2, 2, seed=9)
    cons, st, res, _, _ = e.poa_giant(lng, slice_bytes=8 << 20)
    assert all(r == emu_util.RES_OK for r in res) and cons == orc.poa_batch(lng)[0]
    return n_over


def test_giant_under_asan():
    import os
    import subprocess
    import sys
    emu_util.build()
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path[:0] = [%r, %r]; import test_giant as t; print('refused', t._asan_check())" % (os.path.dirname(here), here)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "refused" in p.stdout, (p.stdout[-400:], p.stderr[-1500:])
