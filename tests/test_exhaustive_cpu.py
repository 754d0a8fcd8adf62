"""Bounded-exhaustive windows (tests/exhaustive_parity.py) on the CPU: the enumerator itself, the oracle against the REAL reference on a
strided slice of every space, and the lockstep emulator of hypo_amd/csrc/poa_core.hpp (every short size class's code, both class-0
geometries) against the real reference on slices of the three-arm and four-arm spaces.  The whole spaces run on the GPU box
(tests/test_gpu_exhaustive.py, profiles/r06_exhaustive.txt)."""
import numpy as np
import pytest

import exhaustive_parity as ex


def test_space_sizes_and_enumeration_is_complete():
    assert [ex.space_size(s) for s in ("a2n2", "a2n3", "a2n4", "a3n2", "a3n3")] == [12_002_256, 147_763_360, 17_287_200, 10_368_000, 23_134_410]
    # one small configuration, whole: every (draft, arm, arm) combination exactly once
    s, dl, lens, kinds = 2, 2, (1, 2), (1, 0, 1)
    b = ex.build_config(s, dl, lens, kinds, 0, s ** (dl + sum(lens)))
    seen = {ex.describe(b, w) for w in range(b.n_windows)}
    assert len(seen) == 32 and "draft CA arms ['C', 'AC'] internal/prefix/suffix 1/0/1" in seen
    # chunks() covers a space exactly once, whatever the chunk size
    tot = sum(c.n_windows for c in ex.chunks("a3n2", 700_000))
    assert tot == ex.space_size("a3n2")
    c = next(ex.chunks("a3n2", 5000))
    assert c.windows.dtype.itemsize == 40 and int(c.windows["first_arm"][-1]) + 2 == c.n_arms


def test_oracle_vs_reference_on_a_slice_of_every_space():
    import oracle
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref/libhyporef.so not built (the real reference only exists in the build container)")
    ref, orc = oracle.Ref(), oracle.Oracle()
    total = 0
    for name in ex.SPACES:
        for i, b in enumerate(ex.chunks(name, 150_000, stride=97, offset=3)):
            off = b.slot_layout()
            for sc in (ex.SCORE_SETS[0], ex.SCORE_SETS[1 + i % 2]):
                rb, _, rln, rst, _ = ref.poa_batch_raw(b, scores=sc, off=off)
                ob, _, oln, ost = orc.poa_batch_raw(b, scores=sc, off=off)[:4]
                w = ex.first_difference((ob, oln, ost), (rb, rln, rst), off)
                assert w < 0, f"{name} scores {sc}: {ex.describe(b, w)}"
            total += b.n_windows
    assert total > 1_500_000, total


@pytest.mark.parametrize("name,stride", [("a2n3", 2300), ("a3n3", 370), ("a2n4", 560)])
def test_emulator_vs_reference_on_exhaustive_slices(name, stride):
    import emu_util
    import oracle
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref/libhyporef.so not built (the real reference only exists in the build container)")
    emu, ref = emu_util.Emu(), oracle.Ref()
    b = next(ex.chunks(name, 4000, stride=stride, offset=1))
    off = b.slot_layout()
    rb, _, rln, rst, _ = ref.poa_batch_raw(b, off=off)
    want = [rb[int(off[i]):int(off[i]) + int(rln[i])].tobytes().decode() for i in range(b.n_windows)]
    for cfg in (0, 6, 1, 2, 3):                       # class 0 (16- and 32-lane groups), 1, 2, 3
        cons, st, res, _, _ = emu.poa_batch(b, cfg, off=off)
        bad = [i for i in range(b.n_windows) if res[i] != emu_util.RES_OK or cons[i] != want[i]]
        assert not bad, f"{name} class config {cfg}: {ex.describe(b, bad[0])}: emulator {cons[bad[0]]!r} reference {want[bad[0]]!r}"
