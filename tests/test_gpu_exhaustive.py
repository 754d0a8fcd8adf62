"""Bounded-exhaustive parity on the device against the REAL reference (tests/exhaustive_parity.py): the one-minute subset of the suite —
every window of the two-arm spaces over {A,C} (12.0 M) and {A,C,G} (10.4 M) in every short size class of the kernel; the five spaces
under three score sets are the opt-in run behind profiles/r06_exhaustive.txt."""
import pytest

import exhaustive_parity as ex

pytestmark = pytest.mark.gpu


def test_two_arm_spaces_in_every_size_class_vs_real_reference():
    import oracle
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref/libhyporef.so not built (the real reference only exists in the build container)")
    msgs = []
    r = ex.run(["a2n2", "a3n2"], ex.SCORE_SETS[:1], ["0", "0w", "1", "2", "3"], chunk=2_000_000, budget=150, log=msgs.append)
    assert r is not None, "\n".join(msgs)
    print("\n".join(msgs))
    assert r["windows"] >= 12_000_000, r


def test_three_and_four_arm_slices_under_other_scores_vs_real_reference():
    import oracle
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref/libhyporef.so not built (the real reference only exists in the build container)")
    msgs = []
    r = ex.run(["a2n3", "a3n3", "a2n4"], ex.SCORE_SETS, ["0", "0w", "1", "2", "3"], chunk=1_000_000, stride=29, budget=150, log=msgs.append)
    assert r is not None, "\n".join(msgs)
    print("\n".join(msgs))
    assert r["windows"] >= 3_000_000, r
