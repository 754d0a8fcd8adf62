"""CPU: the C restatement (oracle/) against every committed golden vector of the real reference."""
import numpy as np
import pytest

from hypo_amd.batch import build_batch, pack2, pack4, unpack2, unpack4
import golden_util as gu


@pytest.mark.parametrize("name", gu.WINDOW_FILES)
def test_window_goldens(oracle_lib, name):
    total = 0
    for scores, items in gu.windows_by_scores(name).items():
        b = build_batch([w for w, _, _ in items])
        cons, st, _, _ = oracle_lib.poa_batch(b, scores=scores)
        for (w, want, tag), got, s in zip(items, cons, st):
            assert s == 0, (tag, s)
            assert got == want, (tag, scores)
            total += 1
    assert total > 100


def test_replay_goldens(oracle_lib):
    cases = gu.load_jsonl("replay_cases.jsonl.gz")
    for c in cases:
        rc, pairs, rank, cons = oracle_lib.replay(c["seqs"], c["modes"], c["scores"])
        assert rc == 0
        assert pairs.reshape(-1).tolist() == c["pairs"]
        assert rank.tolist() == c["rank"]
        assert cons == c["consensus"]


def test_spoa_global_consensus_pin(oracle_lib):
    """spoa's own known answer (external/spoa/test/spoa_test.cpp:220-239): kNW, 5/-4/-8 linear."""
    g = gu.load_json("spoa_sample.json.gz")
    rc, _, rank, cons = oracle_lib.replay(g["reads"], [1] * len(g["reads"]), g["scores"])
    assert rc == 0
    assert cons == g["consensus"]
    assert rank.size == g["rank_len"]


def test_packedseq_goldens():
    import ctypes as C
    import oracle
    lib = oracle.Oracle().lib
    for c in gu.load_json("packedseq_cases.json.gz"):
        t = c["text"]
        n = len(t)
        if c["nb"] == 2:
            p = pack2(t)
            assert unpack2(p, n) == c["unpacked"]
            buf = np.zeros(max((n + 3) // 4, 1), np.uint8)
            lib.oracle_pack2(t.encode(), C.c_uint32(n), buf.ctypes.data_as(C.c_void_p))
            assert (buf[:p.size] == p).all()
            out = C.create_string_buffer(n + 1)
            lib.oracle_unpack2(buf.ctypes.data_as(C.c_void_p), C.c_uint32(n), out)
        else:
            p = pack4(t)
            assert unpack4(p, n) == c["unpacked"]
            buf = np.zeros(max((n + 1) // 2, 1), np.uint8)
            lib.oracle_pack4(t.encode(), C.c_uint32(n), buf.ctypes.data_as(C.c_void_p))
            assert (buf[:p.size] == p).all()
            out = C.create_string_buffer(n + 1)
            lib.oracle_unpack4(buf.ctypes.data_as(C.c_void_p), C.c_uint32(n), out)
        assert out.raw[:n].decode() == c["unpacked"]


def test_dispatch_rules(oracle_lib):
    """Window::generate_consensus dispatch (src/Window.cpp:44-61)."""
    from hypo_amd.batch import TextWindow
    ws = [
        TextWindow("ACGTNACGT", ["ACGTACGT"], [], [], 0),             # 1 arm  -> draft text (N kept)
        TextWindow("ACGTACGT", ["ACGTACGT", "ACGTACGT"], [], [], 3),  # empties win -> ""
        TextWindow("ACGTACGT", [], [], [], 0),                        # no arms -> draft
        TextWindow("ACGTACGT", ["", ""], [], [], 0),                  # only zero-length arms -> draft (Window.cpp:149-151)
        TextWindow("ACGAACGT", ["ACGTACGT", "ACGTACGT", "ACGTACGT"], [], [], 3),  # empties == arms -> POA
    ]
    cons, st, _, _ = oracle_lib.poa_batch(build_batch(ws))
    assert list(st) == [0] * 5
    assert cons == ["ACGTNACGT", "", "ACGTACGT", "ACGTACGT", "ACGTACGT"]


def test_consensus_slot_overflow(oracle_lib):
    from hypo_amd.batch import TextWindow
    b = build_batch([TextWindow("ACGTACGTAC", ["ACGTACGTAC"] * 3)])
    off = np.array([0, 4], dtype=np.uint64)
    cons, st, _, _ = oracle_lib.poa_batch(b, off=off)
    assert st[0] == 1 and cons[0] is None


def test_oracle_branch_completion_windows(oracle_lib):
    """the LONG windows that exercise branch completion with a score tie (expected values from the real reference)"""
    from hypo_amd.batch import TextWindow, build_batch
    doc = gu.load_json("windows_branch_completion.json.gz")
    for c in doc["windows"]:
        w = TextWindow(c["draft"], c["internal"], c["prefix"], c["suffix"], n_empty=c["n_empty"], is_long=c["long"])
        assert oracle_lib.poa_batch(build_batch([w]), scores=tuple(c["scores"]))[0][0] == c["consensus"], c["tag"]
