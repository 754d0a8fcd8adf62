"""Opt-in native flavour (--native-klov / hypo_gpu_set_option("native_klov", 1)): the kLOV end-row rule of the reference's AVX2 /
SSE4.1 engine.  Goldens from the real reference classes built with -march=native (tests/golden/make_native_golden.py): 400
windows with noisy prefix arms, 250 of them polished differently by the two flavours."""
import os

import pytest

from hypo_amd.batch import build_batch
import golden_util as gu


def _cases():
    recs = gu.load_jsonl("windows_native_klov.jsonl.gz")
    wins = [gu.to_window(r) for r in recs]
    return recs, wins, build_batch(wins)


def test_oracle_both_flavours(oracle_lib):
    recs, wins, b = _cases()
    oracle_lib.lib.oracle_set_native_klov.restype = None
    try:
        oracle_lib.lib.oracle_set_native_klov(1)
        nat = oracle_lib.poa_batch(b)[0]
    finally:
        oracle_lib.lib.oracle_set_native_klov(0)
    sca = oracle_lib.poa_batch(b)[0]
    assert nat == [r["consensus_native"] for r in recs]
    assert sca == [r["consensus_scalar"] for r in recs]
    assert sum(1 for r in recs if r["consensus_native"] != r["consensus_scalar"]) >= 200


def test_kernel_source_in_the_emulator():
    import emu_util
    recs, wins, b = _cases()
    emu = emu_util.Emu()
    os.environ["HYPO_EMU_NATIVE_KLOV"] = "1"
    try:
        n = 0
        for cfg in (1, 2, 3):
            cons, st, res, _, _ = emu.poa_batch(b, cfg)
            for r, c, rc in zip(recs, cons, res):
                if rc == emu_util.RES_OK:
                    assert c == r["consensus_native"], cfg
                    n += 1
        assert n > 400
    finally:
        del os.environ["HYPO_EMU_NATIVE_KLOV"]
    cons, st, res, _, _ = emu.poa_batch(b, 3)
    assert all(c == r["consensus_scalar"] for r, c, rc in zip(recs, cons, res) if rc == emu_util.RES_OK)


@pytest.mark.gpu
def test_device_both_flavours():
    import ctypes as C
    from hypo_amd import capi
    recs, wins, b = _cases()
    gpu = capi.HypoGpu(0)
    try:
        assert gpu.lib.hypo_gpu_set_option(b"native_klov", C.c_int(1)) == 0
        nat, st = gpu.poa_consensus(b)
    finally:
        assert gpu.lib.hypo_gpu_set_option(b"native_klov", C.c_int(0)) == 0
    sca, st2 = gpu.poa_consensus(b)
    assert (st == 0).all() and (st2 == 0).all()
    assert nat == [r["consensus_native"] for r in recs]
    assert sca == [r["consensus_scalar"] for r in recs]
    assert gpu.lib.hypo_gpu_set_option(b"no_such_option", C.c_int(1)) != 0
