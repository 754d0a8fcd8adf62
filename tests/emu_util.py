"""Driver of the TEST-ONLY lockstep emulator (tests/emu/) of hypo_amd/csrc/poa_core.hpp."""
import ctypes as C
import os
import subprocess

import numpy as np

from hypo_amd import abi
import oracle as _oracle

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")

RES_OK, RES_OVERFLOW, RES_UNDEFINED, RES_CONS_OVERFLOW, RES_UNSUPPORTED = range(5)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu")])


class Emu:
    def __init__(self, asan=False, exact=True):
        build()
        name = "libhypo_emu_asan.so" if asan else ("libhypo_emu.so" if exact else "libhypo_emu_noexact.so")
        name = os.environ.get("HYPO_EMU_LIB", name)          # (a differently built emulator, for A/B runs)
        self.lib = C.CDLL(os.path.join(BUILD, name))
        self.lib.emu_poa_batch.restype = C.c_int
        self.lib.emu_class_bytes.restype = C.c_int

    def class_bytes(self, cfg):
        return int(self.lib.emu_class_bytes(C.c_int(cfg)))

    def poa_batch(self, b, cfg, scores=abi.DEFAULT_SCORES, off=None):
        sp = abi.ScoreParams(*scores)
        if off is None:
            off = b.slot_layout()
        n = b.n_windows
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        res = np.zeros(n, dtype=np.uint8)
        ins = _oracle.batch_struct(b)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        out = abi.ConsensusBatch(p(bases), p(off), p(ln), p(st))
        cells, aligns = C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.emu_poa_batch(C.byref(sp), C.byref(ins), C.byref(out), C.c_int(cfg), p(res),
                                    C.byref(cells), C.byref(aligns))
        assert rc == 0
        cons = [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes().decode()
                if (res[i] == RES_OK and st[i] == 0) else None for i in range(n)]
        return cons, st, res, int(cells.value), int(aligns.value)

    def poa_chain(self, b, cfg_from, scores=abi.DEFAULT_SCORES, off=None, use_carry=True):
        """Every window starts in class cfg_from and follows the kernel's re-queue chain (emu_poa_chain), taking its graph along
        when use_carry.  Returns (consensus strings, result codes, classes visited, hops that started from a spill)."""
        sp = abi.ScoreParams(*scores)
        if off is None:
            off = b.slot_layout()
        n = b.n_windows
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        res = np.zeros(n, dtype=np.uint8)
        hops = np.zeros(n, dtype=np.uint8)
        carried = np.zeros(n, dtype=np.uint8)
        ins = _oracle.batch_struct(b)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        out = abi.ConsensusBatch(p(bases), p(off), p(ln), p(st))
        self.lib.emu_poa_chain.restype = C.c_int
        rc = self.lib.emu_poa_chain(C.byref(sp), C.byref(ins), C.byref(out), C.c_int(cfg_from), p(res), p(hops), p(carried),
                                    C.c_int(1 if use_carry else 0))
        assert rc == 0
        cons = [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes().decode()
                if (res[i] == RES_OK and st[i] == 0) else None for i in range(n)]
        return cons, res, hops, carried

    def poa_giant(self, b, scores=abi.DEFAULT_SCORES, off=None, slice_bytes=1 << 24):
        """Every window through size class 6 (hypo_amd/csrc/poa_giant.hpp: Giant::run with a slice of `slice_bytes`).  Returns
        (consensus strings or None, status, result codes, cells, alignments)."""
        sp = abi.ScoreParams(*scores)
        if off is None:
            off = b.slot_layout()
        n = b.n_windows
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        res = np.zeros(n, dtype=np.uint8)
        ins = _oracle.batch_struct(b)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        out = abi.ConsensusBatch(p(bases), p(off), p(ln), p(st))
        cells, aligns = C.c_uint64(0), C.c_uint64(0)
        self.lib.emu_poa_giant.restype = C.c_int
        rc = self.lib.emu_poa_giant(C.byref(sp), C.byref(ins), C.byref(out), C.c_uint64(slice_bytes), p(res), C.byref(cells), C.byref(aligns))
        assert rc == 0
        cons = [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes().decode() if (res[i] == RES_OK and st[i] == 0) else None for i in range(n)]
        return cons, st, res, int(cells.value), int(aligns.value)
