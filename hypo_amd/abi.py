"""ctypes mirror of include/hypo_gpu.h (structs and constants only; no library is loaded here).

Reference types mirrored: ScoreParams (include/globalDefs.hpp:58-66), hypo::Window data members
(include/Window.hpp:123-135).
"""
import ctypes as C

ABI_VERSION = 8

HYPO_OK = 0
HYPO_E_INVALID = -1
HYPO_E_NODEVICE = -2
HYPO_E_HIP = -3
HYPO_E_WORKSPACE = -4
HYPO_E_NOTINIT = -5
HYPO_E_CAPACITY = -6
HYPO_E_UNSUPPORTED = -7

ST_OK = 0
ST_CONS_OVERFLOW = 1
ST_CAPACITY = 2
ST_UNDEFINED = 3
ST_INVALID = 4

WIN_SHORT = 0
WIN_LONG = 1


class ScoreParams(C.Structure):
    _fields_ = [("sr_match", C.c_int8), ("sr_mismatch", C.c_int8), ("sr_gap", C.c_int8),
                ("lr_match", C.c_int8), ("lr_mismatch", C.c_int8), ("lr_gap", C.c_int8)]


# reference defaults: src/main.cpp:100-113
DEFAULT_SCORES = (5, -4, -8, 3, -5, -4)


class Window(C.Structure):
    _fields_ = [("type", C.c_uint8), ("reserved", C.c_uint8 * 3), ("draft_len", C.c_uint32),
                ("draft_off", C.c_uint64), ("first_arm", C.c_uint32), ("n_internal", C.c_uint32),
                ("n_prefix", C.c_uint32), ("n_suffix", C.c_uint32), ("n_empty", C.c_uint32),
                ("reserved2", C.c_uint32)]


assert C.sizeof(Window) == 40


class WindowBatch(C.Structure):
    _fields_ = [("n_windows", C.c_uint32), ("n_arms", C.c_uint32), ("windows", C.c_void_p),
                ("draft4", C.c_void_p), ("draft4_bytes", C.c_uint64), ("arm_off", C.c_void_p),
                ("arm_len", C.c_void_p), ("arms2", C.c_void_p), ("arms2_bytes", C.c_uint64)]


class ConsensusBatch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p),
                ("status", C.c_void_p)]


class PoaStats(C.Structure):
    _fields_ = [("n_windows", C.c_uint64), ("n_trivial", C.c_uint64), ("n_class", C.c_uint64 * 8),
                ("n_escalated", C.c_uint64), ("n_failed", C.c_uint64), ("dp_cells", C.c_uint64),
                ("n_alignments", C.c_uint64), ("alg_bytes", C.c_uint64 * 8),
                ("n_reused", C.c_uint64), ("n_threaded", C.c_uint64), ("cells_scored", C.c_uint64),
                ("cells_threaded", C.c_uint64), ("n_carried", C.c_uint64)]


# numpy dtype equivalent of HypoWindow (40 bytes, same offsets)
import numpy as _np

WINDOW_DTYPE = _np.dtype({
    "names": ["type", "draft_len", "draft_off", "first_arm", "n_internal", "n_prefix", "n_suffix",
              "n_empty"],
    "formats": ["u1", "<u4", "<u8", "<u4", "<u4", "<u4", "<u4", "<u4"],
    "offsets": [0, 4, 8, 16, 20, 24, 28, 32],
    "itemsize": 40,
})
