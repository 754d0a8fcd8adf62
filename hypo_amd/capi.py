"""ctypes binding of libhypo_gpu.so (the C-ABI of include/hypo_gpu.h) + torch device-memory plumbing.

There is no Python or CPU implementation of the hot path behind this module: if the HIP library is
missing or no gfx950 device is present, loading / init raises.
"""
import ctypes as C
import os

# more hardware queues than ROCm's default of four per process (the library runs up to seven streams); only effective when the
# HIP runtime has not been initialised yet — see INTEGRATION.md
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import subprocess

import numpy as np

from . import abi
from .batch import HostBatch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libhypo_gpu.so")

EXPORTS = [
    "hypo_gpu_init", "hypo_gpu_shutdown", "hypo_gpu_abi_version", "hypo_gpu_last_error",
    "hypo_gpu_num_cus", "hypo_gpu_solid_scan", "hypo_gpu_solid_scan_workspace_bytes",
    "hypo_gpu_solid_scan_device", "hypo_gpu_poa_batch", "hypo_gpu_poa_workspace_bytes",
    "hypo_gpu_poa_batch_device", "hypo_gpu_poa_slot_layout", "hypo_gpu_poa_last_stats",
    "hypo_gpu_poa_read_stats", "hypo_gpu_profile_begin", "hypo_gpu_profile_calls", "hypo_gpu_profile_read",
    "hypo_gpu_num_devices", "hypo_gpu_use_device", "hypo_gpu_build_id", "hypo_gpu_solid_set_upload",
    "hypo_gpu_poa_batch_sharded", "hypo_gpu_poa_batch_begin", "hypo_gpu_poa_batch_end", "hypo_gpu_set_option",
    "hypo_gpu_arms_build", "hypo_gpu_arms_download", "hypo_gpu_arms_poa",
    "hypo_gpu_arms_build_long", "hypo_gpu_arms_download_long", "hypo_gpu_arms_poa_long",
    "hypo_gpu_reads_upload", "hypo_gpu_support_kmers", "hypo_gpu_support_minimizers",
    "hypo_gpu_solid_scan_keep", "hypo_gpu_solid_release", "hypo_gpu_support_kmers_kept", "hypo_gpu_host_alloc", "hypo_gpu_host_free", "hypo_gpu_host_register", "hypo_gpu_host_unregister",
]


class HypoGpuError(RuntimeError):
    pass


def build_library() -> str:
    """Compiles hypo_amd/csrc for gfx950 with hipcc (works without a GPU)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(HERE, "csrc")])
    return LIB_PATH


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise HypoGpuError(f"{path} is missing: build it with `make -C hypo_amd/csrc` "
                           "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7; importing it first makes the
    # loader resolve this library's NEEDED libamdhip64.so.7 to the same copy (two runtimes in one
    # process cannot both own the GPU).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    lib.hypo_gpu_last_error.restype = C.c_char_p
    lib.hypo_gpu_build_id.restype = C.c_char_p
    lib.hypo_gpu_poa_workspace_bytes.restype = C.c_size_t
    lib.hypo_gpu_solid_scan_workspace_bytes.restype = C.c_size_t
    lib.hypo_gpu_solid_scan_workspace_bytes.argtypes = [C.c_uint64]
    lib.hypo_gpu_poa_workspace_bytes.argtypes = [C.c_uint32, C.c_uint32]
    for name in EXPORTS:
        getattr(lib, name)
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def host_struct(b: HostBatch) -> abi.WindowBatch:
    s = abi.WindowBatch()
    s.n_windows, s.n_arms = b.n_windows, b.n_arms
    s.windows, s.draft4, s.draft4_bytes = _p(b.windows), _p(b.draft4), b.draft4.size
    s.arm_off, s.arm_len, s.arms2, s.arms2_bytes = _p(b.arm_off), _p(b.arm_len), _p(b.arms2), b.arms2.size
    return s


class HypoGpu:
    """One process = one GPU (rank-local), mirroring the reference's single Hypo object."""

    def __init__(self, device: int = 0, path: str = LIB_PATH, devices=None):
        """devices: list of HIP device ids for several contexts in this process (hypo_gpu_poa_batch_sharded); the torch
        plumbing of the *_device entry points uses context 0 = `device`."""
        self.lib = load_library(path)
        if self.lib.hypo_gpu_abi_version() != abi.ABI_VERSION:
            raise HypoGpuError("ABI version mismatch between hypo_amd/abi.py and libhypo_gpu.so")
        devices = [device] if devices is None else list(devices)
        self.device = devices[0]
        ids = (C.c_int * len(devices))(*devices)
        self._check(self.lib.hypo_gpu_init(ids, C.c_int(len(devices))))
        self.num_cus = int(self.lib.hypo_gpu_num_cus())

    def _check(self, rc):
        if rc != 0:
            raise HypoGpuError(f"libhypo_gpu rc={rc}: {self.lib.hypo_gpu_last_error().decode()}")

    # ---- host-buffer entry points (H2D + kernels + D2H inside the library) -----------------------------
    def poa_batch(self, b: HostBatch, scores=abi.DEFAULT_SCORES, off=None):
        """Returns (bases u8, off u64, len u32, status u8)."""
        sp = abi.ScoreParams(*scores)
        n = b.n_windows
        if off is None:
            off = np.zeros(n + 1, dtype=np.uint64)
            ins = host_struct(b)
            self._check(self.lib.hypo_gpu_poa_slot_layout(C.byref(ins), _p(off)))
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        ins = host_struct(b)
        out = abi.ConsensusBatch(_p(bases), _p(off), _p(ln), _p(st))
        self._check(self.lib.hypo_gpu_poa_batch(C.byref(sp), C.byref(ins), C.byref(out)))
        return bases, off, ln, st

    def poa_batch_sharded(self, b: HostBatch, scores=abi.DEFAULT_SCORES, off=None):
        """hypo_gpu_poa_batch over all contexts of this process.  Returns (bases u8, off u64, len u32, status u8)."""
        sp = abi.ScoreParams(*scores)
        n = b.n_windows
        if off is None:
            off = b.slot_layout()
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        ins = host_struct(b)
        out = abi.ConsensusBatch(_p(bases), _p(off), _p(ln), _p(st))
        self._check(self.lib.hypo_gpu_poa_batch_sharded(C.byref(sp), C.byref(ins), C.byref(out)))
        return bases, off, ln, st

    def poa_batch_begin(self, b: HostBatch, off, bases, ln, st, scores=abi.DEFAULT_SCORES, no_arm_off=False):
        """hypo_gpu_poa_batch_begin on caller-owned buffers; returns (ticket, keep-alive objects).  no_arm_off: the arms of `b`
        lie back to back (HostBatch built by this package), let the device compute their offsets."""
        sp = abi.ScoreParams(*scores)
        ins = host_struct(b)
        if no_arm_off:
            ins.arm_off = None
        out = abi.ConsensusBatch(_p(bases), _p(off), _p(ln), _p(st))
        t = C.c_int(-1)
        self._check(self.lib.hypo_gpu_poa_batch_begin(C.byref(sp), C.byref(ins), C.byref(out), C.byref(t)))
        return int(t.value), (sp, ins, out)

    def poa_batch_end(self, ticket: int):
        self._check(self.lib.hypo_gpu_poa_batch_end(C.c_int(ticket)))

    def poa_consensus(self, b: HostBatch, scores=abi.DEFAULT_SCORES, off=None):
        bases, off, ln, st = self.poa_batch(b, scores, off)
        cons = [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes().decode() if st[i] == 0 else None
                for i in range(b.n_windows)]
        return cons, st

    def last_stats(self) -> dict:
        s = abi.PoaStats()
        self._check(self.lib.hypo_gpu_poa_last_stats(C.byref(s)))
        return _stats_dict(s)

    def solid_scan(self, packed4: np.ndarray, n_bases: int, k: int, bits: np.ndarray, kids_cap=None):
        nw = (n_bases + 63) // 64
        words = np.zeros(max(nw, 1), dtype=np.uint64)
        if kids_cap is None:
            kids_cap = n_bases
        kids = np.zeros(max(kids_cap, 1), dtype=np.uint64)
        rank = np.zeros(nw + 1, dtype=np.uint64)
        ns = C.c_uint64(0)
        self._check(self.lib.hypo_gpu_solid_scan(_p(packed4), C.c_uint64(n_bases), C.c_uint32(k), _p(bits),
                                                 _p(words), _p(kids), C.c_uint64(kids_cap), _p(rank),
                                                 C.byref(ns)))
        n = int(ns.value)
        return words[:nw], kids[:min(n, kids_cap)], rank, n

    # ---- HIP-event kernel timing --------------------------------------------------------------------------
    def profile_begin(self, max_calls: int):
        self._check(self.lib.hypo_gpu_profile_begin(C.c_int(max_calls)))

    def profile_read(self):
        """List of per-call lists of elapsed milliseconds (see include/hypo_gpu.h)."""
        out = []
        buf = (C.c_float * 16)()
        for c in range(int(self.lib.hypo_gpu_profile_calls())):
            n = int(self.lib.hypo_gpu_profile_read(C.c_int(c), buf, C.c_int(16)))
            if n < 0:
                self._check(n)
            out.append([float(buf[i]) for i in range(n)])
        return out

    # ---- device-resident entry points (torch tensors own the HBM) --------------------------------------
    def device_batch(self, b: HostBatch, off=None, workspace_bytes=None):
        """workspace_bytes: None = the size hypo_gpu_poa_workspace_bytes recommends"""
        return DeviceBatch(self, b, off, workspace_bytes)

    def device_scan(self, packed4: np.ndarray, n_bases: int, k: int, bits: np.ndarray, kids_cap=None, misalign=0):
        return DeviceScan(self, packed4, n_bases, k, bits, kids_cap, misalign)


def _stats_dict(s: abi.PoaStats) -> dict:
    return {"n_windows": int(s.n_windows), "n_trivial": int(s.n_trivial),
            "n_class": [int(x) for x in s.n_class], "n_escalated": int(s.n_escalated),
            "n_failed": int(s.n_failed), "dp_cells": int(s.dp_cells), "n_alignments": int(s.n_alignments),
            "alg_bytes": [int(x) for x in s.alg_bytes], "n_reused": int(s.n_reused), "n_threaded": int(s.n_threaded),
            "cells_scored": int(s.cells_scored), "cells_threaded": int(s.cells_threaded), "n_carried": int(s.n_carried)}


def _t(arr, dev):
    import torch
    a = np.ascontiguousarray(arr)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    elif a.dtype == np.uint32:
        a = a.view(np.int32)
    elif a.dtype.fields is not None:
        a = a.view(np.uint8)
    if a.size == 0:
        return torch.zeros(16, dtype=torch.uint8, device=dev)
    return torch.from_numpy(a).to(dev)


class DeviceBatch:
    """A window batch resident in HBM + its output buffers and workspace (torch owns the memory)."""

    def __init__(self, gpu: HypoGpu, b: HostBatch, off=None, workspace_bytes=None):
        import torch
        self.gpu, self.host = gpu, b
        dev = torch.device("cuda", gpu.device)
        self.dev = dev
        n = b.n_windows
        if off is None:
            off = b.slot_layout()
        self.off_host = off
        self.windows = _t(b.windows, dev)
        self.draft4, self.arm_off, self.arm_len, self.arms2 = (_t(x, dev) for x in (b.draft4, b.arm_off, b.arm_len, b.arms2))
        self.off = _t(off, dev)
        self.bases = torch.zeros(int(off[-1]) + 16, dtype=torch.uint8, device=dev)
        self.len = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.status = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        wsb = int(gpu.lib.hypo_gpu_poa_workspace_bytes(C.c_uint32(n), C.c_uint32(b.n_arms)))
        if workspace_bytes is not None:
            wsb = int(workspace_bytes)
        self.workspace = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        s = abi.WindowBatch()
        s.n_windows, s.n_arms = n, b.n_arms
        s.windows, s.draft4, s.draft4_bytes = self.windows.data_ptr(), self.draft4.data_ptr(), b.draft4.size
        s.arm_off, s.arm_len = self.arm_off.data_ptr(), self.arm_len.data_ptr()
        s.arms2, s.arms2_bytes = self.arms2.data_ptr(), b.arms2.size
        self.in_struct = s
        self.out_struct = abi.ConsensusBatch(self.bases.data_ptr(), self.off.data_ptr(), self.len.data_ptr(),
                                             self.status.data_ptr())

    def run(self, scores=abi.DEFAULT_SCORES, stream=None):
        """Enqueues the whole POA of the batch on `stream` (torch current stream by default); asynchronous."""
        import torch
        sp = abi.ScoreParams(*scores)
        st = stream if stream is not None else torch.cuda.current_stream(self.dev)
        self.gpu._check(self.gpu.lib.hypo_gpu_poa_batch_device(
            C.byref(sp), C.byref(self.in_struct), C.byref(self.out_struct),
            C.c_void_p(self.workspace.data_ptr()), C.c_size_t(self.workspace.numel()),
            C.c_void_p(st.cuda_stream)))

    def stats(self, stream=None) -> dict:
        import torch
        st = stream if stream is not None else torch.cuda.current_stream(self.dev)
        s = abi.PoaStats()
        self.gpu._check(self.gpu.lib.hypo_gpu_poa_read_stats(C.c_void_p(self.workspace.data_ptr()),
                                                             C.c_void_p(st.cuda_stream), C.byref(s)))
        d = _stats_dict(s)
        d["n_windows"] = self.host.n_windows
        return d

    def results(self):
        """(bases u8, off u64, len u32, status u8) copied back to the host."""
        import torch
        torch.cuda.synchronize(self.dev)
        n = self.host.n_windows
        return (self.bases.cpu().numpy(), self.off_host, self.len.cpu().numpy().view(np.uint32)[:n],
                self.status.cpu().numpy()[:n])

    def consensus(self):
        bases, off, ln, st = self.results()
        return [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes().decode() if st[i] == 0 else None
                for i in range(self.host.n_windows)], st


class DeviceScan:
    """A contig + solid-kmer set resident in HBM and the scan outputs."""

    def __init__(self, gpu: HypoGpu, packed4: np.ndarray, n_bases: int, k: int, bits: np.ndarray, kids_cap=None, misalign=0):
        import torch
        self.gpu, self.n_bases, self.k = gpu, n_bases, k
        dev = torch.device("cuda", gpu.device)
        self.dev = dev
        self.nw = (n_bases + 63) // 64
        self.kids_cap = n_bases if kids_cap is None else kids_cap
        self.packed4, self.bits = _t(packed4, dev), _t(bits, dev)
        if misalign:                 # the contig at an address that is not a multiple of 8 (a view into a larger buffer)
            self._backing = torch.zeros(self.packed4.numel() + misalign, dtype=torch.uint8, device=dev)
            self._backing[misalign:] = self.packed4
            self.packed4 = self._backing[misalign:]
        self.words = torch.zeros(max(self.nw, 1), dtype=torch.int64, device=dev)
        self.kids = torch.zeros(max(self.kids_cap, 1), dtype=torch.int64, device=dev)
        self.rank = torch.zeros(self.nw + 1, dtype=torch.int64, device=dev)
        self.n_solid = torch.zeros(1, dtype=torch.int64, device=dev)
        wsb = int(gpu.lib.hypo_gpu_solid_scan_workspace_bytes(C.c_uint64(n_bases)))
        self.workspace = torch.zeros(wsb, dtype=torch.uint8, device=dev)

    def run(self, stream=None):
        import torch
        st = stream if stream is not None else torch.cuda.current_stream(self.dev)
        self.gpu._check(self.gpu.lib.hypo_gpu_solid_scan_device(
            C.c_void_p(self.packed4.data_ptr()), C.c_uint64(self.n_bases), C.c_uint32(self.k),
            C.c_void_p(self.bits.data_ptr()), C.c_void_p(self.words.data_ptr()),
            C.c_void_p(self.kids.data_ptr()), C.c_uint64(self.kids_cap), C.c_void_p(self.rank.data_ptr()),
            C.c_void_p(self.n_solid.data_ptr()), C.c_void_p(self.workspace.data_ptr()),
            C.c_size_t(self.workspace.numel()), C.c_void_p(st.cuda_stream)))

    def results(self):
        import torch
        torch.cuda.synchronize(self.dev)
        n = int(self.n_solid.item())
        return (self.words.cpu().numpy().view(np.uint64)[:self.nw],
                self.kids.cpu().numpy().view(np.uint64)[:min(n, self.kids_cap)],
                self.rank.cpu().numpy().view(np.uint64), n)
