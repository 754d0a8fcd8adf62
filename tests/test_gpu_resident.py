"""MI355X, straight through the C-ABI (ctypes): the round-4 entry points that keep data on the device — hypo_gpu_solid_scan_keep,
hypo_gpu_support_kmers_kept, hypo_gpu_solid_release, hypo_gpu_host_alloc / _free — against the entry points they shortcut
(hypo_gpu_solid_scan, hypo_gpu_support_kmers), which the oracle and the reference pin elsewhere."""
import ctypes as C

import numpy as np
import pytest

from hypo_amd import abi, capi, sim

pytestmark = pytest.mark.gpu


class ArmsReads(C.Structure):                      # HypoArmsReads, include/hypo_gpu.h
    _fields_ = [("n_alignments", C.c_uint32), ("rb", C.c_void_p), ("re", C.c_void_p), ("qae", C.c_void_p), ("seq_off", C.c_void_p),
                ("reads2", C.c_void_p), ("reads2_bytes", C.c_uint64), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("file_rank", C.c_void_p)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _reads(codes, n_reads, read_len, rng):
    """n_reads exact copies of stretches of the contig, sorted by start: (rb, re, qae, seq_off, reads2, cigar_off, cigar)."""
    n = codes.size
    rb = np.sort(rng.integers(0, n - read_len, size=n_reads)).astype(np.uint32)
    keep = np.array([bool((codes[b:b + read_len] < 4).all()) for b in rb])       # (a read over an N would not pack in 2 bits)
    rb = rb[keep]
    m = rb.size
    nb = (read_len + 3) // 4
    reads2 = np.zeros(m * nb, dtype=np.uint8)
    for i, b in enumerate(rb):
        c = np.concatenate([codes[b:b + read_len], np.zeros((-read_len) % 4, np.uint8)]).reshape(-1, 4)
        reads2[i * nb:(i + 1) * nb] = (c[:, 0] << 6) | (c[:, 1] << 4) | (c[:, 2] << 2) | c[:, 3]
    re = (rb + read_len).astype(np.uint32)
    qae = np.full(m, read_len, dtype=np.uint32)
    seq_off = (np.arange(m, dtype=np.uint64) * nb).astype(np.uint64)
    cigar_off = np.arange(m + 1, dtype=np.uint32)
    cigar = np.full(m, (read_len << 4) | 0, dtype=np.uint32)
    return rb, re, qae, seq_off, reads2, cigar_off, cigar


@pytest.mark.parametrize("n_bases,k", [(300_001, 11), (1_000_000, 13)])
def test_kept_scan_and_kept_votes_equal_the_plain_entry_points(n_bases, k):
    gpu = capi.HypoGpu(0)
    lib = gpu.lib
    rng = np.random.default_rng(n_bases + k)
    codes, p4 = sim.random_contig(n_bases, seed=n_bases, n_frac=0.0003)
    bits = sim.solid_bitset(codes, k)
    assert lib.hypo_gpu_solid_set_upload(_p(bits), C.c_uint32(k)) == abi.HYPO_OK
    words, kids, rank, ns = gpu.solid_scan(p4, n_bases, k, bits)
    nw = (n_bases + 63) // 64
    words2 = np.zeros(nw, dtype=np.uint64)
    rank2 = np.zeros(nw + 1, dtype=np.uint64)
    ns2 = C.c_uint64(0)
    handle = 5
    rc = lib.hypo_gpu_solid_scan_keep(C.c_uint32(handle), _p(p4), C.c_uint64(n_bases), C.c_uint32(k), _p(words2), _p(rank2), C.byref(ns2))
    assert rc == abi.HYPO_OK, lib.hypo_gpu_last_error()
    assert ns2.value == ns and (words2 == words).all() and (rank2 == rank).all()
    assert ns > 1000
    # the marked positions from the mark bits
    bitsarr = np.unpackbits(words.view(np.uint8), bitorder="little")[:n_bases]
    spos = np.nonzero(bitsarr)[0].astype(np.uint32)
    assert spos.size == ns
    # reads
    rb, re, qae, seq_off, reads2, cigar_off, cigar = _reads(codes, 6000, 150, rng)
    A = ArmsReads(rb.size, _p(rb), _p(re), _p(qae), _p(seq_off), _p(reads2), reads2.size, _p(cigar_off), _p(cigar), None)
    ctg = np.zeros(rb.size, dtype=np.uint32)
    total = n_bases + (n_bases & 1)
    assert lib.hypo_gpu_reads_upload(C.byref(A), _p(ctg), C.c_uint64(total)) == abi.HYPO_OK, lib.hypo_gpu_last_error()
    cov = np.zeros(ns, dtype=np.uint32); sup = np.zeros(ns, dtype=np.uint32)
    assert lib.hypo_gpu_support_kmers(C.c_uint32(k), C.c_uint64(ns), _p(spos), _p(kids), _p(cov), _p(sup)) == abi.HYPO_OK, lib.hypo_gpu_last_error()
    assert cov.sum() > 0 and sup.sum() > 0
    # the same votes against the kept scan, results into page-locked memory
    ptr = C.c_void_p()
    assert lib.hypo_gpu_host_alloc(C.c_size_t(ns * 8), C.byref(ptr)) == abi.HYPO_OK and ptr.value
    pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(2 * ns,))
    handles = np.array([handle], dtype=np.uint32); bases = np.array([0], dtype=np.uint32)
    tot = C.c_uint64(0)
    rc = lib.hypo_gpu_support_kmers_kept(C.c_uint32(k), C.c_uint32(1), _p(handles), _p(bases), C.cast(ptr, C.c_void_p),
                                         C.c_void_p(ptr.value + 4 * ns), C.byref(tot))
    assert rc == abi.HYPO_OK, lib.hypo_gpu_last_error()
    assert tot.value == ns
    assert (pinned[:ns] == cov).all() and (pinned[ns:] == sup).all()
    # a handle that holds nothing / a released one is an error, not a fault
    bad = np.array([77], dtype=np.uint32)
    assert lib.hypo_gpu_support_kmers_kept(C.c_uint32(k), C.c_uint32(1), _p(bad), _p(bases), _p(cov), _p(sup), C.byref(tot)) == abi.HYPO_E_INVALID
    assert lib.hypo_gpu_solid_release(C.c_uint32(handle)) == abi.HYPO_OK
    assert lib.hypo_gpu_support_kmers_kept(C.c_uint32(k), C.c_uint32(1), _p(handles), _p(bases), _p(cov), _p(sup), C.byref(tot)) == abi.HYPO_E_INVALID
    assert lib.hypo_gpu_host_free(ptr) == abi.HYPO_OK


def test_resident_tables_are_checked_against_the_resident_reads():
    """the minimizer tables of a caller other than DeviceArms are validated against the reads hypo_gpu_reads_upload left on
    the device (contig count, spans, MWMinimiserInfo range) instead of being followed out of bounds."""
    gpu = capi.HypoGpu(0)
    lib = gpu.lib
    rng = np.random.default_rng(9)
    codes, _ = sim.random_contig(50_000, seed=3, n_frac=0.0)
    rb, re, qae, seq_off, reads2, cigar_off, cigar = _reads(codes, 500, 150, rng)
    A = ArmsReads(rb.size, _p(rb), _p(re), _p(qae), _p(seq_off), _p(reads2), reads2.size, _p(cigar_off), _p(cigar), None)
    ctg = np.zeros(rb.size, dtype=np.uint32)
    assert lib.hypo_gpu_reads_upload(C.byref(A), _p(ctg), C.c_uint64(50_000)) == abi.HYPO_OK

    class Mega(C.Structure):                       # HypoMegaWindows
        _fields_ = [("n_contigs", C.c_uint32), ("contig_base", C.c_void_p), ("reg_base", C.c_void_p), ("win_even", C.c_void_p), ("info_base", C.c_void_p),
                    ("start", C.c_void_p), ("n_info", C.c_uint32), ("mw_off", C.c_void_p), ("rel_pos", C.c_void_p), ("minimisers", C.c_void_p)]
    cb = np.array([0], dtype=np.uint32); rbase = np.array([0, 2], dtype=np.uint32); even = np.array([1], dtype=np.uint8); ib = np.array([0], dtype=np.uint32)
    start = np.array([0, 20_000], dtype=np.uint32)            # the table says the contig ends at 20 000: the reads go on to 50 000
    mw_off = np.array([0, 1], dtype=np.uint32); rel = np.array([5], dtype=np.uint32); mins = np.array([7], dtype=np.uint32)
    W = Mega(1, _p(cb), _p(rbase), _p(even), _p(ib), _p(start), 1, _p(mw_off), _p(rel), _p(mins))
    cov = np.zeros(1, dtype=np.uint32); sup = np.zeros(1, dtype=np.uint32)
    assert lib.hypo_gpu_support_minimizers(C.byref(W), _p(cov), _p(sup)) == abi.HYPO_E_INVALID
    assert b"resident reads span" in lib.hypo_gpu_last_error()
