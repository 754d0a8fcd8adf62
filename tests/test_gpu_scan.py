"""GPU: solid-kmer scan kernel (through the C-ABI) against the oracle; bit-exact words, k-mer ids, ranks."""
import numpy as np
import pytest

from hypo_amd import capi, sim

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    return capi.HypoGpu(0)


@pytest.mark.parametrize("n,k,nfrac", [(200000, 11, 0.0), (200001, 11, 0.002), (50000, 5, 0.01), (300000, 13, 0.001),
                                       (16384, 11, 0.0), (16385, 7, 0.0), (70, 11, 0.0), (10, 11, 0.0), (1, 5, 0.0)])
def test_scan_vs_oracle(gpu, oracle_lib, n, k, nfrac):
    codes, p4 = sim.random_contig(n, seed=n + k, n_frac=nfrac)
    bits = sim.solid_bitset(codes, k, max_count=2 if k <= 7 else 1)
    w, kids, rank, ns = gpu.solid_scan(p4, n, k, bits)
    ow, okids, orank, ons = oracle_lib.solid_scan(p4, n, k, bits)
    assert ns == ons
    assert (w == ow).all()
    assert (kids == okids).all()
    assert (rank == orank).all()


def test_scan_homopolymers_and_device_api(gpu, oracle_lib):
    rng = np.random.default_rng(3)
    # runs of equal bases exercise both homopolymer-edge tests (Contig.cpp:59,63)
    codes = np.repeat(rng.integers(0, 4, size=60000, dtype=np.uint8), rng.integers(1, 4, size=60000))
    n = codes.size
    pad = np.concatenate([codes, np.zeros((-n) % 2, np.uint8)]).reshape(-1, 2)
    p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
    k = 9
    bits = np.full((1 << (2 * k)) // 64, np.uint64(0xFFFFFFFFFFFFFFFF))     # every k-mer solid
    ds = gpu.device_scan(p4, n, k, bits)
    ds.run()
    w, kids, rank, ns = ds.results()
    ow, okids, orank, ons = oracle_lib.solid_scan(p4, n, k, bits)
    assert ns == ons and (w == ow).all() and (kids == okids).all() and (rank == orank).all()
    # kids_cap smaller than the number of hits: count still exact, prefix written
    w2, kids2, _, ns2 = gpu.solid_scan(p4, n, k, bits, kids_cap=100)
    assert ns2 == ons and (kids2 == okids[:100]).all()


def test_scan_random_sizes_and_k(gpu, oracle_lib):
    """80 random (length, k, N density, set density) combinations incl. lengths around the 64-position word and 8 KiB block edges."""
    rng = np.random.default_rng(2026)
    edges = [1, 2, 63, 64, 65, 127, 128, 129, 16383, 16384, 16385, 16447, 32768, 32769]
    for it in range(80):
        k = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 11, 12, 13]))
        n = int(rng.choice(edges)) if it < 28 else int(rng.integers(1, 150000))
        nfrac = float(rng.choice([0.0, 0.0, 0.001, 0.05, 0.5]))
        codes = rng.integers(0, 4, size=n, dtype=np.uint8)
        if rng.random() < 0.3:                                   # homopolymer-rich
            codes = np.repeat(codes[: max(1, n // 3)], 3)[:n]
            n = codes.size
        codes[rng.random(n) < nfrac] = 4
        pad = np.concatenate([codes, np.zeros((-n) % 2, np.uint8)]).reshape(-1, 2)
        p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
        dens = float(rng.choice([0.02, 0.3, 1.0]))
        nwords = max(1, (1 << (2 * k)) // 64)
        bits = np.zeros(nwords, dtype=np.uint64)
        for sh in range(64):
            bits |= (rng.random(nwords) < dens).astype(np.uint64) << np.uint64(sh)
        if (1 << (2 * k)) < 64:
            bits &= np.uint64((1 << (1 << (2 * k))) - 1)
        w, kids, rank, ns = gpu.solid_scan(p4, n, k, bits)
        ow, okids, orank, ons = oracle_lib.solid_scan(p4, n, k, bits)
        assert ns == ons and (w == ow).all() and (kids == okids).all() and (rank == orank).all(), (it, n, k, nfrac, dens)


@pytest.mark.parametrize("mis", [1, 3, 4, 7])
def test_scan_contig_not_8_byte_aligned(gpu, oracle_lib, mis):
    """The k-mer id kernel reads the contig in aligned 8-byte words: a contig that starts anywhere inside such a word, lengths
    that end anywhere inside one."""
    for n, k in ((100003 + mis, 13), (4099, 16), (77, 9), (16, 16)):
        codes, p4 = sim.random_contig(n, seed=n + mis, n_frac=0.001)
        nwords = max(1, (1 << (2 * k)) // 64)
        rng = np.random.default_rng(n)
        bits = rng.integers(0, 1 << 63, size=nwords, dtype=np.int64).view(np.uint64) | np.uint64(1 << 63)      # ~ half of all k-mers
        ds = gpu.device_scan(p4, n, k, bits, misalign=mis)
        ds.run()
        w, kids, rank, ns = ds.results()
        ow, okids, orank, ons = oracle_lib.solid_scan(p4, n, k, bits)
        assert ns == ons and (w == ow).all() and (kids == okids).all() and (rank == orank).all(), (n, k, mis)


def test_scan_vs_reference_fixture(gpu):
    """G3: the device against the outputs of the REAL hypo::Contig::find_solid_pos (tests/golden/scan_cases.json.gz)."""
    from test_scan_golden import scan_cases
    for c, p4, bits, words, kids, rank in scan_cases():
        w, gk, gr, ns = gpu.solid_scan(p4, c["n"], c["k"], bits)
        assert ns == c["n_solid"], (c["n"], c["k"])
        assert (w == words[:w.size]).all() and (gk == kids).all() and (gr == rank).all(), (c["n"], c["k"])


def test_uploaded_solid_set_is_reused(gpu, oracle_lib):
    """hypo_gpu_solid_set_upload once, then scans with bitset_words == NULL (how the host pipeline scans many contigs)."""
    import ctypes as C
    k = 11
    codes0, _ = sim.random_contig(300000, seed=1)
    bits = sim.solid_bitset(codes0, k)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert gpu.lib.hypo_gpu_solid_set_upload(p(bits), C.c_uint32(k)) == 0
    for seed, n in [(1, 300000), (2, 12345)]:
        codes, p4 = sim.random_contig(n, seed=seed)
        nw = (n + 63) // 64
        words = np.zeros(nw, np.uint64); kids = np.zeros(n, np.uint64); rank = np.zeros(nw + 1, np.uint64); ns = C.c_uint64(0)
        rc = gpu.lib.hypo_gpu_solid_scan(p(p4), C.c_uint64(n), C.c_uint32(k), None, p(words), p(kids), C.c_uint64(n), p(rank), C.byref(ns))
        assert rc == 0
        ow, okids, orank, ons = oracle_lib.solid_scan(p4, n, k, bits)
        assert ns.value == ons and (words == ow).all() and (kids[:ons] == okids).all() and (rank == orank).all()
    # a different k without an upload is an error, not a stale set
    rc = gpu.lib.hypo_gpu_solid_scan(p(p4), C.c_uint64(n), C.c_uint32(9), None, p(words), p(kids), C.c_uint64(n), p(rank), C.byref(ns))
    assert rc == abi_invalid()


def abi_invalid():
    from hypo_amd import abi
    return abi.HYPO_E_INVALID
