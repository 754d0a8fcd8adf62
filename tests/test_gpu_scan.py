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
