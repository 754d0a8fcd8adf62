"""CPU: properties of the scan restatement (oracle_solid_scan).  The reference's Contig class cannot be
built here (sdsl needs cmake-generated sources), so the scan oracle is pinned by definition-level
properties and a brute-force numpy recomputation rather than by a reference binary."""
import numpy as np
import pytest

from hypo_amd import sim


def brute(codes, k, bits):
    n = codes.size
    out = []
    for beg in range(0, n - k + 1):
        seg = codes[beg:beg + k]
        if (seg > 3).any():
            continue
        kid = 0
        for b in seg:
            kid = (kid << 2) | int(b)
        if not (int(bits[kid >> 6]) >> (kid & 63)) & 1:
            continue
        i = beg + k - 1
        if i < n - 1 and codes[i + 1] == codes[i]:
            continue
        if beg > 0 and codes[beg - 1] == codes[beg]:
            continue
        out.append((beg, kid))
    return out


@pytest.mark.parametrize("n,k", [(3000, 5), (2999, 7), (64, 5), (65, 5), (4, 5)])
def test_scan_matches_bruteforce(oracle_lib, n, k):
    codes, p4 = sim.random_contig(n, seed=n, n_frac=0.01)
    bits = sim.solid_bitset(codes, k, max_count=3)
    words, kids, rank, ns = oracle_lib.solid_scan(p4, n, k, bits)
    want = brute(codes, k, bits)
    assert ns == len(want)
    got_pos = [i for i in range(n) if (int(words[i >> 6]) >> (i & 63)) & 1]
    assert got_pos == [p for p, _ in want]
    assert kids.tolist() == [kid for _, kid in want]
    assert int(rank[-1]) == ns
