"""The build's own synthetic workload generator: window-level batches with the shape
the reference produces on the C1/C2 configuration, contigs + solid-kmer bit sets for the scan.
Deterministic (numpy Generator with fixed seeds), vectorised, produces packed HostBatch buffers directly.
"""
import os

import numpy as np

from . import abi
from .batch import HostBatch

_SHAPES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shapes")


def _pack_codes(codes: np.ndarray, lens: np.ndarray, bases_per_byte: int):
    """Packs concatenated base codes (per-sequence lengths `lens`) PackedSeq-style, each sequence
    starting on a byte boundary.  Returns (bytes, byte_off u64, lens u32)."""
    lens = lens.astype(np.int64)
    n = lens.size
    nbytes = (lens + bases_per_byte - 1) // bases_per_byte
    byte_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nbytes, out=byte_off[1:])
    starts = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    pos = np.arange(codes.size, dtype=np.int64) - np.repeat(starts[:-1], lens)
    dest = np.repeat(byte_off[:-1], lens) * bases_per_byte + pos
    slots = np.zeros(int(byte_off[-1]) * bases_per_byte, dtype=np.uint8)
    slots[dest] = codes
    c = slots.reshape(-1, bases_per_byte)
    if bases_per_byte == 4:
        out = (c[:, 0] << 6) | (c[:, 1] << 4) | (c[:, 2] << 2) | c[:, 3]
    else:
        out = (c[:, 0] << 4) | c[:, 1]
    return out.astype(np.uint8), byte_off[:-1].astype(np.uint64), lens.astype(np.uint32)


def _segments(src_start: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Index array that concatenates ranges [src_start[i], src_start[i]+lens[i])."""
    lens = lens.astype(np.int64)
    cum = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=cum[1:])
    return np.repeat(src_start.astype(np.int64) - cum[:-1], lens) + np.arange(int(cum[-1]), dtype=np.int64)


def load_shape(name: str = "c1_shape"):
    z = np.load(os.path.join(_SHAPES, name + ".npz"))
    return z["shape"].astype(np.int64), z["count"].astype(np.float64)


def window_batch(n_windows: int, seed: int = 1, shape: str = "c1_shape", read_sub: float = 0.002,
                 draft_err: float = 0.012, shapes=None, long_mask=None, long_err: float = 0.10) -> HostBatch:
    """Synthetic SHORT-window batch.  Each window: random truth; draft = truth with `draft_err`
    (1/3 substitutions, 1/3 deletions, 1/3 insertions); internal arms = truth with `read_sub`
    substitutions; prefix arms = truth prefixes of increasing length, suffix arms = truth suffixes of
    decreasing length (BAM order, cf. Window.cpp:110).  `shapes` (array [n,5] of
    (window_len, n_internal, n_prefix, n_suffix, n_empty)) overrides sampling from the shape table.
    `long_mask` (bool [n]): these windows are LONG windows (Window.cpp:156-254) whose arms are noisy long-read
    segments: `long_err` errors per base, a third each substitutions, deletions and insertions."""
    rng = np.random.default_rng(seed)
    if shapes is None:
        tab, cnt = load_shape(shape)
        pick = rng.choice(tab.shape[0], size=n_windows, p=cnt / cnt.sum())
        shapes = tab[pick]
    shapes = np.asarray(shapes, dtype=np.int64)
    n = shapes.shape[0]
    wl, ni, npre, nsuf, nem = (shapes[:, i] for i in range(5))
    # truth per window (its length is the window length; the draft gets the indels)
    toff = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(wl, out=toff[1:])
    truth = rng.integers(0, 4, size=int(toff[-1]), dtype=np.uint8)
    # draft: per truth base keep/substitute/delete, plus inserted bases
    r = rng.random(truth.size)
    keep = r >= draft_err / 3
    sub = (r >= draft_err / 3) & (r < 2 * draft_err / 3)
    ins = rng.random(truth.size) < draft_err / 3
    dcodes = truth.copy()
    dcodes[sub] = (dcodes[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) % 4
    rep = keep.astype(np.int64) + ins.astype(np.int64)
    owner = np.repeat(np.arange(n), wl)
    dlen = np.bincount(owner, weights=rep, minlength=n).astype(np.int64)
    empty = dlen == 0                                    # never produce an empty draft
    if empty.any():
        first = toff[:-1][empty]
        rep[first] = 1
        dlen[empty] = 1
    draft_codes = np.repeat(dcodes, rep)
    # inserted copies get a random base: the second copy of a (keep & ins) base
    dup_second = np.zeros(draft_codes.size, dtype=bool)
    cum = np.cumsum(rep)
    two = rep == 2
    dup_second[cum[two] - 1] = True
    draft_codes[dup_second] = rng.integers(0, 4, size=int(dup_second.sum()), dtype=np.uint8)
    draft4, doff, dlen32 = _pack_codes(draft_codes, dlen, 2)
    # arms: internal | prefix | suffix per window
    narm = ni + npre + nsuf
    first_arm = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(narm, out=first_arm[1:])
    A = int(first_arm[-1])
    arm_win = np.repeat(np.arange(n), narm)
    idx_in_win = np.arange(A, dtype=np.int64) - np.repeat(first_arm[:-1], narm)
    kind = np.where(idx_in_win < ni[arm_win], 0, np.where(idx_in_win < (ni + npre)[arm_win], 1, 2))
    w_len = wl[arm_win]
    # prefix arm j of npre: length grows with j; suffix arm j of nsuf: length shrinks with j
    jpre = idx_in_win - ni[arm_win]
    jsuf = idx_in_win - (ni + npre)[arm_win]
    fpre = 0.1 + 0.9 * (jpre + rng.random(A)) / np.maximum(npre[arm_win], 1)
    fsuf = 1.0 - 0.9 * (jsuf + rng.random(A)) / np.maximum(nsuf[arm_win], 1)
    alen = np.where(kind == 0, w_len,
                    np.where(kind == 1, np.clip(np.ceil(w_len * fpre), 1, w_len),
                             np.clip(np.ceil(w_len * fsuf), 1, w_len))).astype(np.int64)
    astart = np.where(kind == 2, w_len - alen, 0) + toff[:-1][arm_win]
    src = _segments(astart, alen)
    acodes = truth[src]
    e = rng.random(acodes.size) < read_sub
    acodes[e] = (acodes[e] + rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)) % 4
    if long_mask is not None and np.any(long_mask):
        # indels in the arms of the LONG windows: per base keep / delete, plus an inserted random base behind it
        is_long_base = np.repeat(np.asarray(long_mask, dtype=bool)[arm_win], alen)
        r2 = rng.random(acodes.size)
        sub2 = is_long_base & (r2 < long_err / 3)
        dele = is_long_base & (r2 >= long_err / 3) & (r2 < 2 * long_err / 3)
        ins2 = is_long_base & (rng.random(acodes.size) < long_err / 3)
        acodes[sub2] = (acodes[sub2] + rng.integers(1, 4, size=int(sub2.sum()), dtype=np.uint8)) % 4
        rep2 = (~dele).astype(np.int64) + ins2.astype(np.int64)
        arm_of_base = np.repeat(np.arange(A), alen)
        new_len = np.bincount(arm_of_base, weights=rep2, minlength=A).astype(np.int64)
        gone = (new_len == 0) & (alen > 0)                 # never delete an arm completely
        if gone.any():
            astart0 = np.zeros(A + 1, dtype=np.int64)
            np.cumsum(alen, out=astart0[1:])
            rep2[astart0[:-1][gone]] = 1
            new_len[gone] = 1
        out = np.repeat(acodes, rep2)
        cum2 = np.cumsum(rep2)
        second = np.zeros(out.size, dtype=bool)
        second[cum2[rep2 == 2] - 1] = True
        out[second] = rng.integers(0, 4, size=int(second.sum()), dtype=np.uint8)
        acodes, alen = out, new_len
    arms2, aoff, alen32 = _pack_codes(acodes, alen, 4)
    wd = np.zeros(n, dtype=abi.WINDOW_DTYPE)
    wd["type"] = abi.WIN_SHORT
    if long_mask is not None:
        wd["type"][np.asarray(long_mask, dtype=bool)] = abi.WIN_LONG
    wd["draft_len"] = dlen32
    wd["draft_off"] = doff
    wd["first_arm"] = first_arm[:-1]
    wd["n_internal"], wd["n_prefix"], wd["n_suffix"], wd["n_empty"] = ni, npre, nsuf, nem
    return HostBatch(wd, draft4, aoff, alen32, arms2)


def grid_batch(length: int, arms: int, n_windows: int, arm_err: float, seed: int = 1) -> HostBatch:
    """One cell of the window-level grid of SURVEY.md §8(d): fixed length/arm count, 60/20/20 mix."""
    rng = np.random.default_rng(seed)
    ni = rng.binomial(arms, 0.6, size=n_windows)
    npre = rng.binomial(arms - ni, 0.5)
    nsuf = arms - ni - npre
    shapes = np.stack([np.full(n_windows, length), ni, npre, nsuf, np.zeros(n_windows, np.int64)], axis=1)
    return window_batch(n_windows, seed=seed + 1, shapes=shapes, read_sub=arm_err)


# ---- contigs + solid-kmer sets for the scan ------------------------------------------------------------
def random_contig(n_bases: int, seed: int = 1, n_frac: float = 0.0):
    """Returns (codes u8 [n], packed4 bytes).  codes: A0 C1 G2 T3 N4."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    if n_frac > 0:
        codes[rng.random(n_bases) < n_frac] = 4
    pad = np.concatenate([codes, np.zeros((-n_bases) % 2, np.uint8)]).reshape(-1, 2)
    return codes, ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)


def solid_bitset(codes: np.ndarray, k: int, max_count: int = 1) -> np.ndarray:
    """4^k-bit set (little-endian u64 words) of canonical k-mers occurring <= max_count times in `codes`
    (both strands counted), without a terminal homopolymer, both strands set — mimics
    external/suk/src/SolidKmers.cpp:166-189."""
    n = codes.size
    nwords = max((1 << (2 * k)) // 64, 1)
    words = np.zeros(nwords, dtype=np.uint64)
    if n < k:
        return words
    c = codes.astype(np.int64)
    valid = np.ones(n - k + 1, dtype=bool)
    fwd = np.zeros(n - k + 1, dtype=np.int64)
    rc = np.zeros(n - k + 1, dtype=np.int64)
    for t in range(k):
        b = c[t:n - k + 1 + t]
        valid &= b < 4
        fwd = (fwd << 2) | (b & 3)
        rc |= (3 - (b & 3)) << (2 * t)
    fwd, rc = fwd[valid], rc[valid]
    canon = np.minimum(fwd, rc)
    uniq, cnt = np.unique(canon, return_counts=True)
    sel = uniq[cnt <= max_count]
    # no homopolymer at either end (first two / last two bases equal)
    first2 = (sel >> (2 * (k - 1))) & 3 == (sel >> (2 * (k - 2))) & 3
    last2 = (sel & 3) == ((sel >> 2) & 3)
    sel = sel[~(first2 | last2)]
    # reverse complement of the selected canonical k-mers
    r = np.zeros_like(sel)
    x = sel.copy()
    for t in range(k):
        r = (r << 2) | (3 - (x & 3))
        x >>= 2
    ids = np.concatenate([sel, r]).astype(np.uint64)
    np.bitwise_or.at(words, (ids >> np.uint64(6)).astype(np.int64), np.uint64(1) << (ids & np.uint64(63)))
    return words


def c4_batch(n_short: int, n_long: int, seed: int = 1, long_err: float = 0.10, long_len=(120, 500), long_arms=(12, 45)) -> HostBatch:
    """The window mix of BASELINE config C4 (short reads + noisy long reads, `-B`): C1-shaped SHORT windows and LONG windows
    (120-500 bp, 12-45 long-read arms of about the window's length with `long_err` errors incl. indels, up to two empty arms),
    shuffled into one batch as Hypo::polish hands them over."""
    rng = np.random.default_rng(seed)
    tab, cnt = load_shape("c1_shape")
    sh = tab[rng.choice(tab.shape[0], size=n_short, p=cnt / cnt.sum())]
    zeros = np.zeros(n_long, np.int64)
    lg = np.stack([rng.integers(long_len[0], long_len[1], size=n_long), rng.integers(long_arms[0], long_arms[1], size=n_long), zeros, zeros, rng.integers(0, 3, size=n_long)], axis=1)
    shapes = np.concatenate([sh, lg])
    mask = np.concatenate([np.zeros(n_short, bool), np.ones(n_long, bool)])
    perm = rng.permutation(shapes.shape[0])
    return window_batch(0, seed=seed + 1, shapes=shapes[perm], long_mask=mask[perm], long_err=long_err)
