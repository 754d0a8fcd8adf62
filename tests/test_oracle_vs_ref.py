"""CPU, build container only: randomized comparison of oracle/ with the real reference (oracle/_ref).
Skipped where /root/reference (hence oracle/_ref) does not exist."""
import random

from hypo_amd.batch import TextWindow, build_batch


def _mut(rng, s, e):
    out = []
    for c in s:
        x = rng.random()
        if x < e / 3:
            continue
        out.append(rng.choice("ACGT") if x < 2 * e / 3 else c)
        if rng.random() < e / 3:
            out.append(rng.choice("ACGT"))
    return "".join(out) or "A"


def test_random_short_windows(oracle_lib, ref_lib):
    rng = random.Random(99)
    wins = []
    for _ in range(600):
        L = rng.choice([4, 8, 16, 32, 64, 100, 150])
        truth = "".join(rng.choice("ACGT") for _ in range(L))
        w = TextWindow(_mut(rng, truth, 0.05))
        for _ in range(rng.choice([2, 3, 6, 12, 30])):
            s = _mut(rng, truth, rng.choice([0.005, 0.08, 0.2]))
            k = rng.random()
            if k < 0.5:
                w.internal.append(s)
            elif k < 0.75:
                w.prefix.append(s[:rng.randint(1, len(s))])
            else:
                w.suffix.append(s[rng.randint(0, len(s) - 1):])
        wins.append(w)
    for scores in [(5, -4, -8, 3, -5, -4), (2, -3, -1, 3, -5, -4)]:
        cons, st, _, _ = oracle_lib.poa_batch(build_batch(wins), scores=scores)
        for w, c, s in zip(wins, cons, st):
            assert s == 0
            assert ref_lib.window(w, scores)[0] == c


def test_random_long_windows(oracle_lib, ref_lib):
    rng = random.Random(5)
    wins = []
    for _ in range(12):
        L = rng.choice([200, 300, 500])
        truth = "".join(rng.choice("ACGT") for _ in range(L))
        w = TextWindow(_mut(rng, truth, 0.03), is_long=True)
        for _ in range(rng.choice([3, 10, 25])):
            w.internal.append(_mut(rng, truth, 0.1))
        _, kept = ref_lib.window(w)
        w.internal = [s for s, k in zip(w.internal, kept) if k]
        wins.append(w)
    cons, st, _, _ = oracle_lib.poa_batch(build_batch(wins))
    for w, c, s in zip(wins, cons, st):
        assert s == 0
        assert ref_lib.window(w)[0] == c


def test_scan_vs_real_contig_find_solid_pos(oracle_lib):
    """oracle_solid_scan against hypo::Contig::find_solid_pos itself (oracle/_ref/libhyporef_scan.so), incl. a contig that
    spans several of the oracle's OpenMP chunks."""
    import numpy as np
    import pytest
    import oracle
    from hypo_amd import sim
    if not oracle.RefScan.available():
        pytest.skip("oracle/_ref/libhyporef_scan.so not built (the real reference only exists in the build container)")
    ref = oracle.RefScan()
    rng = np.random.default_rng(11)
    for n, k, nfrac in [(1, 2, 0.0), (9, 5, 0.0), (777, 4, 0.05), (70000, 9, 0.002), (1_300_001, 10, 0.0005), (250000, 11, 0.0)]:
        codes, p4 = sim.random_contig(n, seed=n + k, n_frac=nfrac)
        for _ in range(n // 150):
            s = int(rng.integers(0, max(n - 8, 1)))
            codes[s:s + int(rng.integers(2, 8))] = rng.integers(0, 4)
        pad = np.concatenate([codes, np.zeros((-n) % 2, np.uint8)]).reshape(-1, 2)
        p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
        bits = sim.solid_bitset(codes, k, max_count=25 if k < 11 else 1)
        text = np.frombuffer(b"ACGTN", dtype=np.uint8)[codes].tobytes()
        rw, rk, rr, rn = ref.solid_scan(text, k, bits)
        ow, ok, orank, on = oracle_lib.solid_scan(p4, n, k, bits)
        assert rn == on and (rw == ow).all() and (rk == ok).all() and (rr == orank).all(), (n, k)


def test_host_stage_vs_real_alignment_and_contig_code(tmp_path):
    """Support votes, division into regions and short-arm selection of this repo's host (over the CPU shim) against the
    reference's own Alignment.cpp / Contig.cpp / Window.cpp compiled in place (oracle/_ref/libhyporef_arms.so) on sets
    generated now: 3 clean seeds (k = 7 / 9 / 11) and the single-contig short-read ones of 40 messy seeds."""
    import pytest
    import oracle
    import e2e_util
    if not oracle.RefArms.available():
        pytest.skip("oracle/_ref/libhyporef_arms.so not built (the real reference only exists in the build container)")
    e2e_util.build_binary()
    e2e_util.build_shim()
    total = 0
    for seed in (301, 302, 303):
        total += e2e_util.run_vs_reference_stage(tmp_path / f"c{seed}", seed, "shim", messy=False)
    done = 0
    for seed in range(400, 440):
        n = e2e_util.run_vs_reference_stage(tmp_path / f"m{seed}", seed, "shim", messy=True)
        if n is not None:
            total += n
            done += 1
    assert done >= 4 and total > 500, (done, total)


def test_host_stage_vs_real_code_at_k13(tmp_path):
    """The same stage check at k = 13 (`-s 100m`, BASELINE config C3's k): nearly every k-mer of a 20 kbp genome is solid, so 40 %
    of the positions are marked — the regime of the large sets.  (k = 15 / 17 run on the GPU box: tests/test_gpu_e2e.py.)"""
    import pytest
    import oracle
    import e2e_util
    if not oracle.RefArms.available():
        pytest.skip("oracle/_ref/libhyporef_arms.so not built (the real reference only exists in the build container)")
    e2e_util.build_binary()
    e2e_util.build_shim()
    assert e2e_util.run_vs_reference_stage(tmp_path / "k13", 305, "shim", messy=False, k=13) > 20


def test_host_long_read_stage_vs_real_filter_and_alignment_code(tmp_path):
    """`-B` sets generated now: long-read arm selection (Alignment::find_long_arms, src/Alignment.cpp:262-299) and the minimizer
    filter of LONG windows (include/Filter.hpp through the real Window::add_*) of this repo's host mirror (over the CPU shim)
    against the reference's own code compiled in place (hyporef_arms_long in oracle/_ref/libhyporef_arms.so): region borders and
    types incl. the LONG windows, arm counts and crc32(arms) of every window.  The in-place build has no htslib, so its long reads
    go through the short-read constructor: only sets none of whose long reads fails the reference's NM filter take part."""
    import pytest
    import oracle
    import e2e_util
    if not oracle.RefArms.available():
        pytest.skip("oracle/_ref/libhyporef_arms.so not built (the real reference only exists in the build container)")
    e2e_util.build_binary()
    e2e_util.build_shim()
    total, done, n_long = 0, 0, 0
    for seed in (311, 312, 313):
        n = e2e_util.run_vs_reference_stage(tmp_path / f"c{seed}", seed, "shim", messy=False, long_reads=True)
        assert n is not None
        total += n
        n_long += e2e_util.LAST_LONG_WINDOWS
        done += 1
    for seed in range(400, 440):
        n = e2e_util.run_vs_reference_stage(tmp_path / f"m{seed}", seed, "shim", messy=True, long_reads=True)
        if n is not None:
            total += n
            n_long += e2e_util.LAST_LONG_WINDOWS
            done += 1
    assert done >= 4 and total > 300 and n_long >= 10, (done, total, n_long)


def test_files_through_the_reference_in_place(tmp_path):
    """Rows N3 and T1 on the CPU: whole alignment FILES, not parsed records, go to the reference's code (hyporef_fasta_sam / _bam: the
    harness's independent minimal SAM / BGZF + BAM decoder lays every record out as bam1_t, the reference's own constructor extracts
    the fields) and the FASTA records must be the bytes this repo's `hypo` (own readers, host/SeqIO.hpp, over the CPU shim) wrote: the
    messy seeds without long reads as SAM text and as BAM with tags, and a 6 x 100 kbp set of T1's generator with three picked contigs."""
    import pytest
    import oracle
    import e2e_util
    if not oracle.RefArms.available():
        pytest.skip("oracle/_ref/libhyporef_arms.so not built (the real reference only exists in the build container)")
    e2e_util.build_binary()
    e2e_util.build_shim()
    done, kept, multi = 0, 0, 0
    for seed in range(400, 424):
        for as_bam in (False, True):
            n = e2e_util.run_messy_files_vs_reference(tmp_path / f"m{seed}_{int(as_bam)}", seed, "shim", as_bam)
            if n is not None:
                done += 1
                kept += n
    assert done >= 12 and kept > 5000, (done, kept)
    dt, rr, same = e2e_util.run_t1_slice_vs_reference(tmp_path / "t1", 6, 100000, 11, "1m", [1, 3, 4], 4, device="shim", p=2)
    assert same == 3 and rr["contigs"] == 3 and rr["windows"] > 3000, (same, rr)
