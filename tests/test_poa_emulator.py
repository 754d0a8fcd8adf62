"""CPU: the kernel's per-window code (hypo_amd/csrc/poa_core.hpp) compiled for the lockstep emulator
(tests/emu/) against the committed goldens of the real reference.  This is the same source hipcc
compiles for gfx950; it catches logic and out-of-slice bugs before a GPU run."""
import pytest

from hypo_amd.batch import build_batch
import emu_util
import golden_util as gu


@pytest.fixture(scope="module")
def emu():
    return emu_util.Emu()


def _check(emu, cfg, name, max_items, allow_overflow):
    n_ok = 0
    for scores, items in gu.windows_by_scores(name).items():
        items = [it for it in items if not it[0].is_long][:max_items]
        if not items:
            continue
        b = build_batch([w for w, _, _ in items])
        cons, st, res, _, _ = emu.poa_batch(b, cfg, scores)
        for (w, want, tag), got, r, s in zip(items, cons, res, st):
            if r == emu_util.RES_OVERFLOW and allow_overflow:
                continue
            assert r == emu_util.RES_OK and s == 0, (tag, r, s)
            assert got == want, (tag, scores)
            n_ok += 1
    return n_ok


def test_class0_synth(emu):
    assert _check(emu, 0, "windows_synth.jsonl.gz", 120, True) > 50


def test_class0_two_groups_per_wave(emu):
    # the 32-lane x 2-column geometry of class 0 (PoaClass0W, emulator id 6)
    assert _check(emu, 6, "windows_synth.jsonl.gz", 120, True) > 50


def test_class1_synth(emu):
    assert _check(emu, 1, "windows_synth.jsonl.gz", 120, True) > 100


def test_class2_real(emu):
    assert _check(emu, 2, "windows_real_c1.jsonl.gz", 150, False) == 150


def test_class3_bytes_dir(emu):
    # 8-bit direction codes (in-degree capacity 8)
    assert _check(emu, 3, "windows_synth.jsonl.gz", 60, True) > 50


def test_class4_wide(emu):
    # the 8-columns-per-lane / 16-bit-id / int32 instantiation, including the 200-bp windows
    items = [it for it in gu.windows_by_scores("windows_synth.jsonl.gz")[(5, -4, -8, 3, -5, -4)]
             if not it[0].is_long and len(it[0].draft) >= 150][:12]
    b = build_batch([w for w, _, _ in items])
    cons, st, res, _, _ = emu.poa_batch(b, 4)
    for (w, want, tag), got, r in zip(items, cons, res):
        assert r == emu_util.RES_OK and got == want, tag


def test_class4_long_windows(emu):
    """LONG windows: all-kNW, no markers, generate_consensus_custom + curate, two rounds (Window.cpp:156-254)."""
    n = 0
    for name in ("windows_synth.jsonl.gz", "windows_real_long.jsonl.gz"):
        items = [it for it in gu.windows_by_scores(name)[(5, -4, -8, 3, -5, -4)] if it[0].is_long][:8]
        b = build_batch([w for w, _, _ in items])
        cons, st, res, _, _ = emu.poa_batch(b, 4)
        for (w, want, tag), got, r in zip(items, cons, res):
            assert r == emu_util.RES_OK and got == want, tag
            n += 1
    assert n >= 12


@pytest.mark.parametrize("exact", [True, False])
def test_random_batches_vs_oracle(exact):
    """Seeded C1-shaped batches (0.2 - 3 % read error) and prefix/suffix-heavy grid windows through every LDS size class, with
    exact threading (Poa::rows_exact) and with every alignment forced through the score rows (HYPO_EXACT=0)."""
    import oracle
    from hypo_amd import sim
    emu = emu_util.Emu(exact=exact)
    orc = oracle.Oracle()
    n_ran = 0
    cases = [(2, sim.window_batch(160, seed=5)), (1, sim.window_batch(200, seed=6)), (0, sim.window_batch(250, seed=7)),
             (6, sim.window_batch(250, seed=8)), (3, sim.window_batch(80, seed=9)),
             (2, sim.window_batch(120, seed=10, read_sub=0.02)), (1, sim.window_batch(160, seed=11, read_sub=0.03)),
             (0, sim.grid_batch(30, 24, 60, 0.004, seed=30)), (1, sim.grid_batch(60, 24, 60, 0.004, seed=60)),
             (2, sim.grid_batch(100, 24, 50, 0.004, seed=100)), (3, sim.grid_batch(180, 24, 30, 0.004, seed=180))]
    for cfg, b in cases:
        cons, st, res, _, _ = emu.poa_batch(b, cfg)
        want = orc.poa_batch(b)[0]
        for i in range(b.n_windows):
            if res[i] == emu_util.RES_OK:
                assert cons[i] == want[i], (cfg, i)
                n_ran += 1
    assert n_ran > 1000


def test_class4_long_windows_lazy_rank_order_vs_oracle():
    """LONG windows in the hybrid class keep a valid rank order incrementally and sort literally only before a round's consensus
    and when a kNW end row is tied (Poa::lazy_update, HYPO_LAZY_TOPO): fuzzed LONG windows of every flavour (internal / prefix /
    suffix arms, 0-15 % errors, empty arms) under two score sets and the noisy-long-read windows of the C4 mix, against the
    oracle."""
    import numpy as np
    import oracle
    from hypo_amd import sim
    from test_gpu_fuzz import _window
    orc = oracle.Oracle()
    rng = np.random.default_rng(77)
    cases = [(build_batch([_window(rng, True) for _ in range(60)]), (5, -4, -8, 3, -5, -4)),
             (build_batch([_window(rng, True) for _ in range(30)]), (5, -4, -8, 1, -1, -1)),
             (sim.c4_batch(0, 12, seed=5), (5, -4, -8, 3, -5, -4))]
    n = 0
    e = emu_util.Emu()
    for b, scores in cases:
        cons, st, res, _, _ = e.poa_batch(b, 4, scores)
        want = orc.poa_batch(b, scores=scores)[0]
        for i in range(b.n_windows):
            assert res[i] == emu_util.RES_OK and cons[i] == want[i], i
            n += 1
    assert n >= 100


def test_class4_rows_as_wide_as_the_sequence_short_windows_vs_oracle():
    """The hybrid class's row loop carries 2-5 register pairs per lane, whichever holds the sequence in hand (Poa::rows_pk_hyb_w):
    SHORT fuzz windows pushed straight into class 4 (prefix / suffix arms of 20-190 bases: every split within one window, kLOV /
    kROV end rows and their ties), three score sets, against the oracle.  The first case is the window of
    tests/sweep_parity_gpu.py's round 424242 that differed on the GPU while the tie count of an alignment was read from the lane
    that owns the last column under the class's ten columns per lane (round 3)."""
    import numpy as np
    import oracle
    from test_gpu_fuzz import _window
    orc = oracle.Oracle()
    e = emu_util.Emu()
    rng = np.random.default_rng(7000 + 424242)
    skipped = [_window(rng, False) for _ in range(6000)] + [_window(rng, True) for _ in range(150)]
    the_one = [_window(rng, False) for _ in range(2318)][2317:]
    del skipped
    rng = np.random.default_rng(4242)
    cases = [(build_batch(the_one), (3, -6, -5, 3, -5, -4))]
    for scores in ((5, -4, -8, 3, -5, -4), (3, -6, -5, 3, -5, -4), (1, -1, -1, 1, -1, -1)):
        cases.append((build_batch([_window(rng, False) for _ in range(120)]), scores))
    n = 0
    for b, scores in cases:
        cons, st, res, _, _ = e.poa_batch(b, 4, scores)
        want = orc.poa_batch(b, scores=scores)[0]
        for i in range(b.n_windows):
            if res[i] == emu_util.RES_OK:
                assert cons[i] == want[i], (scores, i)
                n += 1
    assert n >= 300


def _requeue_cases(small):
    from hypo_amd import sim
    n = 40 if small else 200
    return [(0, sim.window_batch(n, seed=21, read_sub=0.03)), (0, sim.window_batch(n, seed=22, read_sub=0.06)),
            (1, sim.window_batch(n, seed=23, read_sub=0.05)), (2, sim.window_batch(n * 3 // 4, seed=24, read_sub=0.05)),
            (0, sim.grid_batch(45, 40, 12 if small else 40, 0.08, seed=8)), (2, sim.grid_batch(120, 60, 6 if small else 20, 0.06, seed=9))]


def _requeue_check(asan):
    import oracle
    orc = oracle.Oracle()
    e = emu_util.Emu(asan=asan)
    n_carried = 0
    for cfg, b in _requeue_cases(asan):
        want = orc.poa_batch(b)[0]
        cons, res, hops, carried = e.poa_chain(b, cfg)
        for i in range(b.n_windows):
            assert res[i] == emu_util.RES_OK and cons[i] == want[i], (cfg, i, int(hops[i]), int(carried[i]))
        n_carried += int((carried > 0).sum())
        if not asan:
            cons0, res0, _, carried0 = e.poa_chain(b, cfg, use_carry=False)
            assert int(carried0.sum()) == 0 and cons0 == cons
    assert n_carried > (20 if asan else 100)
    return n_carried


def test_requeue_chain_carries_the_graph():
    """A SHORT window that outgrows its class is re-queued to class 3 together with the graph of the sequences added so far
    (Poa::spill / Poa::restore, poa_class_kernel's account()): the kernel's chain on the CPU, noisy batches started in every LDS
    class, against the oracle — with the spill and, for comparison, with every window starting over as before round 3."""
    _requeue_check(False)


def test_requeue_chain_under_asan():
    """The same chain (smaller batches) with the AddressSanitizer build of the emulator: spill and restore stay inside the
    group's slice and the spill buffer (run in a child process: the sanitizer runtime has to be loaded first)."""
    import os
    import subprocess
    import sys
    emu_util.build()
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path[:0] = [%r, %r]; import test_poa_emulator as t; print('carried', t._requeue_check(True))" % (os.path.dirname(here), here)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "carried" in p.stdout, (p.stdout[-400:], p.stderr[-1200:])


def _one_sub_windows(rng, n, mixed_frac=0.34):
    """Windows built to sit on the checks of Poa::guided_one_sub: arms that differ from the arm before them in one base, over
    drafts of two or three letters (homopolymer runs, repeats: several in-edge sources carry the letter before), with arms
    that lack a base or carry one more (triangles: the side entrances of the path), and the same substitution seen twice.
    A third of the windows mix in suffix arms (kROV: they end where the window ends and may start anywhere) and prefix arms
    (kLOV), cut at random and edited the same way."""
    from hypo_amd.batch import TextWindow
    wins = []
    for _ in range(n):
        L = int(rng.choice([8, 14, 22, 33, 45, 60, 90, 120]))
        letters = "ACGT"[:int(rng.choice([2, 3, 4]))]
        truth = "".join(letters[i] for i in rng.integers(0, len(letters), size=L))
        mixed = rng.random() < mixed_frac
        internal, prefix, suffix = [], [], []

        def edited(src):
            a = list(src)
            r = rng.random()
            if r < 0.55:                                  # one substitution
                a[int(rng.integers(len(a)))] = letters[int(rng.integers(len(letters)))]
            elif r < 0.70:                                # two
                for _ in range(2):
                    a[int(rng.integers(len(a)))] = letters[int(rng.integers(len(letters)))]
            elif r < 0.80 and len(a) > 4:                 # a base less
                del a[int(rng.integers(len(a)))]
            elif r < 0.90:                                # a base more
                a.insert(int(rng.integers(len(a))), letters[int(rng.integers(len(letters)))])
            return "".join(a)

        for _ in range(int(rng.integers(4, 26))):
            kind = "internal" if not mixed else str(rng.choice(["internal", "suffix", "suffix", "prefix"]))
            if kind == "internal":
                dst, src = internal, truth
            elif kind == "suffix":
                dst, src = suffix, truth[int(rng.integers(0, max(1, L - 3))):]
            else:
                dst, src = prefix, truth[:int(rng.integers(3, L + 1))]
            dst.append(edited(src))
            if rng.random() < 0.25:
                dst.append(dst[int(rng.integers(len(dst)))])
        wins.append(TextWindow(truth, internal, prefix, suffix))
    return wins


@pytest.mark.parametrize("seed,scores", [(1, (5, -4, -8, 3, -5, -4)), (2, (5, -4, -8, 3, -5, -4)), (3, (2, -1, -2, 3, -5, -4)),
                                         (4, (3, -6, -5, 3, -5, -4))])
def test_one_substitution_off_the_guide_vs_oracle(seed, scores):
    """Poa::guided_one_sub (an arm one base off the path of the arm before it is aligned without score rows) against the
    oracle, every window through the re-queue chain from class 0 on; (2, -1, -2) is a score set the shortcut accepts with
    other margins, (3, -6, -5) one it must refuse (a horizontal step costs less than a mismatch there)."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(4200 + seed)
    b = build_batch(_one_sub_windows(rng, 500))
    emu = emu_util.Emu()
    cons, res = emu.poa_chain(b, 0, scores=scores)[:2]
    want = oracle.Oracle().poa_batch(b, scores=scores)[0]
    n_ran = 0
    for i in range(b.n_windows):
        if res[i] == emu_util.RES_OK:
            assert cons[i] == want[i], i
            n_ran += 1
    assert n_ran > 450


def _second_end_window():
    from hypo_amd.batch import TextWindow
    return TextWindow("ATATAAGCACCCACCGTGTACACGAAAACCCGAATTNTCTGGGCACGCGGGATGTCTGCGGCCCGCTGACG",
                      ["ATATAAGCACCCCACCGTGTCACGAAAACCGATTATCTGGGACGCGGGATGTCTGCGGCCGCTGACG",
                       "ATATAAGCACCCACCGTGTACACGAAAACCGAATTATCTGGGCACGCGGGAGGTCTGCGGCCCGCTGACG"],
                      ["ATATAAGC", "ATATAACACCC", "ATTAAGCACCCACCGTGTACACGAAAA", "ATCTAAGCACC",
                       "ATATAAGCACCCACCGTGTACACGCAAAACCGAATTATCTGGACACGCGGGATGGT"],
                      ["TCTGCGGCCCGCTGACG", "GGGATGTCTGCGGCCCGCTGGACG"])


def test_klov_one_substitution_with_a_second_end():
    """Found by tests/sweep_parity_gpu.py (round 4009, window 3941; one in ~5 M windows): the prefix arm ATCTAAGCACC is one base
    off its guide, whose path ends ..A-C-C on nodes 9-11-12 of a run of three C; 9-10-11 spells the same letters, scores the
    same and ranks first.  The kLOV variant of Poa::guided_one_sub must see that second end (Poa::set_alive: a chain that
    joins the path is alive) and leave the arm to the score rows."""
    import oracle
    b = build_batch([_second_end_window()])
    want = oracle.Oracle().poa_batch(b)[0]
    emu = emu_util.Emu()
    for cfg in (1, 2, 3):
        cons, res = emu.poa_chain(b, cfg)[:2]
        assert res[0] == emu_util.RES_OK and cons[0] == want[0], cfg


def test_emulator_fuzz_script_runs():
    """tests/emu_fuzz_cpu.py (opt-in, hours on many cores) stays runnable: a few rounds of it here."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "emu_fuzz_cpu.py"), "5", "0.3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    last = out.stdout.strip().splitlines()[-1]
    assert last.endswith("bad 0") and int(last.split("compared")[1].split()[0]) > 0, last
