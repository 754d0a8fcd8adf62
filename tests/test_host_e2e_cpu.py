"""Host pipeline above the C-ABI, end to end, in the GPU-less container: SAM parsing, solid-kmer / minimiser
support, strong regions, minimiser cuts, arms (short and long reads), FASTA reassembly of hypo_amd/csrc/host are
checked against the REAL reference's outputs with the CPU oracle standing in for the device (tests/shim).  The
device itself is checked by the same goldens in tests/test_gpu_e2e.py."""
import pytest
import e2e_util as eu


@pytest.fixture(scope="module")
def built():
    eu.build_shim()
    try:
        eu.build_binary()
    except Exception as e:                     # hipcc missing: the host binary links the real library
        pytest.skip(f"cannot build the hypo binary here: {e}")


@pytest.mark.parametrize("name", eu.CASES)
def test_e2e_host_pipeline_matches_reference(built, name, tmp_path):
    man, _ = eu.run_case(name, tmp_path, "shim")
    assert eu.check_outputs(name, tmp_path, man) > 0


def test_e2e_batches_and_threads_do_not_change_the_result(built, tmp_path):
    # -p 1 (one contig per batch) and another thread count: same bytes (reference: src/Hypo.cpp:104-113)
    man, _ = eu.run_case("e2e_20k_s1", tmp_path, "shim", threads=7)
    eu.check_outputs("e2e_20k_s1", tmp_path, man)


@pytest.mark.parametrize("name", ["e2e_20k_s1", "e2e_5ctg_long_s21"])
def test_e2e_bam_input_gives_the_same_result(built, name, tmp_path):
    """alignments as BAM (BGZF blocks, binary records, NM as a small-integer tag) instead of SAM text"""
    man, _ = eu.run_case(name, tmp_path, "shim", as_bam=True)
    eu.check_outputs(name, tmp_path, man)


@pytest.mark.parametrize("seed", sorted(eu.messy_seeds())[:12])
def test_e2e_messy_seeds_match_reference(built, seed, tmp_path):
    eu.run_messy_seed(seed, eu.messy_seeds()[seed], tmp_path, "shim")


def test_e2e_two_device_contexts_shard_the_window_batch(built, tmp_path):
    """`--gpus 2`: the host initialises two contexts and sends each contig batch through hypo_gpu_poa_batch_sharded (the shim
    answers the two halves of the window list separately); the polished FASTA and the per-region records stay the reference's."""
    man, p = eu.run_case("e2e_5ctg_long_s21", tmp_path, "shim", extra_args=["--gpus", "2"])
    assert "sharded call over 2 contexts" in p.stderr
    eu.check_outputs("e2e_5ctg_long_s21", tmp_path, man)


def test_e2e_ccs_windows_match_the_reference_with_kind_ccs_applied(built, tmp_path):
    """--ccs-windows: the window sizes `-k ccs` was meant to select (src/main.cpp:572-585; the reference parses -k and never
    applies it, :312).  Golden from a scratch copy of the reference in which line 312 calls set_kind(kind), run with -k ccs on
    3-kbp HiFi-like reads: 236 windows of ~ 500 bp, graphs of up to 546 nodes."""
    man, _ = eu.run_case("e2e_120k_ccs_s47", tmp_path, "shim")
    assert eu.check_outputs("e2e_120k_ccs_s47", tmp_path, man) > 200


@pytest.mark.parametrize("bam", [False, True])
def test_e2e_long_read_edit_distance_filter_counts_and_fasta(built, bam, tmp_path):
    """`-n 7` on the C4-in-small set (one 5 Mbp contig, 40x long reads with 8 % errors): the NM-based filter of the long-read
    constructor (src/Alignment.cpp:51-58: edit distance * 100 / reference span, integers, compared with the threshold) drops 22 401
    of the 24 894 long reads.  The flat long-read parser (Hypo::parse_block) reads NM from a SAM line and from a BAM record's
    optional fields alike: the loaded / invalid counts and the polished FASTA are the REAL reference's (manifest)."""
    import re
    man, p, _, _ = eu.run_fast_case("e2e_c4s_5m_s55_n7", tmp_path, threads=8, bam=bam, device="shim")
    counts = re.findall(r"Number of alignments \(Batch 0\): loaded \((\d+)\) invalid \((\d+)\)", p.stdout)
    assert [int(x) for x in counts[-1]] == man["expected_long_reads_loaded_invalid"], counts
    assert eu.fasta_md5(tmp_path) == man["expected_fasta_md5"]



def test_require_device_turns_a_host_fallback_into_an_error(built, tmp_path):
    """`hypo --require-device`: over the CPU shim (which answers the device stages of the support votes and arm selection with
    "unsupported", so the host loops would take over) the run ends with exit code 1 and says which stage it was; without the flag
    the same run goes through the host loops and matches the reference (the tests above)."""
    import os
    import shlex
    import subprocess
    man = eu.make_inputs("e2e_20k_s1", tmp_path)
    argv = shlex.split(man["command"])
    argv[0] = eu.BIN
    env = dict(os.environ, LD_LIBRARY_PATH=eu.SHIM_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    env.pop("HYPO_REQUIRE_DEVICE", None)
    p = subprocess.run(argv + ["--require-device"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 1, p.stdout[-500:] + p.stderr[-500:]
    assert "--require-device" in p.stderr and "would be computed on the host" in p.stderr
    p = subprocess.run(argv, cwd=str(tmp_path), env=dict(env, HYPO_REQUIRE_DEVICE="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "would be computed on the host" in p.stderr


def test_realistic_golden_with_long_reads_over_the_shim(built, tmp_path):
    """A non-i.i.d. set (repeats, second haplotype, read indels, mis-placed reads, `-B` long reads: tests/golden/make_realistic_golden.py)
    through the host pipeline over the CPU shim: FASTA = what the reference compiled in place wrote (the manifest's md5), and, where
    oracle/_ref/libhyporef_arms.so exists, the reference polishes the same files again here."""
    import oracle
    man, p, dt, rss = eu.run_fast_case("e2e_real_long_s137", tmp_path, threads=4, device="shim")
    assert eu.fasta_md5(tmp_path) == man["expected_fasta_md5"]
    if oracle.RefArms.available():
        rr = oracle.RefArms().fasta_file(str(tmp_path / "draft.fa"), str(tmp_path / "sr.bam"), 11, str(tmp_path / "aux" / "solid_kmers.bvsd"), str(tmp_path / "ref.fa"),
                                         long_path=str(tmp_path / "lr.bam"))
        assert open(tmp_path / "ref.fa", "rb").read() == open(tmp_path / "hypo_draft.fasta", "rb").read()
        assert rr["windows"] == man["reference_counts"]["windows"]


def test_output_appears_only_when_the_run_succeeded(built, tmp_path):
    """The polished FASTA is written to <output>.tmp and renamed when complete: a run that dies after its first batch — here a
    record that names a contig the draft does not have, the reference's own fatal error (src/Hypo.cpp:303-306) — leaves an earlier file under
    the output's name untouched and no .tmp behind; a good run replaces it."""
    import os
    import subprocess
    gen = eu.build_fast_generator()
    subprocess.check_output([gen, str(tmp_path), "151", "3", "60000", "9", "30", "150", "2000"])          # three contigs, SAM text
    argv = [eu.BIN, "-d", "draft.fa", "-r", "reads.fa", "-s", "100k", "-c", "30", "-b", "sr.sam", "-t", "4", "-i", "-p", "1"]
    env = dict(os.environ, LD_LIBRARY_PATH=eu.SHIM_DIR)
    out = tmp_path / "hypo_draft.fasta"
    out.write_text("OLD RESULT\n")
    sam = tmp_path / "sr.sam"
    good = sam.read_text()
    sam.write_text(good + "zz\t0\tno_such_contig\t5\t60\t10M\t*\t0\t0\tACGTACGTAC\t*\n")
    p = subprocess.run(argv, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "does not exist in the draft" in p.stderr
    assert "polished windows (Batch 0)" in p.stdout                       # it had got past the first batch
    assert out.read_text() == "OLD RESULT\n" and not (tmp_path / "hypo_draft.fasta.tmp").exists()
    sam.write_text(good)
    p = subprocess.run(argv, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-500:]
    recs = eu.fasta_records(str(out))
    assert list(recs) == ["ctg1", "ctg2", "ctg3"] and all(len(v) > 59000 for v in recs.values()) and not (tmp_path / "hypo_draft.fasta.tmp").exists()


def test_window_longer_than_every_table_driven_class_over_the_shim(built, tmp_path):
    """The 4 000-base LONG window of tests/e2e_util.py: run_homopolymer_set through the host pipeline (CPU shim): FASTA = the reference compiled
    in place.  (On the device the window runs in size class 6 and piece mode widens its halo: tests/test_gpu_e2e.py.)"""
    import oracle
    if not oracle.RefArms.available():
        pytest.skip("oracle/_ref/libhyporef_arms.so not built (the real reference only exists in the build container)")
    p, same, longest = eu.run_homopolymer_set(tmp_path, "shim", threads=4)
    assert longest > 3900 and same
