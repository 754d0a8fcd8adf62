"""Command line of the `hypo` binary against the reference's flag sheet (src/main.cpp:46-67,100-113,
124-358,490-528): mandatory arguments, guards, k derivation, default output name.  Everything here happens before the
device is touched, so it runs in the GPU-less container (the run itself then stops at hypo_gpu_init: no CPU fallback)."""
import os
import subprocess

import pytest

import e2e_util as eu


@pytest.fixture(scope="module")
def hypo_bin():
    try:
        return eu.build_binary()
    except Exception as e:
        pytest.skip(f"cannot build the hypo binary here: {e}")


def _run(hypo_bin, argv, cwd):
    return subprocess.run([hypo_bin] + argv, cwd=str(cwd), capture_output=True, text=True, timeout=120)


@pytest.fixture()
def files(tmp_path):
    for n in ("draft.fa", "reads.fa", "sr.sam", "lr.sam"):
        (tmp_path / n).write_text(">x\nACGT\n" if n.endswith(".fa") else "@HD\tVN:1.6\n")
    return tmp_path


def test_help_and_unknown_option_print_usage_and_exit_0(hypo_bin, tmp_path):
    for argv in (["-h"], ["--help"], ["--no-such-flag"]):
        p = _run(hypo_bin, argv, tmp_path)
        assert p.returncode == 0 and "--reads-short" in p.stdout and "--bam-lr" in p.stdout     # main.cpp:302-304


def test_missing_mandatory_arguments(hypo_bin, files):
    full = {"-r": "reads.fa", "-d": "draft.fa", "-b": "sr.sam", "-c": "30", "-s": "5m"}
    for drop in full:
        argv = [x for k, v in full.items() if k != drop for x in (k, v)]
        p = _run(hypo_bin, argv, files)
        assert p.returncode == 1 and "Too few arguments" in p.stderr, drop


def test_guards(hypo_bin, files):
    base = ["-r", "reads.fa", "-d", "draft.fa", "-b", "sr.sam", "-c", "30", "-s", "5m"]
    assert _run(hypo_bin, base + ["-g", "3"], files).returncode == 1             # gap penalties must be negative (main.cpp:234-237)
    assert _run(hypo_bin, base + ["-G", "0"], files).returncode == 1             # (main.cpp:253-256)
    assert _run(hypo_bin, ["-r", "reads.fa", "-d", "nope.fa", "-b", "sr.sam", "-c", "30", "-s", "5m"], files).returncode == 1
    assert _run(hypo_bin, ["-r", "@nolist.txt", "-d", "draft.fa", "-b", "sr.sam", "-c", "30", "-s", "5m"], files).returncode == 1
    assert _run(hypo_bin, base[:-4] + ["-c", "0", "-s", "5m"], files).returncode == 1
    p = _run(hypo_bin, base[:-2] + ["-s", "5x"], files)
    assert p.returncode == 1 and "units" in p.stderr
    p = _run(hypo_bin, base[:-2] + ["-s", "2.5"], files)
    assert p.returncode == 1 and "absolute number" in p.stderr


@pytest.mark.parametrize("size,k", [("10k", 7), ("100k", 9), ("1m", 11), ("5m", 11), ("100m", 13), ("250m", 15), ("3g", 17),
                                    ("4096", 7), ("999", 5), ("2g", 15)])
def test_kmer_length_from_genome_size(hypo_bin, files, size, k):
    """smallest odd k with 4^k >= genome size, through the reference's integer arithmetic (main.cpp:490-528)"""
    p = _run(hypo_bin, ["-r", "reads.fa", "-d", "draft.fa", "-b", "sr.sam", "-c", "30", "-s", size], files)
    assert f"({size}): {k}\n" in p.stdout, p.stdout


def test_read_list_file_and_no_cpu_fallback(hypo_bin, files):
    (files / "list.txt").write_text("reads.fa\n")
    p = _run(hypo_bin, ["-r", "@list.txt", "-d", "draft.fa", "-b", "sr.sam", "-c", "30", "-s", "5m"], files)
    assert "Beginning from stage: 0" in p.stdout
    assert os.path.isdir(files / "aux")                                         # created regardless of -i (main.cpp:326)
    if p.returncode != 0:                                                       # GPU-less container: the product refuses to run
        assert "hip" in p.stderr.lower() or "device" in p.stderr.lower() or "KMC" in p.stderr
