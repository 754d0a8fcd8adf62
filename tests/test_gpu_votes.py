"""MI355X: the support votes of the device (support_kernel.hip behind hypo_gpu_support_kmers[_kept] / hypo_gpu_support_minimizers)
against the host loops that restate Alignment::update_solidkmers_support / update_minimisers_support (src/Alignment.cpp:65-220,
hypo_amd/csrc/host/Alignment.cpp — themselves pinned to the real reference by the end-to-end goldens and region dumps of the CPU
suite), counter by counter: the `hypo` binary writes every KmerInfo / MWMinimiserInfo coverage and support counter of every contig
(HYPO_DUMP_VOTES) once with the votes on the device and once with HYPO_HOST_SUPPORT=1; the two files must be the same bytes."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import e2e_util as eu

pytestmark = pytest.mark.gpu


def _votes(workdir, argv, host):
    out = os.path.join(str(workdir), "votes_host.bin" if host else "votes_dev.bin")
    env = dict(os.environ)
    env["HYPO_DUMP_VOTES"] = out
    if host:
        env["HYPO_HOST_SUPPORT"] = "1"
    p = subprocess.run(argv, cwd=str(workdir), env=env, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "oracle_device_shim" not in p.stderr
    on_dev = "k-mer support counted on the device" in p.stdout
    assert on_dev != host, "the votes were not counted where the test asked for"
    return np.fromfile(out, dtype=np.uint8), p


@pytest.mark.parametrize("k,size_flag,contigs,contig_len,sub_ppm,p", [
    (11, "5m", 6, 200_000, 2_000, 0),        # k <= 16: 32-bit k-mer ids, few solid k-mers
    (15, "250m", 5, 400_000, 10_000, 2),     # dense (a random genome: nearly every 15-mer is unique), 1 % read errors, three batches
    (17, "3g", 3, 300_000, 5_000, 0),        # k = 17: 64-bit ids, the shape of the 3 Gbp row
])
def test_device_votes_equal_the_host_loops_counter_by_counter(tmp_path, k, size_flag, contigs, contig_len, sub_ppm, p):
    if not os.path.exists(eu.BIN):
        eu.build_binary()
    if k == 17 and shutil.disk_usage("/dev/shm").free < (40 << 30):
        pytest.skip("the generator counts 17-mers in 17 GB")
    gen = eu.build_fast_generator()
    subprocess.check_output([gen, str(tmp_path), str(1000 + k), str(contigs), str(contig_len), str(k), "30", "150", str(sub_ppm), "--bam", "--fast-hash"])
    argv = [eu.BIN, "-d", "draft.fa", "-r", "reads.fa", "-s", size_flag, "-c", "30", "-b", "sr.bam", "-t", "16", "-i", "-p", str(p)]
    dev, pd = _votes(tmp_path, argv, host=False)
    host, ph = _votes(tmp_path, argv, host=True)
    assert dev.size == host.size and dev.size > 100_000, (dev.size, host.size)
    if not (dev == host).all():
        first = int(np.nonzero(dev != host)[0][0])
        raise AssertionError(f"vote counters differ from byte {first} on ({int((dev != host).sum())} bytes of {dev.size})")
    assert int(dev.astype(np.uint64).sum()) > 1_000_000          # (the counters are not all zero)


@pytest.mark.parametrize("seed", sorted(eu.messy_seeds())[:24])
def test_device_votes_equal_the_host_loops_on_messy_sets(tmp_path, seed):
    """the "messy" generator's sets: indels, clipped reads, unsorted files, contigs without reads, Ns"""
    import importlib.util
    if not os.path.exists(eu.BIN):
        eu.build_binary()
    spec = importlib.util.spec_from_file_location("gen_e2e", os.path.join(eu.GOLD, "gen_e2e.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    argv, _, _ = gen.generate_messy(str(tmp_path), seed)
    argv = [eu.BIN] + argv
    if "--host-arms" in argv:
        pytest.skip("this seed runs the host loops anyway")
    env_probe = dict(os.environ); env_probe["HYPO_DUMP_VOTES"] = os.path.join(str(tmp_path), "probe.bin")
    p = subprocess.run(argv, cwd=str(tmp_path), env=env_probe, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    if "k-mer support counted on the device" not in p.stdout:
        pytest.skip("this set's votes are counted on the host (its alignments could not be made resident)")
    dev = np.fromfile(env_probe["HYPO_DUMP_VOTES"], dtype=np.uint8)
    host, _ = _votes(tmp_path, argv, host=True)
    assert dev.size == host.size and (dev == host).all()
