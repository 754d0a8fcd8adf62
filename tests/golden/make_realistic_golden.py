#!/usr/bin/env python3
"""Makes the round-6 goldens of NON-i.i.d. inputs (tests/golden/e2e_real_*.manifest.json): genomes with tandem repeats, homopolymer runs and
dispersed copies, a second haplotype, read indel errors, mis-placed reads (tests/golden/gen_e2e_fast.cpp: --repeats --diploid --read-indel
--mismap), one of them with noisy long reads (`-B`).

Build container only.  The expected FASTA is what the REFERENCE's own code wrote for the files: its sources compiled in place by
oracle/Makefile (no CMake, nothing stubbed) behind oracle/ref_arms_harness.cpp's hyporef_fasta_bam2 — an independent BGZF / BAM decoder
hands every record to hypo::Alignment as bam1_t, then the reference's short- and long-read stages, Window::generate_consensus and
operator<<(Contig) run (src/Hypo.cpp:126-268).  The manifest also records which rarely-taken branches of the segmentation the set reaches,
counted by this repo's host mirror on the same files (HYPO_STAGE_COUNTERS=1; its regions and FASTA equal the reference's on these sets, which
tests/test_gpu_e2e.py and tests/test_host_e2e_cpu.py check): solid k-mers accepted / refused in the 40-80 % support branch of SR detection
(src/Contig.cpp:96-127), Contig::force_divide calls (:630-711), minimizers dropped as recurring or poly-base (:455-524), and the histogram of
region types (OTH / SW / WS / MW / WM windows only come out of force_divide)."""
import collections
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_util as eu  # noqa: E402
import oracle  # noqa: E402

SETS = {
    "e2e_real_5m_s131": dict(seed=131, contigs=5, contig_len=1_000_000, k=11, coverage=30, read_len=150, read_sub_ppm=2000, size="5m",
                             flags=["--bam", "--fast-hash", "--repeats", "150000", "--diploid", "1000", "--read-indel", "500", "--mismap", "2000"], extra=["-p", "2"]),
    "e2e_real_hifi_s133": dict(seed=133, contigs=3, contig_len=1_000_000, k=13, coverage=30, read_len=8000, read_sub_ppm=500, size="100m",
                               flags=["--bam", "--fast-hash", "--repeats", "100000", "--diploid", "1500", "--read-indel", "1500"], extra=[]),
    "e2e_real_long_s137": dict(seed=137, contigs=3, contig_len=1_000_000, k=11, coverage=30, read_len=150, read_sub_ppm=2000, size="3m",
                               flags=["--bam", "--fast-hash", "--repeats", "100000", "--diploid", "1000", "--read-indel", "300", "--mismap", "1000",
                                      "--long", "25", "6000", "--gaps", "40000", "3000"], extra=[]),
}


def main():
    gen = eu.build_fast_generator()
    eu.build_binary()
    eu.build_shim()
    ref = oracle.RefArms()
    for name, a in SETS.items():
        d = tempfile.mkdtemp(prefix="hypo_real_")
        try:
            rep = json.loads(subprocess.check_output([gen, d, str(a["seed"]), str(a["contigs"]), str(a["contig_len"]), str(a["k"]), str(a["coverage"]),
                                                      str(a["read_len"]), str(a["read_sub_ppm"])] + a["flags"], text=True))
            with_long = "--long" in a["flags"]
            rr = ref.fasta_file(os.path.join(d, "draft.fa"), os.path.join(d, "sr.bam"), a["k"], os.path.join(d, "aux", "solid_kmers.bvsd"), os.path.join(d, "ref.fa"),
                                long_path=os.path.join(d, "lr.bam") if with_long else None)
            md5 = hashlib.md5(open(os.path.join(d, "ref.fa"), "rb").read()).hexdigest()
            cmd = ["hypo", "-d", "draft.fa", "-r", "reads.fa", "-s", a["size"], "-c", str(a["coverage"]), "-b", "sr.bam"] + (["-B", "lr.bam"] if with_long else []) + ["-t", "8", "-i"] + a["extra"]
            env = dict(os.environ, LD_LIBRARY_PATH=eu.SHIM_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), HYPO_STAGE_COUNTERS="1",
                       HYPO_REGION_DUMP=os.path.join(d, "regions.tsv"))
            p = subprocess.run([eu.BIN] + cmd[1:], cwd=d, env=env, capture_output=True, text=True, timeout=3000)
            assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-1000:]
            ours = hashlib.md5(open(os.path.join(d, "hypo_draft.fasta"), "rb").read()).hexdigest()
            assert ours == md5, f"{name}: this repo's host pipeline (CPU shim) differs from the reference compiled in place"
            m = re.search(r"stage counters: solid k-mers accepted with 40-80 % support (\d+), refused after another such k-mer (\d+); force_divide calls (\d+); "
                          r"minimizers dropped as recurring (\d+), as poly-base (\d+)", p.stdout)
            types = collections.Counter(l.split("\t")[3].strip() for l in open(os.path.join(d, "regions.tsv")))
            man = {"generator": "tests/golden/gen_e2e_fast.cpp",
                   "args": {"seed": a["seed"], "contigs": a["contigs"], "contig_len": a["contig_len"], "k": a["k"], "coverage": a["coverage"], "read_len": a["read_len"],
                            "read_sub_ppm": a["read_sub_ppm"], "flags": a["flags"]},
                   "command": " ".join(cmd), "size_flag": a["size"], "generator_report": rep, "expected_fasta_md5": md5,
                   "pinned_by": "oracle/_ref/libhyporef_arms.so (hyporef_fasta_bam2): the reference's sources compiled in place, records handed over as bam1_t by the harness's own BAM decoder",
                   "reference_counts": {"alignments": rr["alignments"], "invalid": rr["invalid"], "regions": rr["regions"], "windows": rr["windows"]},
                   "branches_reached": {"sr_kmers_accepted_40_80": int(m.group(1)), "sr_kmers_refused_40_80": int(m.group(2)), "force_divide_calls": int(m.group(3)),
                                        "minimizers_dropped_recurring": int(m.group(4)), "minimizers_dropped_poly_base": int(m.group(5)),
                                        "counted_by": "this repo's host mirror on the same files (HYPO_STAGE_COUNTERS=1), FASTA identical to the reference's"},
                   "region_types": dict(sorted(types.items()))}
            json.dump(man, open(os.path.join(HERE, name + ".manifest.json"), "w"), indent=1)
            print(name, md5, man["branches_reached"], man["region_types"], flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
