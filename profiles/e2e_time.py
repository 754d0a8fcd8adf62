"""Wall-clock of the whole polisher on BASELINE config C2 (5 Mbp draft, 30x short reads, one contig): regenerates the inputs of
tests/golden/e2e_5m_s11 with the committed generator, runs hypo_amd/_build/hypo a few times and prints the phase table of the
fastest run (the reference prints the same RESOURCES lines).  usage: python profiles/e2e_time.py [threads] [runs] [extra args...]"""
import hashlib
import os
import shlex
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_util as eu  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
extra = sys.argv[3:]
name = "e2e_5m_s11"
with tempfile.TemporaryDirectory() as d:
    man = eu.make_inputs(name, d)
    argv = shlex.split(man["command"])
    argv[0] = eu.BIN
    argv[argv.index("-t") + 1] = str(threads)
    argv += extra
    best = None
    for r in range(runs):
        t0 = time.time()
        p = subprocess.run(argv, cwd=d, capture_output=True, text=True)
        dt = time.time() - t0
        assert p.returncode == 0, p.stderr[-2000:]
        md5 = hashlib.md5(open(os.path.join(d, "hypo_draft.fasta"), "rb").read()).hexdigest()
        assert md5 == man["expected_fasta_md5"], "FASTA differs from the reference's"
        overall = [l for l in p.stdout.splitlines() if "Overall" in l]
        print(f"run {r}: process wall {dt:.3f} s; {overall[0].strip() if overall else ''}", flush=True)
        ov = float(overall[0].split("TIME=")[1].split()[0]) if overall else dt
        if best is None or ov < best[0]:
            best = (ov, p.stdout, p.stderr)
    print("---- fastest run ----")
    print(best[1])
    print(best[2])
