#!/usr/bin/env python3
"""Differential fuzz (build container only): `gen_e2e.generate_messy(seed)` inputs through the REAL reference binary
(HYPO_REF_BIN, built per SURVEY.md Appendix B; HYPO_REF_LD = its htslib directory) and through this repo's `hypo` over
the oracle shim; the polished FASTA must be byte-identical.   usage: fuzz_e2e.py <first seed> <last seed (exclusive)>
Seeds 100-259 were run in round 1 without a mismatch; two of them are committed as `e2e_messy_*` goldens."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import gen_e2e  # noqa: E402

REF = os.environ.get("HYPO_REF_BIN", "/tmp/oracle/ref/build/bin/hypo")
REF_LD = os.environ.get("HYPO_REF_LD", "/tmp/oracle/hts")
MINE = os.path.join(ROOT, "hypo_amd", "_build", "hypo")
SHIM = os.path.join(ROOT, "tests", "_build", "shim")


def run(seed, base="/tmp/fuzz"):
    d = os.path.join(base, str(seed))
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    cmd, _, _ = gen_e2e.generate_messy(d, seed)
    cmd[cmd.index("-t") + 1] = "4"
    a = subprocess.run([REF] + cmd + ["-o", "ref.fa"], cwd=d, env=dict(os.environ, LD_LIBRARY_PATH=REF_LD), capture_output=True, text=True, timeout=900)
    b = subprocess.run([MINE] + cmd + ["-o", "mine.fa"], cwd=d, env=dict(os.environ, LD_LIBRARY_PATH=SHIM), capture_output=True, text=True, timeout=900)
    ra = open(d + "/ref.fa", "rb").read() if os.path.exists(d + "/ref.fa") else None
    rb = open(d + "/mine.fa", "rb").read() if os.path.exists(d + "/mine.fa") else None
    ok = ra is not None and ra == rb
    print(seed, "OK" if ok else "MISMATCH", " ".join(cmd[10:]), "rc", a.returncode, b.returncode, flush=True)
    if not ok:
        print("  ref :", (a.stdout + a.stderr)[-300:].replace("\n", " | "))
        print("  mine:", (b.stdout + b.stderr)[-300:].replace("\n", " | "))
    return ok


if __name__ == "__main__":
    bad = [s for s in range(int(sys.argv[1]), int(sys.argv[2])) if not run(s)]
    print("bad seeds", bad)
    sys.exit(1 if bad else 0)
