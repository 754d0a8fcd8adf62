#!/usr/bin/env python3
"""End to end at k = 15 (the k of BASELINE config C4: a 128 MiB solid set) and beyond C3's size: N x 1 Mbp contigs, 30x 150-bp reads from
the C++ generator, `hypo -p 10` and `-p 50` (different contig batches must give the same FASTA).  No reference md5 at this size (the real
reference binary needs ~4 min per 100 Mbp on 8 threads in the build container, and it is not on the GPU box): this run is about Mbp/s, peak RSS
and batching, not parity.  usage: r03_e2e_k15.py [contigs=250]"""
import hashlib
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = os.path.join(os.environ.get("TMPDIR", "/tmp"), "e2e_k15")
shutil.rmtree(D, ignore_errors=True)
os.makedirs(D)
free_gb = shutil.disk_usage(D).free / 2**30
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
if free_gb < 40:
    n = min(n, 150)
if free_gb < 20:
    n = min(n, 60)
print(f"free disk {free_gb:.0f} GB -> {n} contigs of 1 Mbp; {os.cpu_count()} host threads", flush=True)
gen = os.path.join(ROOT, "tests", "_build", "gen_e2e_fast")
os.makedirs(os.path.dirname(gen), exist_ok=True)
subprocess.check_call(["g++", "-O2", "-fopenmp", "-o", gen, os.path.join(ROOT, "tests", "golden", "gen_e2e_fast.cpp"), "-lz"])
t0 = time.time()
rep = subprocess.check_output([gen, D, "77", str(n), "1000000", "15", "30", "150", "2000"], text=True)
print(f"inputs generated in {time.time() - t0:.1f} s: {rep.strip()}; sr.sam {os.path.getsize(os.path.join(D, 'sr.sam')) / 2**30:.1f} GiB", flush=True)
md5s = []
for p in (10, 50):
    out = f"out_p{p}.fa"
    argv = [os.path.join(ROOT, "hypo_amd", "_build", "hypo"), "-d", "draft.fa", "-r", "reads.fa", "-s", f"{n}m", "-c", "30", "-b", "sr.sam",
            "-t", str(min(64, os.cpu_count() or 8)), "-i", "-p", str(p), "-o", out]
    t0 = time.time()
    proc = subprocess.Popen(argv, cwd=D, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    hwm = 0
    while proc.poll() is None:
        try:
            for line in open(f"/proc/{proc.pid}/status"):
                if line.startswith("VmHWM"):
                    hwm = max(hwm, int(line.split()[1]))
        except OSError:
            pass
        time.sleep(0.05)
    so, se = proc.communicate()
    dt = time.time() - t0
    assert proc.returncode == 0, so[-1500:] + se[-1500:]
    h = hashlib.md5()
    with open(os.path.join(D, out), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    md5s.append(h.hexdigest())
    overall = [l.strip() for l in so.splitlines() if "Overall" in l]
    kline = [l.strip() for l in so.splitlines() if "Value of K" in l]
    ov = float(overall[0].split("TIME=")[1].split()[0]) if overall and "TIME=" in overall[0] else dt
    print(f"== -p {p}: {kline[0] if kline else ''}\n   process wall {dt:.2f} s, Overall {ov:.2f} s = {n / ov:.1f} Mbp/s, peak RSS {hwm / 1024:.0f} MB, FASTA md5 {md5s[-1]}", flush=True)
    for l in so.splitlines():
        if "TIME=" in l:
            print("   ", l.strip()[:150])
    print("   ", [l for l in se.splitlines() if "hypo_gpu_arms_poa" in l][:2])
print("FASTA identical across batchings:", md5s[0] == md5s[1])
shutil.rmtree(D, ignore_errors=True)
