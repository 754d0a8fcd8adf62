#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X polishing hot path (BASELINE.json `metric`).

One step = one pass of the hot path over one synthetic batch with the shape of BASELINE.json configs[1]
("E. coli 5 Mbp, 30x short reads, 1 x MI355X — GPU POA + kmer scan"): the solid-kmer scan of a 5 Mbp
4-bit packed contig (k = 11) followed by the POA consensus of the ~97 k windows the reference builds on
such a draft (window-shape table hypo_amd/shapes/c1_shape.npz, see hypo_amd/sim.py).  Inputs are resident
in HBM before the timed region; each step runs entirely through the C-ABI of libhypo_gpu.so.

N > 1 (launched by torch.distributed.run, one rank per GPU) runs BASELINE.json configs[2] ("synthetic 100 Mbp contig set,
30x short reads, 1 vs 2 vs 4 vs 8 MI355X window sharding over xGMI") as STRONG scaling: ONE batch of 20 x 97 078 C1-shaped
windows (what 100 Mbp of draft yields) and the 100 contigs of 1 Mbp are the same whatever N is; the windows are
cut into cost-balanced contiguous ranges (hypo_amd/dist.py: shard_contiguous, the cost model the C++ host uses), every rank
scans its contigs and polishes its range, and the step ends with the one real exchange of the path: an RCCL all-gather of
the per-window consensus lengths and bytes (what contig re-assembly needs).  value = windows of the whole
batch / max-over-ranks time; `--workload c3` runs the same job on one GPU, `--workload c2 --gpus N` the weak-scaling
variant of round 1 (one C2 batch per rank).

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events recorded by the library on
the stream its kernels run on; `cpu_baseline` times oracle/ (this repo's bit-exact CPU restatement of the
reference's OpenMP/spoa path) on the same batch on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# The library runs up to seven HIP streams (two batches in flight + three side streams of a POA call + the caller's); a
# ROCm process gets four hardware queues unless this is set before the runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_WINDOWS = 97078          # valid windows of the C1-shaped 5 Mbp run
CONTIG_BASES = 5_000_000
K = 11                     # -s 5m => k = 11 (src/main.cpp:490-528)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def end_to_end_leg():
    """Runs hypo_amd/_build/hypo on the 5 Mbp golden set; returns the JSON object for the bench line (or an error note)."""
    import hashlib
    import importlib.util
    import re
    import shutil
    import subprocess
    import tempfile
    binp = os.path.join(ROOT, "hypo_amd", "_build", "hypo")
    manp = os.path.join(ROOT, "tests", "golden", "e2e_5m_s11.manifest.json")
    if not (os.path.exists(binp) and os.path.exists(manp)):
        return {"error": "hypo binary or golden manifest missing"}
    man = json.load(open(manp))
    spec = importlib.util.spec_from_file_location("gen_e2e", os.path.join(ROOT, "tests", "golden", "gen_e2e.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    d = tempfile.mkdtemp(prefix="hypo_bench_e2e_")
    try:
        a = man["args"]
        gen.generate(d, a["seed"], a["G"], a["long"], a["k"])
        for f, want in man["inputs_md5"].items():
            if hashlib.md5(open(os.path.join(d, f), "rb").read()).hexdigest() != want:
                return {"error": f"regenerated {f} differs from the golden's input"}
        threads = min(32, os.cpu_count() or 1)
        argv = [binp] + man["command"].split()[1:]
        argv[argv.index("-t") + 1] = str(threads)
        best, best_wall = None, None
        for _ in range(2):
            tw = time.perf_counter()
            p = subprocess.run(argv, cwd=d, capture_output=True, text=True, timeout=600)
            wall = time.perf_counter() - tw
            if p.returncode != 0:
                return {"error": (p.stdout + p.stderr)[-300:]}
            m = re.search(r"Overall\. \): TIME= ([0-9.eE+-]+) sec", p.stdout)
            t = float(m.group(1)) if m else None
            if t is not None and (best is None or t < best):
                best = t
            best_wall = wall if best_wall is None else min(best_wall, wall)
        got = hashlib.md5(open(os.path.join(d, "hypo_draft.fasta"), "rb").read()).hexdigest()
        if got != man["expected_fasta_md5"]:
            raise SystemExit("bench: end-to-end FASTA differs from the real reference's — refusing to report a number")
        # The reference's own polish of the same contig ON THIS BOX (the Mbp/s half of the metric's CPU baseline): its Alignment / Contig /
        # Window / spoa code compiled in place (oracle/_ref/libhyporef_arms.so, hyporef_fasta: support votes, division, arms, POA,
        # operator<<(Contig)) on all host cores, the records handed over in memory — so without the reference's BAM / SAM parsing and
        # solid-k-mer loading, which this repo's seconds include.  Its FASTA must be the bytes this repo wrote.
        same_box = None
        try:
            import oracle
            if oracle.RefArms.available():
                ref = oracle.RefArms()
                fa = open(os.path.join(d, "draft.fa")).read().split("\n")
                cname, draft = fa[0][1:].split()[0], "".join(fa[1:])
                recs = ref.sam_records(os.path.join(d, "sr.sam"), cname, 2)
                tr = time.perf_counter()
                ref.fasta(draft.encode(), cname, a["k"], os.path.join(d, "aux", "solid_kmers.bvsd"), recs, os.path.join(d, "ref_same_box.fa"))
                tr = time.perf_counter() - tr
                if open(os.path.join(d, "ref_same_box.fa"), "rb").read() != open(os.path.join(d, "hypo_draft.fasta"), "rb").read():
                    raise SystemExit("bench: the in-place reference's FASTA differs from this repo's — refusing to report a number")
                same_box = {"seconds": round(tr, 3), "mbp_per_s": round(a["G"] / 1e6 / tr, 2), "threads": os.cpu_count(), "kind": "reference",
                            "what": "hyporef_fasta: the reference's own stage + Window::generate_consensus + operator<<(Contig) compiled in place, all host cores (OpenMP), "
                                    "records in memory (no alignment-file parsing, no solid-k-mer loading); FASTA byte-identical to this repo's"}
        except (ImportError, OSError, RuntimeError) as ex:
            same_box = {"error": str(ex)[:200]}
        return {"mbp_per_s": round(a["G"] / 1e6 / best, 2), "seconds": round(best, 4), "process_wall_seconds": round(best_wall, 4),
                "host_threads": threads, "cpu_reference_same_box": same_box,
                "workload": "C2 end to end: 5 Mbp draft, 30x 150-bp reads (1 M records of SAM text), k = 11, hypo binary = host pipeline + device, best of 2; "
                            "seconds = the binary's Overall timer (the reference's own measure), process_wall_seconds adds process start, device init and teardown",
                "fasta": "md5 identical to the real reference's output for these inputs"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _scratch_dir(prefix, need_bytes):
    """Memory-backed scratch when /dev/shm has the room (the GPU boxes: 1.5 TB), else the default temp directory."""
    import shutil
    import tempfile
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need_bytes * 2:
            return tempfile.mkdtemp(prefix=prefix, dir="/dev/shm")
    except OSError:
        pass
    return tempfile.mkdtemp(prefix=prefix)


def _gpu_busy_sampler():
    """(start, stop) of a 20 Hz sampler of the first GPU's busy percentage (amdgpu sysfs); stop() returns the mean or None."""
    import glob
    import threading
    paths = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
    vals, flag = [], [True]
    def run():
        while flag[0] and paths:
            try:
                vals.append(float(open(paths[0]).read().strip()))
            except (OSError, ValueError):
                pass
            time.sleep(0.05)
    th = threading.Thread(target=run, daemon=True)
    def stop():
        flag[0] = False
        th.join(timeout=1)
        return round(sum(vals) / len(vals), 2) if vals else None
    return th.start, stop


def _run_hypo(argv, cwd, timeout=3000):
    """Runs the binary; returns (returncode, stdout, stderr, wall seconds, peak RSS in MB from /proc VmHWM, mean GPU busy %)."""
    import subprocess
    import threading
    peak = [0]
    start_busy, stop_busy = _gpu_busy_sampler()
    tw = time.perf_counter()
    # (HYPO_REQUIRE_DEVICE: a stage that falls back to the host loops ends the run with an error instead of a number)
    proc = subprocess.Popen(argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, HYPO_REQUIRE_DEVICE="1"))
    start_busy()
    def watch():
        # (VmHWM belongs to the new address space; getrusage's ru_maxrss, which the binary prints, starts from this python process's own high-water mark)
        while proc.poll() is None:
            try:
                for line in open(f"/proc/{proc.pid}/status"):
                    if line.startswith("VmHWM:"):
                        peak[0] = max(peak[0], int(line.split()[1]))
            except OSError:
                pass
            time.sleep(0.05)
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    out, err = proc.communicate(timeout=timeout)
    th.join(timeout=1)
    busy = stop_busy()
    return proc.returncode, out, err, time.perf_counter() - tw, (round(peak[0] / 1024.0, 1) if peak[0] else None), busy


def _fasta_md5(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def _build_generator():
    import subprocess
    src = os.path.join(ROOT, "tests", "golden", "gen_e2e_fast.cpp")
    gen = os.path.join(ROOT, "tests", "_build", "gen_e2e_fast")
    if not os.path.exists(gen) or os.path.getmtime(gen) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(gen), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-fopenmp", "-o", gen, src, "-lz"])
    return gen


def end_to_end_fast_leg(name, bam, what):
    """The `hypo` binary on a set of the C++ generator (tests/golden/gen_e2e_fast.cpp) whose golden manifest holds the md5 of the FASTA
    the REAL reference produced for it: BASELINE config C3 (100 x 1 Mbp, SAM text) and the 250 Mbp / k = 15 set (BAM)."""
    import re
    import shutil
    import subprocess
    binp = os.path.join(ROOT, "hypo_amd", "_build", "hypo")
    manp = os.path.join(ROOT, "tests", "golden", name + ".manifest.json")
    if not (os.path.exists(binp) and os.path.exists(manp)):
        return {"error": "hypo binary or golden manifest missing"}
    man = json.load(open(manp))
    a = man["args"]
    d = _scratch_dir("hypo_bench_e2e_", a["contigs"] * a["contig_len"] * (8 if bam else 45))
    try:
        gen = _build_generator()
        tg = time.perf_counter()
        rep = json.loads(subprocess.check_output([gen, d, str(a["seed"]), str(a["contigs"]), str(a["contig_len"]), str(a["k"]),
                                                  str(a["coverage"]), str(a["read_len"]), str(a["read_sub_ppm"])] +
                                                 (list(a["flags"]) if "flags" in a else (["--bam"] if bam else [])), text=True))
        tg = time.perf_counter() - tg
        want = man["generator_report"]
        if "flags" in a:
            bam = "--bam" in a["flags"]
        keys = ("contigs", "draft_bases", "reads", "solid_kmers", "fnv_draft", "fnv_bitvector") if (bam and "flags" not in a) else tuple(want.keys())
        if any(rep.get(k) != want[k] for k in keys):
            return {"error": "the generator's output differs from the golden's inputs"}
        threads = min(64, os.cpu_count() or 1)        # (128 threads were measured slower on the 2 x 64-core box: 14.0 against 12.1 s on the 1 Gbp set)
        argv = [binp] + man["command"].split()[1:]
        argv[argv.index("-t") + 1] = str(threads)
        if "our_p" in a and "-p" in argv:
            argv[argv.index("-p") + 1] = str(a["our_p"])
        if bam:
            argv[argv.index("-b") + 1] = "sr.bam"
            if "-B" in argv:
                argv[argv.index("-B") + 1] = "lr.bam"
        rc, out, err, wall, rss, busy = _run_hypo(argv, d)
        if rc != 0:
            return {"error": (out + err)[-300:]}
        m = re.search(r"Overall\. \): TIME= ([0-9.eE+-]+) sec", out)
        overall = float(m.group(1)) if m else wall
        if _fasta_md5(os.path.join(d, "hypo_draft.fasta")) != man["expected_fasta_md5"]:
            raise SystemExit(f"bench: {name}: end-to-end FASTA differs from the real reference's — refusing to report a number")
        poa = [float(x) for x in re.findall(r"POA of windows\. \): TIME= ([0-9.eE+-]+) sec", out)]
        nwin = sum(int(x) for x in re.findall(r"polished windows \(Batch \d+\): (\d+)", out))
        G = rep["draft_bases"]
        return {"mbp_per_s": round(G / 1e6 / overall, 2), "seconds": round(overall, 3), "process_wall_seconds": round(wall, 3),
                "peak_rss_mb": rss, "gpu_busy_percent_mean": busy, "host_threads": threads,
                "windows": nwin, "poa_seconds_total": round(sum(poa), 3), "contig_batches": len(poa),
                "input_generation_seconds": round(tg, 1), "alignment_file": ("BAM (BGZF, inflated in parallel by " + (re.search(r"BGZF blocks are inflated by (\w+)", out).group(1) if re.search(r"BGZF blocks are inflated by (\w+)", out) else "?") + ")") if bam else "SAM text",
                "reference": ({"seconds": man["reference_run"]["overall_seconds"], "threads": man["reference_run"]["threads"],
                               "peak_rss_mb": man["reference_run"]["peak_rss_mb"], "where": man["reference_run"]["host"] + " (not this box)"} if "reference_run" in man else
                              {"pinned_by": man.get("pinned_by")}),
                "workload": what,
                "fasta": "md5 identical to the real reference's output for these inputs"}
    except (subprocess.SubprocessError, OSError, ValueError) as ex:
        return {"error": str(ex)[:300]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def end_to_end_c3_leg():
    return end_to_end_fast_leg("e2e_c3_100m_s31", False,
                               "C3 end to end: 100 x 1 Mbp draft, 30x 150-bp reads (19.9 M records, 3.9 GB of SAM text), k = 13, -p 10, hypo binary = host pipeline + device, one run")


def end_to_end_k15_leg():
    return end_to_end_fast_leg("e2e_k15_250m_s77", True,
                               "250 Mbp at the k of C4 end to end: 250 x 1 Mbp draft, 30x 150-bp reads (49.7 M records as BAM), -s 250m -> k = 15 (128 MiB solid set, dense tiny-window shape), -p 10, one run")


def end_to_end_c4_leg():
    return end_to_end_fast_leg("e2e_c4_250m_s77", True,
                               "C4 at size end to end: ONE 250 Mbp contig, 30x 150-bp reads with coverage gaps (48.9 M records, BAM) + 40x noisy 8 kbp long reads (1.24 M records, -B, BAM), k = 15, one batch (3.7 M SHORT + ~10 k LONG windows), one run")


def end_to_end_1g_leg():
    return end_to_end_fast_leg("e2e_1g_s91", True,
                               "Row T1 at 1 Gbp end to end: 1000 x 1 Mbp draft, 30x 150-bp reads (198.8 M records as BAM), -s 1g -> k = 15, -p 10, hypo binary = host pipeline + device on ONE MI355X, one run")


def end_to_end_k17_leg():
    return end_to_end_fast_leg("e2e_k17_10m_s117", True,
                               "k = 17 against the reference: 10 x 1 Mbp draft, 30x 150-bp reads (1.99 M records as BAM), -s 3g -> k = 17 (2 GiB solid set), one run")


def end_to_end_real_leg():
    return end_to_end_fast_leg("e2e_real_5m_s131", True,
                               "non-i.i.d. inputs end to end: 5 x 1 Mbp with 15 % of the genome in tandem repeats / homopolymer runs / dispersed copies, a second haplotype (0.1 % SNPs, "
                               "0.04 % 1-base indels), 150-bp reads at 30x with 0.2 % substitutions + 0.05 % indels each way (x 5 in homopolymers), 0.2 % mis-placed reads, k = 11, -p 2; "
                               "the md5 is that of the reference compiled in place (oracle/_ref/libhyporef_arms.so) on the same files")


def value_repeat_leg(gpu, torch, dev):
    """windows/s of the POA call on the REAL windows of the non-i.i.d. 5 Mbp set, as the reference itself cut them (tests/e2e_util.py:
    realistic_window_batch), next to the headline's simulator batch; every consensus is compared with the string the reference's own
    Window::generate_consensus left in its dump."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import e2e_util as eu
        import oracle
        if not oracle.RefArms.available():
            return {"error": "oracle/_ref/libhyporef_arms.so missing (the windows are cut by the reference compiled in place)"}
        d = tempfile.mkdtemp(prefix="hypo_bench_real_")
        try:
            b, cons, man, rr = eu.realistic_window_batch(d)
        finally:
            shutil.rmtree(d, ignore_errors=True)
        off = b.slot_layout()
        dbs = [gpu.device_batch(b, off=off) for _ in range(2)]
        for _ in range(2):
            for x in dbs:
                x.run()
        cnt = [0]
        def call():
            dbs[cnt[0] % 2].run()
            cnt[0] += 1
        t = timed(call, 6, lambda: torch.cuda.synchronize(dev))
        bases, _, ln, st = dbs[1].results()
        same = bool((st == 0).all()) and all(bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes() == cons[i].encode() for i in range(b.n_windows))
        if not same:
            raise SystemExit("bench: value_repeat: device consensus differs from the reference's own — refusing to report a number")
        s = dbs[1].stats()
        na = max(s["n_alignments"], 1)
        w = b.windows
        return {"value": round(b.n_windows / t, 1), "unit": "windows/s", "ms_per_call": round(t * 1e3, 3), "windows": b.n_windows, "arms": b.n_arms,
                "mean_draft_len": round(float(w["draft_len"].mean()), 1), "windows_per_class": s["n_class"][:6], "requeued_windows": s["n_escalated"], "failed": s["n_failed"],
                "alignments": {"total": s["n_alignments"], "reused": round(s["n_reused"] / na, 4), "threaded": round(s["n_threaded"] / na, 4),
                               "scored": round(1.0 - (s["n_reused"] + s["n_threaded"]) / na, 4), "cells_scored": s["cells_scored"]},
                "gcups": round(s["dp_cells"] / t / 1e9, 1),
                "reference_same_box": {"poa_seconds": round(rr["poa_seconds"], 3), "windows_per_s": round(rr["windows"] / rr["poa_seconds"], 1), "threads": rr["threads"], "kind": "reference"},
                "parity": f"all {b.n_windows} consensus strings identical to the reference's own dump",
                "workload": "POA call only on the real windows of the non-i.i.d. 5 Mbp set (e2e_real_5m_s131: repeats, second haplotype, read indels, mis-placed reads), "
                            "cut by the reference compiled in place; two resident copies alternate"}
    except (OSError, RuntimeError, ImportError, AssertionError) as ex:
        return {"error": str(ex)[:300]}


def end_to_end_c5_leg():
    return end_to_end_fast_leg("e2e_c5_250m_s207", True,
                               "C5 slice end to end: 250 x 1 Mbp draft, 50x HiFi-like 15 kbp reads passed as the short reads (822 163 records, 3.0 GB of BAM), -s 3g -> k = 17, -c 50, -p 50, one run")


T1_3GBP_MD5_ROUND4 = "a4b0764b66b0a285dca0f630a0bbfde9"      # profiles/history/r04_t1_3gbp.json: 3000 x 1 Mbp, seed 97, -s 3g, identical for -p 50 / -p 100


def t1_reference_pin(d, out_name, n_contigs, k, run, n_pick):
    """Row T1 pinned at size: a fixed-seed sample of the run's contigs goes, with the records of the SAME BAM file, through the reference's
    own code compiled in place (oracle/_ref/libhyporef_arms.so, hyporef_fasta_bam: an independent minimal BGZF / BAM decoder hands every
    record to the reference's Alignment constructor as bam1_t; its own stage, Window::generate_consensus and operator<<(Contig) follow,
    OpenMP on all host cores in the reference's own loop shape, src/Hypo.cpp:126-268) and every FASTA record must be the bytes the `hypo`
    binary wrote.  The reference's seconds (without any file decoding: the decoder here is not its work) give the same-box baseline of
    this shape.  A difference ends the bench: no number is reported for a run that is not the reference's output."""
    try:
        import oracle
        if not oracle.RefArms.available():
            return {"error": "oracle/_ref/libhyporef_arms.so missing"}
        pick = sorted(int(x) for x in np.random.default_rng(97).choice(n_contigs, size=min(n_pick, n_contigs), replace=False))
        ref = oracle.RefArms()
        tw = time.perf_counter()
        rr = ref.fasta_file(os.path.join(d, "draft.fa"), os.path.join(d, "sr.bam"), k, os.path.join(d, "aux", "solid_kmers.bvsd"), os.path.join(d, "ref_pick.fa"), pick=pick)
        tw = time.perf_counter() - tw
        want = {}
        with open(os.path.join(d, "ref_pick.fa")) as f:
            for line in f:
                if line.startswith(">"):
                    name = line[1:].split()[0]
                else:
                    want[name] = want.get(name, "") + line.rstrip("\n")
        same, seen, name = 0, 0, None
        with open(os.path.join(d, out_name)) as f:
            for line in f:
                if line.startswith(">"):
                    name = line[1:].split()[0]
                elif name in want:
                    seen += 1
                    same += line.rstrip("\n") == want[name]
    except (OSError, RuntimeError) as ex:
        return {"error": str(ex)[:300]}
    if len(want) != len(pick) or seen != len(pick) or same != len(pick):
        raise SystemExit(f"bench: T1: {len(pick) - same} of {len(pick)} sampled contigs differ from the reference's own polish of the same BAM — refusing to report a number")
    ref_s = rr["alignment_object_seconds"] + rr["stage_seconds"] + rr["poa_seconds"] + rr["write_seconds"]
    ref_mbp, ref_wps = rr["draft_bases"] / 1e6 / ref_s, rr["windows"] / rr["poa_seconds"]
    our_wps = run["windows"] / run["poa_seconds_total"] if run["poa_seconds_total"] else None
    return {"contigs_checked": len(pick), "identical": same, "sample": f"numpy default_rng(97).choice({n_contigs}, {len(pick)}) of the run's contigs, records from the same sr.bam",
            "ref_seconds": round(ref_s, 2), "ref_threads": rr["threads"], "ref_phases": {"alignment_objects": round(rr["alignment_object_seconds"], 2), "stage": round(rr["stage_seconds"], 2),
                                                                                       "poa": round(rr["poa_seconds"], 2), "write": round(rr["write_seconds"], 2)},
            "ref_excludes": f"all file decoding (this harness's BGZF / BAM decoder took {rr['decode_seconds']:.1f} s for the whole file) and loading the solid set; wall of the call {tw:.1f} s",
            "ref_draft_bases": rr["draft_bases"], "ref_alignments": rr["alignments"], "ref_windows": rr["windows"],
            "ref_mbp_per_s": round(ref_mbp, 3), "ref_poa_windows_per_s": round(ref_wps, 1),
            "ours_mbp_per_s": run["mbp_per_s"], "ours_poa_windows_per_s": round(our_wps, 1) if our_wps else None,
            "same_box_ratio_mbp_per_s": round(run["mbp_per_s"] / ref_mbp, 1), "same_box_ratio_poa_windows_per_s": round(our_wps / ref_wps, 1) if our_wps else None,
            "what": "ours = the whole `hypo` run of all contigs (file parsing, solid set loading and output included) on this box's GPU + host; "
                    "ref = the reference's polish of the sampled contigs on this box's host cores, scaled per base / per window"}


def end_to_end_t1_leg(n_contigs, contig_len=1_000_000, batchings=(10, 50), pin_contigs=0):
    """Row T1 (north_star: 3 Gbp / 30x short reads on one GPU): `n_contigs` x 1 Mbp from the C++ generator as BAM, `-s <size>` picks k as the
    reference does (1g -> 15, 3g -> 17), the run is made once per contig-batch size in `batchings` and the FASTA of all of them must be
    identical (`-p` invariance); no reference md5 at this size (the real reference needs hours and > 100 GB for it), the same binary
    reproduces the reference's FASTA on every smaller golden."""
    import re
    import shutil
    import subprocess
    binp = os.path.join(ROOT, "hypo_amd", "_build", "hypo")
    if not os.path.exists(binp):
        return {"error": "hypo binary missing"}
    total = n_contigs * contig_len
    size_flag = f"{total // 1_000_000_000}g" if total % 1_000_000_000 == 0 else f"{total // 1_000_000}m"
    import math
    d = _scratch_dir("hypo_bench_t1_", total * 8)
    try:
        gen = _build_generator()
        # k as src/main.cpp:490-528 derives it from the size flag
        val, unit = (total // 1_000_000_000, 30) if size_flag.endswith("g") else (total // 1_000_000, 20)
        kk = (unit + int(math.ceil(math.log2(val)))) // 2
        kk = kk + 1 if kk % 2 == 0 else kk
        tg = time.perf_counter()
        rep = json.loads(subprocess.check_output([gen, d, "97", str(n_contigs), str(contig_len), str(kk), "30", "150", "2000", "--bam", "--fast-hash"], text=True))
        tg = time.perf_counter() - tg
        threads = min(64, os.cpu_count() or 1)
        runs, md5s = [], []
        for pb in batchings:
            outp = f"out_p{pb}.fa"
            argv = [binp, "-d", "draft.fa", "-r", "reads.fa", "-s", size_flag, "-c", "30", "-b", "sr.bam", "-t", str(threads), "-i", "-p", str(pb), "-o", outp]
            rc, out, err, wall, rss, busy = _run_hypo(argv, d, timeout=5000)
            if rc != 0:
                return {"error": (out + err)[-300:]}
            if f"chosen for the given genome size ({size_flag}): {kk}" not in out:
                return {"error": f"k mismatch: expected {kk}: " + out[:200]}
            m = re.search(r"Overall\. \): TIME= ([0-9.eE+-]+) sec", out)
            overall = float(m.group(1)) if m else wall
            md5s.append(_fasta_md5(os.path.join(d, outp)))
            if pb != batchings[-1]:
                os.remove(os.path.join(d, outp))
            poa = [float(x) for x in re.findall(r"POA of windows\. \): TIME= ([0-9.eE+-]+) sec", out)]
            nwin = sum(int(x) for x in re.findall(r"polished windows \(Batch \d+\): (\d+)", out))
            phases = {}
            for lab, sec in re.findall(r"RESOURCES \(\[Hypo:Hypo\]: (.*?)\. \): TIME= ([0-9.eE+-]+)", out):
                phases[lab] = round(phases.get(lab, 0.0) + float(sec), 3)
            runs.append({"p": pb, "seconds": round(overall, 2), "mbp_per_s": round(rep["draft_bases"] / 1e6 / overall, 2), "process_wall_seconds": round(wall, 2),
                         "peak_rss_mb": rss, "gpu_busy_percent_mean": busy, "windows": nwin, "poa_seconds_total": round(sum(poa), 2), "contig_batches": len(poa), "phases": phases})
        best = min(runs, key=lambda r: r["seconds"])
        pin = t1_reference_pin(d, f"out_p{batchings[-1]}.fa", n_contigs, kk, runs[-1], pin_contigs) if pin_contigs else None
        return {"reference_pin": pin, "mbp_per_s": best["mbp_per_s"], "seconds": best["seconds"], "draft_bases": rep["draft_bases"], "reads": rep["reads"], "k": kk, "size_flag": size_flag,
                "solid_kmers": rep["solid_kmers"], "peak_rss_mb": best["peak_rss_mb"], "gpu_busy_percent_mean": best["gpu_busy_percent_mean"], "windows": best["windows"],
                "runs": runs, "fasta_identical_across_batchings": len(set(md5s)) == 1, "fasta_md5": md5s[0], "host_threads": threads,
                "input_generation_seconds": round(tg, 1), "bam_bytes": os.path.getsize(os.path.join(d, "sr.bam")),
                "workload": f"T1: {n_contigs} x {contig_len // 1000} kbp draft, 30x 150-bp short reads as BAM, -s {size_flag} -> k = {kk}, hypo binary = host pipeline + device on ONE MI355X, "
                            f"-p {' and -p '.join(str(x) for x in batchings)}; seconds = the binary's Overall timer of the faster run",
                "fasta": "reference_pin: a fixed-seed sample of the contigs byte-identical to the reference's own polish of the same BAM on this box (no md5 of a whole "
                         "reference run exists at this size; the same binary matches the real reference's md5 at 5 / 10 (k = 17) / 17 / 100 / 250 / 1000 Mbp)"}
    except (subprocess.SubprocessError, OSError, ValueError) as ex:
        return {"error": str(ex)[:300]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


C3_REPLICAS = 20           # 100 Mbp of draft = 20 x the windows of the 5 Mbp run
C3_CONTIGS, C3_CONTIG_BASES, C3_K = 100, 1_000_000, 13     # -s 100m => k = 13


def timed(fn, steps, fence):
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    fence()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["auto", "c2", "c3"], default="auto",
                    help="auto: C2 (BASELINE configs[1]) on one GPU, C3 strong scaling (configs[2]) on several")
    ap.add_argument("--windows", type=int, default=N_WINDOWS, help="windows per GPU of the C2 workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-e2e-c3", action="store_true", help="skip the 100 Mbp end-to-end run (about a minute and 4 GB of scratch files)")
    ap.add_argument("--no-e2e-k15", action="store_true", help="skip the 250 Mbp / k = 15 end-to-end run (BAM input, about half a minute)")
    ap.add_argument("--no-e2e-c4", action="store_true", help="skip the C4-at-size end-to-end run (one 250 Mbp contig, short + long reads, about a minute)")
    ap.add_argument("--no-e2e-1g", action="store_true", help="(kept for old command lines: the 1 Gbp run is opt-in now)")
    ap.add_argument("--e2e-1g", action="store_true", help="also run the 1 Gbp end-to-end set (BAM input, about a minute of input generation + half a minute; md5 of the CMake-built "
                                                           "reference binary; since round 6 the 3 Gbp run is pinned against the reference compiled in place, which made this leg redundant)")
    ap.add_argument("--no-e2e-c5", action="store_true", help="skip the 250 Mbp C5 slice (15 kbp reads as -b, k = 17; about 40 s of input generation)")
    ap.add_argument("--no-e2e-k17", action="store_true", help="skip the 10 Mbp run at -s 3g (k = 17, the real reference's md5)")
    ap.add_argument("--t1-contigs", type=int, default=int(os.environ.get("HYPO_BENCH_T1_CONTIGS", "3000")),
                    help="also run row T1 end to end on this many 1 Mbp contigs (3000 = the north star's 3 Gbp; needs ~20 GB of /dev/shm and a few minutes)")
    ap.add_argument("--t1-pin", type=int, default=int(os.environ.get("HYPO_BENCH_T1_PIN", "150")),
                    help="contigs of the T1 run (fixed-seed sample) that also go through the reference compiled in place and are compared record by record (0 = none)")
    ap.add_argument("--no-extras", action="store_true", help="skip value_at_0p5pct / value_at_1pct / value_dense / value_c4mix and host_api")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hypo_amd import capi, sim
    from hypo_amd import dist as hd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    workload = args.workload if args.workload != "auto" else ("c2" if world == 1 else "c3")
    strong = workload == "c3"
    # Debug knobs for a box with fewer GPUs than ranks (never set by the driver): all ranks on device 0 and a gloo
    # group exercise the whole N>1 code path except RCCL itself.
    share = os.environ.get("HYPO_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HYPO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    gpu = capi.HypoGpu(dev_index)

    # ---- synthetic workload, resident in HBM ----------------------------------------------------------
    imbalance = None
    if strong:
        # the same batch on every rank (fixed seed); each rank keeps its cost-balanced contiguous range
        n_total = int(os.environ.get("HYPO_BENCH_C3_WINDOWS", C3_REPLICAS * N_WINDOWS))
        whole = sim.window_batch(n_total, seed=3000)
        costs = hd.window_costs(whole.windows, whole.arm_len)
        ranges = hd.shard_contiguous(costs, world)
        b, e = ranges[rank]
        batch = hd.take_windows(whole, b, e, compact=True)
        per = np.array([costs[x:y].sum() for x, y in ranges])
        imbalance = {"planned_cost_max_over_mean": round(float(per.max() / per.mean()), 4)}
        del whole, costs
        contig_bases, k = C3_CONTIG_BASES, C3_K
        my_contigs = [c for c in range(C3_CONTIGS) if c % world == rank]
    else:
        batch = sim.window_batch(args.windows, seed=1000 + rank)
        contig_bases, k = CONTIG_BASES, K
        my_contigs = [rank]
    scans = []
    bits = None
    for c in my_contigs:                                  # one bit set per run, uploaded once per rank
        codes, packed4 = sim.random_contig(contig_bases, seed=2000 + c, n_frac=0.0)
        if bits is None:
            bits = sim.solid_bitset(codes, k) if not strong else None
            if bits is None:                              # C3: a random set of the right size and density (the scan's cost does not depend on which k-mers are solid)
                rb = np.random.default_rng(7)
                bits = rb.integers(0, 1 << 63, size=(1 << (2 * k)) // 64, dtype=np.int64).view(np.uint64) & \
                    rb.integers(0, 1 << 63, size=(1 << (2 * k)) // 64, dtype=np.int64).view(np.uint64)
            first = (packed4, codes.size)
        scans.append(gpu.device_scan(packed4, contig_bases, k, bits, kids_cap=contig_bases // 2))
    for s in scans[1:]:
        s.bits = scans[0].bits                            # one device copy of the set
    off = batch.slot_layout()
    db = gpu.device_batch(batch, off=off)
    # C2: a SECOND resident batch of the same shape (other seed) alternates with the first in the timed loop, so that the grid hints
    # a call takes from the call before it (poa_kernel.hip: plan history) come from a different batch, as in a real run
    db_alt = None
    if not strong and os.environ.get("HYPO_BENCH_ALTERNATE", "1") == "1":
        batch_alt = sim.window_batch(args.windows, seed=5000 + rank)
        db_alt = gpu.device_batch(batch_alt, off=batch_alt.slot_layout())
    ds = scans[0]
    packed4 = first[0]
    n_w = batch.n_windows

    # exchange step: sizes agreed once (all_reduce MAX), then one fixed-size all-gather per batch from preallocated buffers
    if world > 1:
        max_bytes, max_windows = hd.agree_sizes(int(off[-1]), n_w, dev)
        exchange = hd.ConsensusExchange(max_bytes, max_windows, dev)

    step_no = [0]

    def step():
        for s in scans:
            s.run()
        cur = db if (db_alt is None or step_no[0] % 2 == 0) else db_alt
        step_no[0] += 1
        cur.run()
        if world > 1:
            exchange.gather(db.bases, db.len[:n_w])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    gpu.profile_begin(min(256, (1 + len(scans)) * args.steps))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    my_dt = dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof = gpu.profile_read()
    windows_timed = n_w * args.steps if db_alt is None else n_w * ((args.steps + 1) // 2) + db_alt.host.n_windows * (args.steps // 2)
    if db_alt is not None and args.steps % 2 == 0:
        db.run()                                           # the checks below look at the FIRST batch: make it the most recent call
        torch.cuda.synchronize(dev)
    stats = db.stats()
    bases, _, ln, st = db.results()
    words, kids, rank_dir, n_solid = ds.results()

    # ---- parity spot check (outside the timed region): the HIP results against the oracle -----------
    parity = None
    if not args.no_check and rank == 0:
        import oracle
        orc = oracle.Oracle()
        sub = sim.window_batch(4000, seed=77)
        sdb = gpu.device_batch(sub)
        sdb.run()
        sb, soff, sln, sst = sdb.results()
        ob, _, oln, ost, _, _ = orc.poa_batch_raw(sub, off=soff)
        ok = bool((sst == ost).all() and (sln == oln).all())
        if ok:
            for i in range(sub.n_windows):
                a, l = int(soff[i]), int(sln[i])
                if not (sb[a:a + l] == ob[a:a + l]).all():
                    ok = False
                    break
        ow, okids, orank, ons = orc.solid_scan(packed4, contig_bases, k, bits, kids_cap=contig_bases // 2)
        ok = ok and ons == n_solid and bool((ow == words).all()) and bool((okids == kids).all())
        parity = "bit-exact vs oracle (4000 windows + full scan)" if ok else "MISMATCH"
        if ok and db_alt is not None:                          # the second batch of the timed loop, whole, against the oracle
            ab, aoff, aln, ast = db_alt.results()
            xb, _, xln, xst, _, _ = orc.poa_batch_raw(db_alt.host, off=aoff)
            ok = bool((ast == xst).all() and (aln == xln).all())
            if ok:
                o64, l64 = aoff[:-1].astype(np.int64), aln.astype(np.int64)
                idx = np.repeat(o64, l64) + (np.arange(int(l64.sum()), dtype=np.int64) - np.repeat(np.cumsum(l64) - l64, l64))
                ok = bool((ab[idx] == xb[idx]).all())
            parity += f"; alternate timed batch bit-exact vs oracle ({db_alt.host.n_windows} windows)" if ok else "; alternate batch MISMATCH"
        if not ok:
            raise SystemExit("bench: HIP results differ from the oracle — refusing to report a number")

    # ---- roofline of the dominant kernel (HIP events on the kernels' stream) -----------------------------
    poa_calls = [p for p in prof if len(p) == 8]             # plan, 6 size-class kernels, whole call
    scan_calls = [p for p in prof if len(p) == 3]
    roofline = None
    extra = {}
    if poa_calls:
        ms = np.array(poa_calls, dtype=np.float64)             # [calls, 1 + classes]
        cls_ms = ms[:, 1:-1].mean(axis=0)                      # the size-class kernels
        # (a class that finished no window does not count: class 3's polling launch lives as long as the classes that feed it)
        busy = np.array([1.0 if stats["n_class"][c] > 0 else 0.0 for c in range(cls_ms.size)])
        dom = int(np.argmax(cls_ms * busy))
        alg = float(stats["alg_bytes"][dom])
        achieved = alg / (cls_ms[dom] * 1e-3) / 1e9 if cls_ms[dom] > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": f"poa_class_kernel<class {dom}>", "achieved": round(achieved, 4),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 8),
                    "traffic": None, "kernel_ms": round(float(cls_ms[dom]), 4),
                    "algorithmic_bytes_per_launch": int(alg), "windows_per_launch": int(stats["n_class"][dom])}
        tr = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tr):
            try:
                j = json.load(open(tr)).get(roofline["kernel"], {})      # one entry per size-class kernel (profiles/make_traffic_json.py)
                if j.get("windows") == roofline["windows_per_launch"]:
                    roofline["traffic"] = j.get("hbm_bytes_per_launch")
                    # VALU issue of the same launch from the PMC pass: instructions, and the share of the chip's VALU issue
                    # slots (256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD, MI355X_MICROARCH.md) they
                    # fill over the measured kernel time
                    if j.get("valu_insts"):
                        slots = 256 * 4 * (float(cls_ms[dom]) * 1e-3) * 2.4e9 / 2.0
                        roofline["valu"] = {"insts": int(j["valu_insts"]), "salu_insts": int(j.get("salu_insts", 0)),
                                            "issue_frac": round(j["valu_insts"] / slots, 4), "source": j.get("source", "profiles/")}
            except Exception:
                pass
        # every size-class kernel of the step (classes 0-2 run SIDE BY SIDE on three streams: which of them is the longest changes from run
        # to run, so "the dominant kernel" above is whichever took longest this time) and the step as a whole
        kern = []
        for c in range(cls_ms.size):
            if busy[c] == 0.0:
                continue
            a_c, ms_c = float(stats["alg_bytes"][c]), float(cls_ms[c])
            e = {"name": f"poa_class_kernel<class {c}>", "ms": round(ms_c, 4), "windows": int(stats["n_class"][c]), "algorithmic_bytes": int(a_c),
                 "achieved_gbs": round(a_c / (ms_c * 1e-3) / 1e9, 3) if ms_c > 0 else None, "traffic": None}
            try:
                j = json.load(open(tr)).get(e["name"], {}) if os.path.exists(tr) else {}
                if j.get("windows") == e["windows"]:
                    e["traffic"] = j.get("hbm_bytes_per_launch")
                    e["valu_insts"], e["salu_insts"] = j.get("valu_insts"), j.get("salu_insts")
            except Exception:
                pass
            kern.append(e)
        roofline["kernels"] = kern
        roofline["concurrent"] = "classes 0-2 run concurrently on three streams (fixed wave shares per CU); `kernel` = the longest of them in THIS run"
        poa_ms_total = float(ms[:, -1].mean())                    # whole POA call (the class kernels overlap)
        extra["poa_kernels_ms"] = [round(float(x), 4) for x in ms.mean(axis=0)]
        extra["gcups"] = round(stats["dp_cells"] / (poa_ms_total * 1e-3) / 1e9, 3)     # reference-equivalent cells
        # cells the device really pushed through row loops: score rows + one-bit threading rows (failed attempts included)
        extra["gcups_executed"] = round((stats["cells_scored"] + stats["cells_threaded"]) / (poa_ms_total * 1e-3) / 1e9, 3)
        na = max(stats["n_alignments"], 1)
        extra["alignments"] = {"total": stats["n_alignments"], "reused": round(stats["n_reused"] / na, 4),
                               "threaded": round(stats["n_threaded"] / na, 4),
                               "scored": round(1.0 - (stats["n_reused"] + stats["n_threaded"]) / na, 4),
                               "cells_scored": stats["cells_scored"], "cells_threaded": stats["cells_threaded"]}
        extra["windows_per_class"] = stats["n_class"]
    if scan_calls:
        sm = np.array(scan_calls, dtype=np.float64).mean(axis=0)
        # A_scan = ceil(L/2) + ceil(L/8) + 8*n_solid + min(4^k/8, 32*(L-k+1))
        a_scan = (contig_bases + 1) // 2 + (contig_bases + 7) // 8 + 8 * n_solid + min((1 << (2 * k)) // 8, 32 * (contig_bases - k + 1))
        extra["scan_kernels_ms"] = [round(float(x), 4) for x in sm]
        extra["scan_gbs"] = round(a_scan / (float(sm.sum()) * 1e-3) / 1e9, 2)
        if roofline is not None:                                   # the step: every kernel's algorithmic bytes over the step's wall time
            step_alg = float(sum(stats["alg_bytes"])) + float(a_scan) * len(scans)
            step_s = dt / args.steps
            step_traffic = None
            if all(kk.get("traffic") for kk in roofline.get("kernels", [])) and roofline.get("kernels"):
                step_traffic = int(sum(kk["traffic"] for kk in roofline["kernels"]))
            roofline["step"] = {"algorithmic_bytes": int(step_alg), "ms": round(step_s * 1e3, 4), "achieved_gbs": round(step_alg / step_s / 1e9, 3),
                                "frac": round(step_alg / step_s / 1e9 / HBM_PEAK_GBS, 8), "poa_traffic": step_traffic,
                                "what": "POA kernels (all size classes) + solid-kmer scan(s) of one step: sum of algorithmic bytes / ms_per_step"}

    # ---- the same POA call on noisier reads and through the host-pointer entry point (N = 1 only, a few calls each) ----
    if rank == 0 and world == 1 and not args.no_extras:
        for key, sub in (("value_at_0p5pct", 0.005), ("value_at_1pct", 0.01)):
            # two batches of that error rate alternate, like the headline's
            ndbs = [gpu.device_batch(sim.window_batch(args.windows, seed=sd, read_sub=sub)) for sd in (1000, 5000)]
            for _ in range(2):
                for d in ndbs:
                    d.run()
            cnt = [0]
            def noisy():
                ndbs[cnt[0] % 2].run()
                cnt[0] += 1
            t1 = timed(noisy, 6, lambda: torch.cuda.synchronize(dev))
            nst = ndbs[1].stats()
            extra[key] = {"value": round(args.windows / t1, 1), "unit": "windows/s", "ms_per_call": round(t1 * 1e3, 3),
                          "requeued_windows": nst["n_escalated"], "carried_graphs": nst.get("n_carried"), "failed": nst["n_failed"],
                          "workload": f"the C2 batch shape with {sub * 100:g} % substitutions in the reads instead of 0.2 %, two batches alternating, POA call only"}
            del ndbs
        # other window shapes of BASELINE's configs, POA call only (profiles/dense_rate.py, profiles/c4_rate.py)
        rng = np.random.default_rng(3)
        nd = 1000000
        wl = rng.choice([3, 5, 8, 12, 16, 24, 32, 48, 64, 99], size=nd, p=[.15, .15, .15, .13, .13, .1, .1, .05, .03, .01])
        shapes = np.stack([wl, rng.integers(3, 45, size=nd), np.zeros(nd, np.int64), np.zeros(nd, np.int64), np.zeros(nd, np.int64)], axis=1)
        ddb = gpu.device_batch(sim.window_batch(nd, seed=9, shapes=shapes, read_sub=0.002))
        for _ in range(3):          # one at a time: a call sizes its launches from the last FINISHED call of the context, as in a hypo run
            ddb.run()
            torch.cuda.synchronize(dev)
        t4 = timed(ddb.run, 4, lambda: torch.cuda.synchronize(dev))
        extra["value_dense"] = {"value": round(nd / t4, 1), "unit": "windows/s", "ms_per_call": round(t4 * 1e3, 3), "failed": ddb.stats()["n_failed"],
                                "workload": "dense short-read shape of C4/C5 (1 M tiny windows, 45 % <= 8 bp, 3-44 arms), POA call only"}
        del ddb
        cdb = gpu.device_batch(sim.c4_batch(400000, 8000, seed=404))
        cdb.run()
        t5 = timed(cdb.run, 2, lambda: torch.cuda.synchronize(dev))
        extra["value_c4mix"] = {"value": round(408000 / t5, 1), "unit": "windows/s", "ms_per_call": round(t5 * 1e3, 3), "failed": cdb.stats()["n_failed"],
                                "workload": "C4 window mix: 400 000 C1-shaped SHORT + 8 000 LONG windows (120-500 bp, 12-45 noisy long-read arms) in one batch, POA call only"}
        del cdb
        # the window-level grid — length x arms x arm error, 60 / 20 / 20 internal / prefix / suffix arms — POA call only:
        # windows/s, the TRUE cell rate of the score-row loop (cells that went through rows_pk / the int32 rows per second of the call), and
        # the share of alignments that needed no rows.  The 8 % column is where every alignment is scored.
        if os.environ.get("HYPO_BENCH_GRID", "1") == "1":
            grid = []
            for err in (0.005, 0.08):
                for length in (8, 16, 32, 64, 100, 200):
                    for arms in (6, 12, 30, 50):
                        nwin = max(4000, min(60000, 24_000_000 // (length * arms)))
                        gdb = gpu.device_batch(sim.grid_batch(length, arms, nwin, err, seed=11))
                        for _ in range(2):
                            gdb.run()
                            torch.cuda.synchronize(dev)
                        tg = timed(gdb.run, 3, lambda: torch.cuda.synchronize(dev))
                        gs = gdb.stats()
                        gna = max(gs["n_alignments"], 1)
                        grid.append({"len": length, "arms": arms, "err": err, "windows": nwin, "windows_per_s": round(nwin / tg, 1), "ms": round(tg * 1e3, 3),
                                     "gcups_reference_cells": round(gs["dp_cells"] / tg / 1e9, 1), "gcups_scored_rows": round(gs["cells_scored"] / tg / 1e9, 2),
                                     "scored_share": round(1.0 - (gs["n_reused"] + gs["n_threaded"]) / gna, 4), "classes": gs["n_class"][:4], "requeued": gs["n_escalated"], "failed": gs["n_failed"]})
                        del gdb
            extra["grid"] = {"cells": grid, "what": "SURVEY 8(d) grid, POA call only, one resident batch per cell (3 timed calls); gcups_scored_rows = cells that went through the score rows / call time "
                                                    "(the call also threads, sorts and builds consensus: a lower bound of the row loop's own rate)"}
        if os.environ.get("HYPO_BENCH_VALUE_REPEAT", "1") == "1":
            extra["value_repeat"] = value_repeat_leg(gpu, torch, dev)
        # the N > 1 workload on ONE GPU: the like-for-like base of the scaling curve (BASELINE configs[2]: the same 1.94 M windows,
        # the same 100 scans of 1 Mbp at k = 13, the same resident-input protocol; only the all-gather has nobody to talk to)
        if os.environ.get("HYPO_BENCH_VALUE_C3", "1") == "1":
            n_total = int(os.environ.get("HYPO_BENCH_C3_WINDOWS", C3_REPLICAS * N_WINDOWS))
            whole = sim.window_batch(n_total, seed=3000)
            c3db = gpu.device_batch(whole, off=whole.slot_layout())
            rb3 = np.random.default_rng(7)
            bits3 = rb3.integers(0, 1 << 63, size=(1 << (2 * C3_K)) // 64, dtype=np.int64).view(np.uint64) & \
                rb3.integers(0, 1 << 63, size=(1 << (2 * C3_K)) // 64, dtype=np.int64).view(np.uint64)
            scans3 = []
            for c in range(C3_CONTIGS):
                _, p4 = sim.random_contig(C3_CONTIG_BASES, seed=2000 + c, n_frac=0.0)
                scans3.append(gpu.device_scan(p4, C3_CONTIG_BASES, C3_K, bits3, kids_cap=C3_CONTIG_BASES // 2))
            for s3 in scans3[1:]:
                s3.bits = scans3[0].bits
            def c3_step():
                for s3 in scans3:
                    s3.run()
                c3db.run()
            c3_step()
            torch.cuda.synchronize(dev)
            t6 = timed(c3_step, 3, lambda: torch.cuda.synchronize(dev))
            c3st = c3db.stats()
            extra["value_c3"] = {"value": round(n_total / t6, 1), "unit": "windows/s", "ms_per_step": round(t6 * 1e3, 3), "steps": 3, "failed": c3st["n_failed"],
                                 "workload": f"C3 on one GPU, exactly the step `--gpus N` runs for N > 1 (strong scaling): ONE batch of {n_total} C1-shaped windows "
                                             f"+ the solid-kmer scan of {C3_CONTIGS} x 1 Mbp (k = 13), inputs resident; divide the N-GPU `value` by this for the speed-up"}
            del c3db, scans3, whole
        hoff = batch.slot_layout()
        for _ in range(2):
            gpu.poa_batch(batch, off=hoff)
        t2 = timed(lambda: gpu.poa_batch(batch, off=hoff), 5, lambda: None)
        extra["host_api"] = {"value": round(n_w / t2, 1), "unit": "windows/s", "ms_per_call": round(t2 * 1e3, 3),
                             "note": "hypo_gpu_poa_batch with host pointers (pageable memory): H2D of the batch + kernels + D2H of the consensus, one call at a time, PCIe-inclusive"}
        # the same through hypo_gpu_poa_batch_begin / _end with two batches in flight: page-locked buffers, no arm_off upload
        # (the arms lie back to back), batch i + 1 is uploaded while batch i computes
        from hypo_amd.batch import HostBatch
        pin = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).pin_memory().numpy()
        pb = HostBatch(pin(batch.windows).view(batch.windows.dtype), pin(batch.draft4), batch.arm_off,
                       pin(batch.arm_len).view(np.uint32), pin(batch.arms2))
        poff = pin(hoff).view(np.uint64)
        outs = [(torch.zeros(int(hoff[-1]) + 16, dtype=torch.uint8).pin_memory().numpy(), torch.zeros(n_w * 4, dtype=torch.uint8).pin_memory().numpy().view(np.uint32),
                 torch.zeros(n_w, dtype=torch.uint8).pin_memory().numpy()) for _ in range(2)]
        def pipelined(calls):
            pending = []
            for i in range(calls):
                ob, ol, os_ = outs[i % 2]
                if len(pending) == 2:
                    gpu.poa_batch_end(pending.pop(0)[0])
                pending.append(gpu.poa_batch_begin(pb, poff, ob, ol, os_, no_arm_off=True))
            while pending:
                gpu.poa_batch_end(pending.pop(0)[0])
        pipelined(3)
        t0 = time.perf_counter()
        pipelined(8)
        t3 = (time.perf_counter() - t0) / 8
        same = bool((outs[1][1] == ln).all() and (outs[1][2] == st).all())
        if same:                                               # ... and the consensus bytes
            o64, l64 = hoff[:-1].astype(np.int64), ln.astype(np.int64)
            idx = np.repeat(o64, l64) + (np.arange(int(l64.sum()), dtype=np.int64) - np.repeat(np.cumsum(l64) - l64, l64))
            same = bool((outs[1][0][idx] == bases[idx]).all())
        extra["host_api_pipelined"] = {"value": round(n_w / t3, 1), "unit": "windows/s", "ms_per_call": round(t3 * 1e3, 3), "results_equal": same,
                                       "note": "hypo_gpu_poa_batch_begin/_end, two batches in flight, page-locked host buffers, arm offsets computed on the device"}

    # ---- CPU baseline: the bit-exact port of the reference's OpenMP/spoa path on this box -----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        orc = oracle.Oracle()
        thr = orc.num_threads()
        cb, coff, cln, cst, cbases = batch, off, ln, st, bases
        sample_note = f"the same {n_w}-window batch"
        if n_w > 200000:                                   # C3 on one GPU: a bounded sample of the job for the CPU legs
            cb = hd.take_windows(batch, 0, N_WINDOWS, compact=True)
            coff = off[:N_WINDOWS + 1]
            cln, cst = ln[:N_WINDOWS], st[:N_WINDOWS]
            sample_note = f"the first {N_WINDOWS} windows of the batch"
        cn = cb.n_windows
        best = None
        for _ in range(3):
            c0 = time.perf_counter()
            orc.poa_batch_raw(cb, off=coff, n_threads=thr)
            c1 = time.perf_counter() - c0
            best = c1 if best is None or c1 < best else best
        cpu = {"value": round(cn / best, 1), "unit": "windows/s", "cores": thr, "kind": "port",
               "sample": f"{sample_note}, POA only, OpenMP schedule(static,1), best of 3"}
        # The real reference classes (hypo::Window + its spoa, compiled from /root/reference in the build container into
        # oracle/_ref/libhyporef.so, which travels prebuilt): the reference's own POA loop on the same batch.  When it is
        # there it is the baseline of record (kind "reference"), the port's rate stays beside it, and the device results of
        # the timed batch are compared with the reference's byte for byte.
        if oracle.Ref.available():
            try:
                ref = oracle.Ref()
                rbest, rout = None, None
                for _ in range(2):
                    rb, _, rln, rst, sec = ref.poa_batch_raw(cb, off=coff, n_threads=thr)
                    if rbest is None or sec < rbest:
                        rbest, rout = sec, (rb, rln, rst)
                same = bool((rout[1] == cln).all() and (rout[2] == cst).all())
                if same:
                    o64, l64 = coff[:-1].astype(np.int64), cln.astype(np.int64)
                    idx = np.repeat(o64, l64) + (np.arange(int(l64.sum()), dtype=np.int64) - np.repeat(np.cumsum(l64) - l64, l64))
                    same = bool((rout[0][idx] == cbases[idx]).all())
                if not same:
                    raise SystemExit("bench: HIP results differ from the real reference — refusing to report a number")
                extra["cpu_port"] = cpu
                cpu = {"value": round(cn / rbest, 1), "unit": "windows/s", "cores": thr, "kind": "reference",
                       "sample": f"{sample_note} through the reference's own Window::generate_consensus loop "
                                 "(OpenMP schedule(static,1), consensus loop only, best of 2)"}
                parity = (parity or "") + f"; timed batch bit-exact vs the real reference classes ({cn} windows)"
            except (OSError, RuntimeError) as ex:            # stale or unloadable prebuilt library: keep the port
                extra["cpu_reference_error"] = str(ex)[:200]

    # ---- end-to-end leg (the second half of BASELINE's metric): the `hypo` binary of this repo — host pipeline + device —
    # on the 5 Mbp / 30x C2 set regenerated by the committed generator (~20 s of Python); its FASTA must have the md5 the REAL
    # reference produced for these inputs (tests/golden/e2e_5m_s11.manifest.json).  Wall time = the run's own "Overall" timer, like the reference's.
    e2e = None
    e2e_c3 = None
    e2e_k15 = None
    e2e_c4 = None
    e2e_t1 = None
    e2e_1g = None
    e2e_k17 = None
    e2e_c5 = None
    e2e_real = None
    if rank == 0 and world == 1 and not args.no_e2e and not strong:
        e2e = end_to_end_leg()
        if not args.no_e2e_c3:
            e2e_c3 = end_to_end_c3_leg()
        if not args.no_e2e_k15:
            e2e_k15 = end_to_end_k15_leg()
        if not args.no_e2e_c4:
            e2e_c4 = end_to_end_c4_leg()
        e2e_real = end_to_end_real_leg()                   # non-i.i.d. genome, second haplotype, read indels (round 6), md5 of the reference compiled in place
        if not args.no_e2e_k17:
            e2e_k17 = end_to_end_k17_leg()                 # k = 17 (-s 3g) pinned to the real reference on 10 Mbp
        if not args.no_e2e_c5:
            e2e_c5 = end_to_end_c5_leg()                   # BASELINE config C5 as a workload: 15 kbp reads as -b at k = 17, 250 Mbp, the real reference's md5
        if args.e2e_1g and not args.no_e2e_1g:
            e2e_1g = end_to_end_1g_leg()                   # 1 Gbp with the real reference's md5
        if args.t1_contigs > 0:
            # row T1 at size: 3 Gbp / 30x short reads, -s 3g -> k = 17, one -p 50 run (about two minutes of input generation + the run)
            # (one -p 50 run; the pin: --t1-pin contigs of it against the reference compiled in place, on this box)
            e2e_t1 = end_to_end_t1_leg(args.t1_contigs, batchings=(50,), pin_contigs=args.t1_pin)
            if e2e_t1 and "fasta_md5" in e2e_t1 and args.t1_contigs == 3000:
                # recorded, not enforced: the md5 of this repo's own round-4 runs of the same generator arguments
                # (profiles/history/r04_t1_3gbp.json, -p 50 and -p 100) — the check that counts is reference_pin above
                e2e_t1["fasta_same_as_round4_runs"] = e2e_t1["fasta_md5"] == T1_3GBP_MD5_ROUND4
                e2e_t1["batchings_run"] = [50]

    # ---- N > 1: what the all-gather delivered, re-assembled in global window order, against the oracle on the WHOLE batch (opt-in:
    # HYPO_BENCH_CHECK_GATHER=1; tests/test_gpu_distributed.py runs it with every rank on device 0) -----------------------------------
    parity_gathered = None
    if strong and world > 1 and os.environ.get("HYPO_BENCH_CHECK_GATHER") == "1":
        gb, gl = exchange.gather(db.bases, db.len[:n_w])
        torch.cuda.synchronize(dev)
        if rank == 0:
            import oracle
            n_total = int(os.environ.get("HYPO_BENCH_C3_WINDOWS", C3_REPLICAS * N_WINDOWS))
            whole = sim.window_batch(n_total, seed=3000)
            rngs = hd.shard_contiguous(hd.window_costs(whole.windows, whole.arm_len), world)
            offs = [hd.take_windows(whole, b0, e0, compact=True).slot_layout() for b0, e0 in rngs]
            got = hd.reassemble(gb.cpu().numpy(), gl.cpu().numpy(), offs, rngs)
            want, wst, _, _ = oracle.Oracle().poa_batch(whole)
            if len(got) != n_total or got != list(want):
                raise SystemExit("bench: the gathered consensus of the sharded batch differs from the oracle's — refusing to report a number")
            parity_gathered = f"all-gathered consensus of {n_total} windows from {world} ranks, re-assembled in window order, bit-exact vs oracle"
        dist.barrier()

    if strong and world > 1:                               # per-rank imbalance of the measured step
        tt = torch.tensor([my_dt], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = np.array([float(x.item()) for x in allt])
        imbalance["step_time_max_over_mean"] = round(float(per_rank.max() / per_rank.mean()), 4)
    if world > 1:
        nt = torch.tensor([n_w], dtype=torch.int64, device=dev)
        dist.all_reduce(nt)
        total_per_step = int(nt.item())
    else:
        total_per_step = n_w
    value = total_per_step * args.steps / dt
    total_bases = contig_bases * (C3_CONTIGS if strong else world)
    if rank == 0:
        if strong:
            wl = (f"C3: synthetic 100 Mbp contig set ({C3_CONTIGS} x 1 Mbp, k=13), 30x 150-bp short reads — ONE batch of {total_per_step} "
                  f"C1-shaped windows cut into {world} cost-balanced contiguous range(s), solid-kmer scan of every contig + POA + "
                  "all-gather of the consensus")
        else:
            wl = ("C2: E. coli-sized 5 Mbp draft, 30x 150-bp short reads, k=11 — solid-kmer scan + POA "
                  f"of {n_w} C1-shaped windows per GPU (mean 38.5 bp, 18.7 arms)" +
                  ("; two resident batches (different seeds) alternate from step to step" if db_alt is not None else ""))
        out = {
            "metric": "polished windows/sec (whole node)", "value": round(value, 1), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",   # exact int16 score rows (guarded; int32 in the catch-all class)
            "config": {"workload": wl, "windows_total": total_per_step, "windows_this_rank": n_w, "arms_this_rank": batch.n_arms,
                       "contig_bases": total_bases, "k": k,
                       "parallelism": f"window sharding x{world}" + (" + RCCL all-gather of consensus" if world > 1 else "")},
            "mbp_per_s": round(total_bases * args.steps / dt / 1e6, 2),
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "e2e": e2e, "e2e_c3": e2e_c3, "e2e_k15_250m": e2e_k15, "e2e_c4_250m": e2e_c4, "e2e_k17_10m": e2e_k17, "e2e_real_5m": e2e_real, "e2e_c5_slice": e2e_c5, "e2e_1g": e2e_1g, "e2e_t1": e2e_t1,
        }
        if imbalance:
            out["imbalance"] = imbalance
        if parity_gathered:
            out["parity_gathered"] = parity_gathered
        out.update(extra)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
