"""Shared driver of the end-to-end goldens (tests/golden/e2e_*): regenerates the inputs with the committed
deterministic generator, checks their md5 against the manifest, runs the `hypo` binary of this repo with the
reference's own command line and compares the polished FASTA and the per-region digests with what the REAL reference
produced (tests/golden/make_e2e_golden.py).  `device="shim"` runs the host pipeline over tests/shim (CPU oracle behind
the C-ABI, host-logic check only); `device="gpu"` runs it over the real libhypo_gpu.so."""
import gzip
import hashlib
import importlib.util
import json
import os
import shlex
import subprocess
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
BIN = os.path.join(ROOT, "hypo_amd", "_build", "hypo")
SHIM_DIR = os.path.join(HERE, "_build", "shim")
CASES = ["e2e_20k_s1", "e2e_200k_long_s3", "e2e_200k_k9_s5", "e2e_100k_k7_s7", "e2e_5ctg_long_s21", "e2e_messy_s102", "e2e_messy_s103"]


def _md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def build_binary():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "hypo_amd", "csrc")], check=True)
    return BIN


def build_shim():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "shim")], check=True)
    return SHIM_DIR


def make_inputs(name, outdir):
    man = json.load(open(os.path.join(GOLD, name + ".manifest.json")))
    spec = importlib.util.spec_from_file_location("gen_e2e", os.path.join(GOLD, "gen_e2e.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    a = man["args"]
    if "messy" in a:
        gen.generate_messy(str(outdir), a["messy"])
    elif a.get("contigs", 1) > 1:
        gen.generate_multi(str(outdir), a["seed"], a["G"], a["long"], a["k"], a["contigs"])
    else:
        kw = {k: a[k] for k in ("read_len", "read_sub") if k in a}            # HiFi-like reads of the --ccs-windows set
        gen.generate(str(outdir), a["seed"], a["G"], a["long"], a["k"], **kw)
    for f, want in man["inputs_md5"].items():
        assert _md5(os.path.join(outdir, f)) == want, f"{name}: regenerated {f} differs from the golden's input"
    return man


def run_case(name, outdir, device, threads=4, extra_env=None, as_bam=False, extra_args=()):
    man = make_inputs(name, outdir)
    argv = shlex.split(man["command"])
    argv[0] = BIN
    if as_bam:                                      # same records as BAM (BGZF): tests/bam_util.py
        import bam_util
        for flag, nm in (("-b", "i"), ("-B", "C")):
            if flag in argv:
                i = argv.index(flag) + 1
                bam = os.path.splitext(argv[i])[0] + ".bam"
                bam_util.sam_to_bam(os.path.join(str(outdir), argv[i]), os.path.join(str(outdir), bam), nm_type=nm, extra_tags=True)
                os.remove(os.path.join(str(outdir), argv[i]))
                argv[i] = bam
    argv[argv.index("-t") + 1] = str(threads)
    argv += list(extra_args)
    env = dict(os.environ)
    env["HYPO_REGION_DUMP"] = os.path.join(str(outdir), "regions.tsv")
    if device == "shim":
        env["LD_LIBRARY_PATH"] = SHIM_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    else:
        env.setdefault("HYPO_REQUIRE_DEVICE", "1")      # a stage that quietly ran in the host loops fails the run (hypo --require-device)
    env.update(extra_env or {})
    p = subprocess.run(argv, cwd=str(outdir), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    used_shim = "oracle_device_shim" in p.stderr
    assert used_shim == (device == "shim"), "wrong device library behind the C-ABI"
    return man, p


def check_outputs(name, outdir, man):
    got = open(os.path.join(str(outdir), "hypo_draft.fasta"), "rb").read()
    if not os.path.exists(os.path.join(GOLD, name + ".expected.fa.gz")):      # full-size set: md5 only
        assert hashlib.md5(got).hexdigest() == man["expected_fasta_md5"], f"{name}: polished FASTA differs from the reference's"
        return sum(1 for _ in open(os.path.join(str(outdir), "regions.tsv")))
    want = gzip.open(os.path.join(GOLD, name + ".expected.fa.gz")).read()
    regions = json.load(gzip.open(os.path.join(GOLD, name + ".regions.json.gz")))
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(str(outdir), "regions.tsv"))]
    assert len(rows) == len(regions), f"{name}: {len(rows)} regions, reference has {len(regions)}"
    for r, g in zip(rows, regions):
        beg, end, typ = int(r[1]), int(r[2]) - 1, r[3]
        assert [beg, end, typ] == g[:3], f"{name}: region {r[:4]} vs reference {g[:3]}"
        if typ not in ("SR", "MSR"):
            counts = [int(x) for x in r[4:8]]
            assert counts == g[3:7], f"{name}: arms of window {beg}-{end}: {counts} vs reference {g[3:7]}"
            assert int(r[8]) == g[7], f"{name}: arm bytes of window {beg}-{end} differ"
            assert zlib.crc32(r[9].encode()) == g[8], f"{name}: consensus of window {beg}-{end} differs"
    assert hashlib.md5(got).hexdigest() == man["expected_fasta_md5"]
    assert got == want
    return len(regions)


def messy_seeds():
    """{seed: {"args": [...], "fasta_md5": ...}}: what the REAL reference binary produced for gen_e2e.generate_messy(seed)
    (tests/golden/fuzz_e2e.py, build container)."""
    return {int(k): v for k, v in json.load(open(os.path.join(GOLD, "e2e_messy_md5.json"))).items()}


def run_messy_seed(seed, rec, outdir, device, threads=4):
    spec = importlib.util.spec_from_file_location("gen_e2e", os.path.join(GOLD, "gen_e2e.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    argv, _, _ = gen.generate_messy(str(outdir), seed)
    assert argv == rec["args"], f"seed {seed}: the generator's options changed"
    argv = [BIN] + argv
    argv[argv.index("-t") + 1] = str(threads)
    env = dict(os.environ)
    if device == "shim":
        env["LD_LIBRARY_PATH"] = SHIM_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    else:
        env.setdefault("HYPO_REQUIRE_DEVICE", "1")      # a stage that quietly ran in the host loops fails the run (hypo --require-device)
    p = subprocess.run(argv, cwd=str(outdir), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert ("oracle_device_shim" in p.stderr) == (device == "shim")
    got = hashlib.md5(open(os.path.join(str(outdir), "hypo_draft.fasta"), "rb").read()).hexdigest()
    assert got == rec["fasta_md5"], f"messy seed {seed} ({' '.join(rec['args'][10:])}): FASTA differs from the reference's"


# ---- sets made by the C++ generator (tests/golden/gen_e2e_fast.cpp): BASELINE config C3 and larger ------------------------------
FAST_GEN_SRC = os.path.join(GOLD, "gen_e2e_fast.cpp")
FAST_GEN_BIN = os.path.join(HERE, "_build", "gen_e2e_fast")


def build_fast_generator():
    if not os.path.exists(FAST_GEN_BIN) or os.path.getmtime(FAST_GEN_BIN) < os.path.getmtime(FAST_GEN_SRC):
        os.makedirs(os.path.dirname(FAST_GEN_BIN), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-fopenmp", "-o", FAST_GEN_BIN, FAST_GEN_SRC, "-lz"])
    return FAST_GEN_BIN


def run_fast_case(name, outdir, threads, extra_args=(), extra_env=None, timeout=1800, bam=False, device="gpu"):
    """Generates the inputs of golden `name` with the C++ generator (checked against the manifest's checksums), runs the `hypo`
    binary on them.  Returns (manifest, CompletedProcess, seconds of the run, peak RSS of the child in MB).  bam: the same records
    as BAM (BGZF) instead of SAM text (the generator's --bam; the draft, the solid set and the number of records are checked against
    the manifest, the records themselves through the FASTA they lead to)."""
    import resource
    import time
    man = json.load(open(os.path.join(GOLD, name + ".manifest.json")))
    a = man["args"]
    gen = build_fast_generator()
    flags = list(a["flags"]) if "flags" in a else []
    golden_is_bam = "--bam" in flags
    if bam and not golden_is_bam:
        flags.append("--bam")
    rep = json.loads(subprocess.check_output([gen, str(outdir), str(a["seed"]), str(a["contigs"]), str(a["contig_len"]), str(a["k"]),
                                              str(a["coverage"]), str(a["read_len"]), str(a["read_sub_ppm"])] + flags, text=True))
    want = man["generator_report"]
    if golden_is_bam or ("flags" in a and not bam):  # a golden made from the flagged output itself: everything must match
        bam = golden_is_bam
        assert rep == want, f"{name}: the generator's output changed: {rep}"
    elif bam:                                        # the records of a SAM golden as BAM
        assert "fnv_bam_blocks" in rep
        for key in ("contigs", "draft_bases", "reads", "solid_kmers", "fnv_draft", "fnv_bitvector", "long_reads"):
            assert rep.get(key) == want.get(key), f"{name}: the generator's output changed: {key} = {rep.get(key)}"
    else:
        assert rep == want, f"{name}: the generator's output changed: {rep}"
    argv = [BIN] + man["command"].split()[1:]
    if bam:
        argv[argv.index("-b") + 1] = "sr.bam"
    if "-B" in argv and bam:
        argv[argv.index("-B") + 1] = "lr.bam"
    if "our_p" in a and "-p" in argv:
        argv[argv.index("-p") + 1] = str(a["our_p"])
    argv[argv.index("-t") + 1] = str(threads)
    argv += list(extra_args)
    env = dict(os.environ)
    if device == "shim":                             # the CPU oracle behind the C-ABI (tests/shim): the host pipeline without a GPU
        env["LD_LIBRARY_PATH"] = SHIM_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    else:
        env.setdefault("HYPO_REQUIRE_DEVICE", "1")      # a stage that quietly ran in the host loops fails the run (hypo --require-device)
    env.update(extra_env or {})
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    t0 = time.perf_counter()
    p = subprocess.run(argv, cwd=str(outdir), env=env, capture_output=True, text=True, timeout=timeout)
    dt = time.perf_counter() - t0
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert ("oracle_device_shim" in p.stderr) == (device == "shim"), "wrong device library behind the C-ABI"
    rss = max(resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss, before) / 1024.0
    return man, p, dt, rss


def fasta_md5(outdir):
    h = hashlib.md5()
    with open(os.path.join(str(outdir), "hypo_draft.fasta"), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


# ---- fresh inputs against the reference's own stage compiled in place (oracle/_ref/libhyporef_arms.so) --------------------------
def _gen():
    spec = importlib.util.spec_from_file_location("gen_e2e", os.path.join(GOLD, "gen_e2e.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def _reference_dump_regions(path):
    spec = importlib.util.spec_from_file_location("make_e2e_golden", os.path.join(GOLD, "make_e2e_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.regions(path)


def long_reads_pass_nm_filter(sam_path, ned_th=20):
    """True when no record of the long-read file would be dropped by the reference's NM filter (src/Alignment.cpp:51-58: NM * 100 /
    reference span > -n): the in-place harness builds long reads through the short-read constructor, which has no such filter."""
    import re
    for line in open(sam_path):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        if int(f[1]) & (4 | 256 | 512 | 1024):
            continue
        span = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", f[5]) if op in "MDN=X")
        m = re.search(r"\tNM:i:(-?\d+)", line)
        if m and span and int(m.group(1)) * 100 // span > ned_th:
            return False
    return True


def run_vs_reference_stage(outdir, seed, device, messy, threads=4, long_reads=False, k=None):
    """One single-contig short-read set made NOW (no committed golden behind it): this repo's `hypo` (region dump) against the
    real Alignment / Contig / Window code of the reference run on the same records (oracle.RefArms): region borders and types
    (A14, through the support votes N1), arm counts and crc32 of the arms of every window (A13 / N2).  Returns the number of
    windows compared, or None when the seed's messy set has several contigs or long reads (the harness is short-read only)."""
    import oracle
    os.makedirs(str(outdir), exist_ok=True)
    gen = _gen()
    with_long = False
    if messy:
        args, nc, with_long = gen.generate_messy(str(outdir), seed)
        if nc != 1 or with_long != long_reads:
            return None
    else:
        # k: the caller's (13 / 15 / 17: the 32- and 64-bit-id vote kernels, nearly every position of a small genome marked: the
        # regime of the 250 Mbp - 3 Gbp sets) or 7 / 9 / 11 by seed; -s is the size flag that derives it (src/main.cpp:490-528)
        k = k if k is not None else [7, 9, 11][seed % 3]
        gen.generate(str(outdir), seed, [8000, 20000, 40000][seed % 3], long_reads, k)
        with_long = long_reads
        args = ["-d", "draft.fa", "-r", "reads.fa", "-s", SIZE_FLAG_OF_K[k], "-c", "30", "-b", "sr.sam"] + (["-B", "lr.sam"] if long_reads else []) + ["-t", "1", "-i"]
    if with_long:
        ned = int(args[args.index("-n") + 1]) if "-n" in args else 20
        if not long_reads_pass_nm_filter(os.path.join(str(outdir), "lr.sam"), ned):
            return None
    k = {v: kk for kk, v in SIZE_FLAG_OF_K.items()}[args[args.index("-s") + 1]]
    mq = int(args[args.index("-q") + 1]) if "-q" in args else 2
    argv = [BIN] + args
    argv[argv.index("-t") + 1] = str(threads)
    env = dict(os.environ)
    env["HYPO_REGION_DUMP"] = os.path.join(str(outdir), "regions.tsv")
    if device == "shim":
        env["LD_LIBRARY_PATH"] = SHIM_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    else:
        env.setdefault("HYPO_REQUIRE_DEVICE", "1")      # a stage that quietly ran in the host loops fails the run (hypo --require-device)
    p = subprocess.run(argv, cwd=str(outdir), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert ("oracle_device_shim" in p.stderr) == (device == "shim"), "wrong device library behind the C-ABI"
    if device != "shim":
        assert "short arms cut on the device" in p.stdout          # (unsorted records too: sorted on ingest, arms in file order)
        if with_long:
            assert "long arms cut on the device" in p.stdout          # (unsorted -B files too since round 5: sorted on ingest, arms in file order)
    fa = open(os.path.join(str(outdir), "draft.fa")).read().split("\n")
    name, draft = fa[0][1:].split()[0], "".join(fa[1:])
    ref = oracle.RefArms()
    recs = ref.sam_records(os.path.join(str(outdir), "sr.sam"), name, mq)
    work = os.path.join(str(outdir), "refstage")
    os.makedirs(work, exist_ok=True)
    lrecs = ref.sam_records(os.path.join(str(outdir), "lr.sam"), name, mq) if with_long else None
    if not with_long:
        # row A15 in place: the FASTA record the reference's own operator<<(Contig) writes after its own stage and its own POA
        # (hyporef_fasta) against the file this repo's binary wrote — no CMake-built binary in between
        ref_fa = os.path.join(work, "ref.fa")
        sc = [5, -4, -8, 3, -5, -4]
        for i, fl in enumerate(("-m", "-x", "-g", "-M", "-X", "-G")):
            if fl in args:
                sc[i] = int(args[args.index(fl) + 1])
        ref.fasta(draft.encode(), name, k, os.path.join(str(outdir), "aux", "solid_kmers.bvsd"), recs, ref_fa, scores=sc)
        ours = open(os.path.join(str(outdir), "hypo_draft.fasta"), "rb").read()
        assert ours == open(ref_fa, "rb").read(), f"seed {seed}: polished FASTA differs from the reference's own operator<<(Contig)"
    dump = ref.regions_dump(draft.encode(), k, os.path.join(str(outdir), "aux", "solid_kmers.bvsd"), recs, work, long_records=lrecs)
    regions = _reference_dump_regions(dump)
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(str(outdir), "regions.tsv"))]
    assert len(rows) == len(regions), f"seed {seed}: {len(rows)} regions, reference stage has {len(regions)}"
    n = 0
    global LAST_LONG_WINDOWS
    LAST_LONG_WINDOWS = 0
    for r, g in zip(rows, regions):
        beg, end, typ = int(r[1]), int(r[2]) - 1, r[3]
        assert [beg, end, typ] == g[:3], f"seed {seed}: region {r[:4]} vs reference {g[:3]}"
        if typ not in ("SR", "MSR"):
            assert [int(x) for x in r[4:8]] == g[3:7], f"seed {seed}: arms of window {beg}-{end}: {r[4:8]} vs reference {g[3:7]}"
            assert int(r[8]) == g[7], f"seed {seed}: arm bytes of window {beg}-{end} differ"
            n += 1
            LAST_LONG_WINDOWS += typ == "LNG"
    return n


SIZE_FLAG_OF_K = {7: "10k", 9: "100k", 11: "1m", 13: "100m", 15: "250m", 17: "3g"}     # -s values and the k they derive
LAST_LONG_WINDOWS = 0          # LONG windows among those the last run_vs_reference_stage call compared


# ---- whole FILES through the reference compiled in place (rows T1 and N3; oracle/ref_arms_harness.cpp: hyporef_fasta_bam / _sam) -------
def fasta_records(path):
    """{name: sequence} of a FASTA file (the name up to the first blank)."""
    d, name, parts = {}, None, []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    d[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line.rstrip("\n"))
    if name is not None:
        d[name] = "".join(parts)
    return d


def run_messy_files_vs_reference(outdir, seed, device, as_bam, threads=4):
    """One messy set WITHOUT long reads (1-3 contigs; clips, `=`/`X`, unmapped / secondary / duplicate / QC-fail flags, low mapping
    qualities, N and lower case, non-default -q / scores): this repo's `hypo` reads the SAM or BAM file with its own readers
    (host/SeqIO.hpp); the reference's code gets the same file through the harness's independent minimal decoder as bam1_t records and
    applies its own field extraction (Alignment::initialise_pos / copy_data).  The two FASTA files must hold the same records byte for
    byte: row N3 pinned in place, no CMake-built binary in between.  Returns the number of alignments the reference kept, or None when
    the seed's set has long reads."""
    import oracle
    os.makedirs(str(outdir), exist_ok=True)
    gen = _gen()
    args, nc, with_long = gen.generate_messy(str(outdir), seed)
    if with_long:
        return None
    aln = "sr.sam"
    if as_bam:
        import bam_util
        bam_util.sam_to_bam(os.path.join(str(outdir), "sr.sam"), os.path.join(str(outdir), "sr.bam"), nm_type="i", extra_tags=True)
        aln = "sr.bam"
        args = list(args)
        args[args.index("-b") + 1] = aln
    k = {v: kk for kk, v in SIZE_FLAG_OF_K.items()}[args[args.index("-s") + 1]]
    mq = int(args[args.index("-q") + 1]) if "-q" in args else 2
    sc = [5, -4, -8, 3, -5, -4]
    for i, fl in enumerate(("-m", "-x", "-g", "-M", "-X", "-G")):
        if fl in args:
            sc[i] = int(args[args.index(fl) + 1])
    argv = [BIN] + list(args)
    argv[argv.index("-t") + 1] = str(threads)
    env = dict(os.environ)
    if device == "shim":
        env["LD_LIBRARY_PATH"] = SHIM_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    else:
        env.setdefault("HYPO_REQUIRE_DEVICE", "1")
    p = subprocess.run(argv, cwd=str(outdir), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert ("oracle_device_shim" in p.stderr) == (device == "shim"), "wrong device library behind the C-ABI"
    ref = oracle.RefArms()
    rep = ref.fasta_file(os.path.join(str(outdir), "draft.fa"), os.path.join(str(outdir), aln), k, os.path.join(str(outdir), "aux", "solid_kmers.bvsd"),
                         os.path.join(str(outdir), "ref_files.fa"), min_mapq=mq, scores=sc)
    ours, theirs = fasta_records(os.path.join(str(outdir), "hypo_draft.fasta")), fasta_records(os.path.join(str(outdir), "ref_files.fa"))
    assert list(ours) and set(ours) == set(theirs) and rep["contigs"] == nc, (seed, list(ours), list(theirs))
    for name in ours:
        assert ours[name] == theirs[name], f"messy seed {seed} ({aln}): record {name} differs from the reference's own polish of the same file"
    return rep["alignments"]


def run_t1_slice_vs_reference(outdir, n_contigs, contig_len, k, size_flag, pick, threads, device="gpu", seed=97, p=50):
    """Row T1's shape (x 1 Mbp contigs from the C++ generator as BAM, `-s <size_flag>` -> k) through this repo's `hypo`, and the contigs
    `pick` through the reference compiled in place with the records of the same BAM file (hyporef_fasta_bam).  Returns (our seconds,
    the reference's report, number of picked records that are byte-identical)."""
    import time
    import oracle
    gen = build_fast_generator()
    rep = json.loads(subprocess.check_output([gen, str(outdir), str(seed), str(n_contigs), str(contig_len), str(k), "30", "150", "2000", "--bam", "--fast-hash"], text=True))
    argv = [BIN, "-d", "draft.fa", "-r", "reads.fa", "-s", size_flag, "-c", "30", "-b", "sr.bam", "-t", str(threads), "-i", "-p", str(p), "-o", "out.fa"]
    env = dict(os.environ)
    if device == "shim":
        env["LD_LIBRARY_PATH"] = SHIM_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    else:
        env.setdefault("HYPO_REQUIRE_DEVICE", "1")
    t0 = time.perf_counter()
    pr = subprocess.run(argv, cwd=str(outdir), env=env, capture_output=True, text=True, timeout=3000)
    dt = time.perf_counter() - t0
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-2000:]
    assert f"({size_flag}): {k}" in pr.stdout, pr.stdout[:400]
    ref = oracle.RefArms()
    rr = ref.fasta_file(os.path.join(str(outdir), "draft.fa"), os.path.join(str(outdir), "sr.bam"), k, os.path.join(str(outdir), "aux", "solid_kmers.bvsd"),
                        os.path.join(str(outdir), "ref_pick.fa"), pick=pick)
    ours, theirs = fasta_records(os.path.join(str(outdir), "out.fa")), fasta_records(os.path.join(str(outdir), "ref_pick.fa"))
    assert len(ours) == rep["contigs"] and len(theirs) == len(set(pick))
    same = sum(1 for name in theirs if ours.get(name) == theirs[name])
    return dt, rr, same


def realistic_window_batch(outdir, name="e2e_real_5m_s131"):
    """The REAL windows of a non-i.i.d. set, as the reference itself cut them: the inputs of golden `name` are generated, the reference
    compiled in place polishes them (oracle.RefArms.fasta_file) and leaves its own per-region dump (Contig::generate_inspect_file: arms,
    draft and CONSENSUS of every window); returns (HostBatch of all windows, the reference's consensus per window, manifest, the reference's
    report).  These windows have what the simulator's lack: read indels inside SHORT windows, repeats, a second haplotype, mis-placed reads."""
    import importlib.util
    import oracle
    from hypo_amd.batch import build_batch
    man = json.load(open(os.path.join(GOLD, name + ".manifest.json")))
    a = man["args"]
    gen = build_fast_generator()
    rep = json.loads(subprocess.check_output([gen, str(outdir), str(a["seed"]), str(a["contigs"]), str(a["contig_len"]), str(a["k"]), str(a["coverage"]),
                                              str(a["read_len"]), str(a["read_sub_ppm"])] + list(a["flags"]), text=True))
    assert rep == man["generator_report"], f"{name}: the generator's output changed"
    with_long = "--long" in a["flags"]
    dump = os.path.join(str(outdir), "refdump")
    os.makedirs(dump, exist_ok=True)
    rr = oracle.RefArms().fasta_file(os.path.join(str(outdir), "draft.fa"), os.path.join(str(outdir), "sr.bam"), a["k"], os.path.join(str(outdir), "aux", "solid_kmers.bvsd"),
                                     os.path.join(str(outdir), "ref.fa"), long_path=os.path.join(str(outdir), "lr.bam") if with_long else None, dump_dir=dump)
    assert hashlib.md5(open(os.path.join(str(outdir), "ref.fa"), "rb").read()).hexdigest() == man["expected_fasta_md5"]
    spec = importlib.util.spec_from_file_location("inspect_dump", os.path.join(GOLD, "inspect_dump.py"))
    idump = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(idump)
    wins, cons = [], []
    for c in range(a["contigs"]):
        for hdr, typ, w, cs in idump.parse(os.path.join(dump, "aux", f"inspect_ctg{c + 1}.txt")):
            wins.append(w)
            cons.append(cs)
    return build_batch(wins), cons, man, rr


def run_homopolymer_set(outdir, device, devices=None, threads=8):
    """One 90 kbp contig whose truth holds a 4 000-base poly-A run, 30x short reads and 25x noisy 6 kbp long reads (`-B`): no window border can
    be put into the run (Contig::force_divide, src/Contig.cpp:641-666), short reads cannot cover a window that long (it is pruned,
    src/Contig.cpp:264-275), so it becomes ONE LONG window of ~4 000 bases with ~40 long-read arms — beyond every table-driven size class
    (size class 6) and longer than piece mode's first halo.  Runs `hypo` (optionally with --devices) and the reference compiled in place on
    the same files; returns (CompletedProcess, True when the two FASTA files are byte-identical, length of the longest LONG window)."""
    import oracle
    gen = build_fast_generator()
    subprocess.check_output([gen, str(outdir), "141", "1", "90000", "9", "30", "150", "2000", "--bam", "--fast-hash", "--homopolymer", "29950", "4000", "--long", "25", "6000"])
    argv = [BIN, "-d", "draft.fa", "-r", "reads.fa", "-s", "100k", "-c", "30", "-b", "sr.bam", "-B", "lr.bam", "-t", str(threads), "-i", "-o", "out.fa"]
    env = dict(os.environ, HYPO_REGION_DUMP=os.path.join(str(outdir), "regions.tsv"))
    if devices:
        argv += ["--devices", devices]
        env["HYPO_ALLOW_DUP_DEVICES"] = "1"
    if device == "shim":
        env["LD_LIBRARY_PATH"] = SHIM_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    else:
        env.setdefault("HYPO_REQUIRE_DEVICE", "1")
    p = subprocess.run(argv, cwd=str(outdir), env=env, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    oracle.RefArms().fasta_file(os.path.join(str(outdir), "draft.fa"), os.path.join(str(outdir), "sr.bam"), 9, os.path.join(str(outdir), "aux", "solid_kmers.bvsd"),
                                os.path.join(str(outdir), "ref.fa"), long_path=os.path.join(str(outdir), "lr.bam"))
    same = open(os.path.join(str(outdir), "ref.fa"), "rb").read() == open(os.path.join(str(outdir), "out.fa"), "rb").read()
    longest = max((int(r[2]) - int(r[1]) for r in (l.split("\t") for l in open(os.path.join(str(outdir), "regions.tsv"))) if r[3].strip() == "LNG"), default=0)
    return p, same, longest
