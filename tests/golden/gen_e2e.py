#!/usr/bin/env python3
"""Deterministic synthetic end-to-end inputs: truth, draft with sub/ins/del 0.4 %
each, 30x 150-bp reads with 0.2 % substitutions as coordinate-sorted SAM text whose CIGARs are the composition of
the truth->draft edit script, the solid-kmer bit vector file the reference loads with `-i`
(aux/solid_kmers.bvsd + aux/stage.txt), and optionally 40x 8-kbp ONT-like long reads (~8 % error, NM tag) with ten
3-kbp short-read coverage gaps so that LONG windows appear.

The byte-exact inputs of the committed end-to-end goldens are reproduced by
    gen_e2e.py <outdir> 1 20000          (tests/golden/e2e_20k_s1.*)
    gen_e2e.py <outdir> 3 200000 --long  (tests/golden/e2e_200k_long_s3.*)
    gen_e2e.py <outdir> 5 200000 --k 9   (tests/golden/e2e_200k_k9_s5.*: `-s 100k` => k = 9, many minimizer-cut windows)
    gen_e2e.py <outdir> 21 60000 --long --contigs 5   (tests/golden/e2e_5ctg_long_s21.*: five contigs, run with `-p 2`)
(python's `random` module, seed and call order fixed; md5 of every file is checked against the manifest by the
tests).  The expected outputs in those goldens were produced from exactly these inputs by the reference binary
built per SURVEY.md Appendix B (`hypo -d draft.fa -r reads.fa -s 1m -c 30 -b sr.sam [-B lr.sam] -t 1 -i`)."""
import os
import random
import struct
import sys

COV, RL = 30, 150
A = "ACGT"


def rle(cig):
    out, last, cnt = [], None, 0
    for c in cig:
        if c == last:
            cnt += 1
        else:
            if last:
                out.append(f"{cnt}{last}")
            last, cnt = c, 1
    out.append(f"{cnt}{last}")
    return "".join(out)


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def generate(outdir, seed, G, with_long, K=11, read_len=RL, read_sub=0.002):
    """read_len / read_sub: HiFi-like reads for the --ccs-windows golden (defaults: the 150-bp short reads of every other set)"""
    random.seed(seed)
    truth = "".join(random.choice(A) for _ in range(G))
    ops = []                                     # (op, truth base, draft base)
    for c in truth:
        r = random.random()
        if r < 0.004:
            ops.append(('D', c, None))           # draft lacks this base -> I in read CIGARs
        elif r < 0.008:
            ops.append(('X', c, random.choice([b for b in A if b != c])))
        else:
            ops.append(('M', c, c))
        if random.random() < 0.004:
            ops.append(('I', None, random.choice(A)))   # draft has an extra base -> D in read CIGARs
    draft = "".join(o[2] for o in ops if o[2] is not None)
    tpos = [i for i, o in enumerate(ops) if o[1] is not None]
    dprefix = [0] * (len(ops) + 1)
    for i, o in enumerate(ops):
        dprefix[i + 1] = dprefix[i] + (1 if o[2] is not None else 0)
    gaps = [(g * G // 10 + 3000, g * G // 10 + 6000) for g in range(10)] if with_long else []
    nreads = G * COV // read_len
    recs = []
    for n in range(nreads):
        s = random.randrange(0, G - read_len)
        e = s + read_len
        if with_long and any(s < b and e > a for a, b in gaps):
            continue
        i0, i1 = tpos[s], tpos[e - 1] + 1
        while ops[i0][0] not in "MX":
            i0 += 1
        while ops[i1 - 1][0] not in "MX":
            i1 -= 1
        seq, cig = [], []
        for o in ops[i0:i1]:
            if o[0] in "MX":
                b = o[1]
                if random.random() < read_sub:
                    b = random.choice(A)
                seq.append(b)
                cig.append('M')
            elif o[0] == 'D':
                seq.append(o[1])
                cig.append('I')
            else:
                cig.append('D')
        recs.append((dprefix[i0], n, rle(cig), "".join(seq)))
    recs.sort()
    os.makedirs(os.path.join(outdir, "aux"), exist_ok=True)
    w = lambda name, text: open(os.path.join(outdir, name), "w").write(text)
    w("draft.fa", ">ctg1\n" + draft + "\n")
    w("truth.fa", ">ctg1\n" + truth + "\n")
    w("reads.fa", "".join(f">r{n}\n{s}\n" for p, n, c, s in recs))
    w("sr.sam", f"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ctg1\tLN:{len(draft)}\n" +
      "".join(f"r{n}\t0\tctg1\t{p + 1}\t60\t{c}\t*\t0\t0\t{s}\t*\n" for p, n, c, s in recs))
    # solid k-mers: canonical k-mers occurring exactly once in the truth, no homopolymer at the terminals,
    # both strands set (external/suk/src/SolidKmers.cpp:166-189)
    from collections import Counter
    cnt = Counter()

    def enc(s):
        v = 0
        for ch in s:
            v = (v << 2) | A.index(ch)
        return v
    for i in range(G - K + 1):
        km = truth[i:i + K]
        cnt[min(km, rc(km))] += 1
    nb = 1 << (2 * K)
    ns = 0
    if K >= 12:                                  # 4^k bits no longer fit a Python list (k = 17: 2 GiB): same bits through numpy
        import numpy as np
        words = np.zeros(nb // 64, dtype=np.uint64)
        for km, c in cnt.items():
            if c == 1 and km[0] != km[1] and km[-1] != km[-2]:
                for x in (km, rc(km)):
                    v = enc(x)
                    words[v >> 6] |= np.uint64(1 << (v & 63))
                ns += 1
        with open(os.path.join(outdir, "aux", "solid_kmers.bvsd"), "wb") as f:
            f.write(struct.pack("<Q", nb))
            words.tofile(f)
    else:
        words = [0] * (nb // 64)
        for km, c in cnt.items():
            if c == 1 and km[0] != km[1] and km[-1] != km[-2]:
                for x in (km, rc(km)):
                    v = enc(x)
                    words[v >> 6] |= (1 << (v & 63))
                ns += 1
        with open(os.path.join(outdir, "aux", "solid_kmers.bvsd"), "wb") as f:
            f.write(struct.pack("<Q", nb))
            f.write(struct.pack(f"<{len(words)}Q", *words))
    w("aux/stage.txt", "Stage:SolidKmers [2026-09-28 12:00:00]\t1\n")
    if with_long:
        LCOV, LL = 40, 8000
        nl = G * LCOV // LL
        lrecs = []
        for n in range(nl):
            s = random.randrange(0, max(1, G - LL))
            e = min(G, s + LL)
            i0, i1 = tpos[s], tpos[e - 1] + 1
            while ops[i0][0] not in "MX":
                i0 += 1
            while ops[i1 - 1][0] not in "MX":
                i1 -= 1
            seq, cig, nm, first = [], [], 0, True
            for idx, o in enumerate(ops[i0:i1]):
                last = (idx == i1 - i0 - 1)
                if o[1] is not None:
                    r = random.random()
                    has, b = True, o[1]
                    if not first and not last:
                        if r < 0.03:
                            has = False
                        elif r < 0.06:
                            b = random.choice([x for x in A if x != o[1]])
                    if o[0] in "MX":
                        if has:
                            seq.append(b); cig.append('M'); nm += (b != o[2])
                        else:
                            cig.append('D'); nm += 1
                    else:
                        if has:
                            seq.append(b); cig.append('I'); nm += 1
                    if not last and random.random() < 0.02:
                        seq.append(random.choice(A)); cig.append('I'); nm += 1
                else:
                    cig.append('D'); nm += 1
                first = False
            lrecs.append((dprefix[i0], n, rle(cig), "".join(seq), nm))
        lrecs.sort()
        w("lr.sam", f"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ctg1\tLN:{len(draft)}\n" +
          "".join(f"l{n}\t0\tctg1\t{p + 1}\t60\t{c}\t*\t0\t0\t{s}\t*\tNM:i:{nm}\n" for p, n, c, s, nm in lrecs))
    return len(draft), len(recs), ns


def generate_multi(outdir, seed, G, with_long, K, n_contigs):
    """n_contigs independent sets (seeds seed, seed+1, ...; lengths G, G/2, G/3, ...) merged into one input: contigs
    ctg1..ctgN in one draft, records contig after contig (coordinate-sorted per contig, header order), the OR of the
    solid-kmer sets.  Exercises contig batches (-p) and the reader's look-ahead across batch borders."""
    import shutil
    import tempfile
    drafts, reads, sam_hdr, sam_body, lsam_body, words_all, total = [], [], [], [], [], None, [0, 0, 0]
    for c in range(n_contigs):
        tmp = tempfile.mkdtemp(prefix="gen_e2e_")
        try:
            res = generate(tmp, seed + c, G // (c + 1), with_long, K)
            name = f"ctg{c + 1}"
            d = open(os.path.join(tmp, "draft.fa")).read().split("\n")[1]
            drafts.append(f">{name} part {c + 1}\n{d}\n")           # header with a comment: only the first token is the name
            reads.append(open(os.path.join(tmp, "reads.fa")).read().replace(">r", f">c{c + 1}r"))
            for line in open(os.path.join(tmp, "sr.sam")):
                if line.startswith("@SQ"):
                    sam_hdr.append(line.replace("SN:ctg1", "SN:" + name))
                elif not line.startswith("@"):
                    f = line.split("\t")
                    f[0] = f"c{c + 1}{f[0]}"; f[2] = name
                    sam_body.append("\t".join(f))
            if with_long:
                for line in open(os.path.join(tmp, "lr.sam")):
                    if not line.startswith("@"):
                        f = line.split("\t")
                        f[0] = f"c{c + 1}{f[0]}"; f[2] = name
                        lsam_body.append("\t".join(f))
            raw = open(os.path.join(tmp, "aux", "solid_kmers.bvsd"), "rb").read()
            words = list(struct.unpack(f"<{(len(raw) - 8) // 8}Q", raw[8:]))
            words_all = words if words_all is None else [a | b for a, b in zip(words_all, words)]
            for i in range(3):
                total[i] += res[i]
        finally:
            shutil.rmtree(tmp)
    os.makedirs(os.path.join(outdir, "aux"), exist_ok=True)
    w = lambda name, text: open(os.path.join(outdir, name), "w").write(text)
    w("draft.fa", "".join(drafts))
    w("reads.fa", "".join(reads))
    hdr = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(sam_hdr)
    w("sr.sam", hdr + "".join(sam_body))
    if with_long:
        w("lr.sam", hdr + "".join(lsam_body))
    with open(os.path.join(outdir, "aux", "solid_kmers.bvsd"), "wb") as f:
        f.write(struct.pack("<Q", 1 << (2 * K)))
        f.write(struct.pack(f"<{len(words_all)}Q", *words_all))
    w("aux/stage.txt", "Stage:SolidKmers [2026-09-28 12:00:00]\t1\n")
    return tuple(total)


def generate_messy(outdir, seed):
    """A small set with everything the clean sets lack: N and lower-case stretches in the draft, soft and hard clips,
    `=`/`X` CIGAR operators, unmapped / secondary / duplicate / QC-fail / reverse-strand flags, low mapping qualities,
    N and lower-case bases in reads, long reads without an NM tag, 1-3 contigs, with or without long reads, and
    non-default -p / -q / -m -x -g / -n options.  Returns the command-line arguments (after the program name) that go
    with the set.  Used by tests/golden/fuzz_e2e.py (differential fuzz against the real reference binary) and by the
    committed `e2e_messy_*` goldens."""
    rnd = random.Random(seed * 7919 + 13)
    nc = rnd.choice([1, 2, 3])
    G = rnd.choice([6000, 15000, 30000])
    with_long = rnd.random() < 0.5
    K = rnd.choice([7, 9, 11])
    if nc > 1:
        generate_multi(outdir, seed, G, with_long, K, nc)
    else:
        generate(outdir, seed, G, with_long, K)
    lines = open(os.path.join(outdir, "draft.fa")).read().split("\n")
    out = []
    for l in lines:
        if l.startswith(">") or not l:
            out.append(l)
            continue
        s = list(l)
        for _ in range(max(1, len(s) // 3000)):
            s[rnd.randrange(len(s))] = "N"
        if rnd.random() < 0.5:
            a = rnd.randrange(len(s))
            b = min(len(s), a + rnd.randrange(1, 200))
            s[a:b] = [c.lower() for c in s[a:b]]
        out.append("".join(s))
    open(os.path.join(outdir, "draft.fa"), "w").write("\n".join(out))
    for name in ["sr.sam"] + (["lr.sam"] if with_long else []):
        p = os.path.join(outdir, name)
        res = []
        for line in open(p):
            if line.startswith("@"):
                res.append(line)
                continue
            f = line.rstrip("\n").split("\t")
            r = rnd.random()
            cig, seq = f[5], f[9]
            if r < 0.08:
                a, b = rnd.randrange(0, 12), rnd.randrange(0, 12)
                seq = "".join(rnd.choice(A) for _ in range(a)) + seq + "".join(rnd.choice(A) for _ in range(b))
                cig = (f"{a}S" if a else "") + cig + (f"{b}S" if b else "")
            elif r < 0.12:
                cig = f"{rnd.randrange(1, 30)}H" + cig + f"{rnd.randrange(1, 30)}H"
            elif r < 0.16:
                cig = cig.replace("M", "=") if rnd.random() < 0.5 else cig.replace("M", "X")
            r = rnd.random()
            if r < 0.03:
                f[1] = str(int(f[1]) | 4)
            elif r < 0.06:
                f[1] = str(int(f[1]) | 256)
            elif r < 0.08:
                f[1] = str(int(f[1]) | 1024)
            elif r < 0.10:
                f[1] = str(int(f[1]) | 512)
            elif r < 0.3:
                f[1] = str(int(f[1]) | 16)
            if rnd.random() < 0.1:
                f[4] = str(rnd.choice([0, 1, 2, 3, 10]))
            if rnd.random() < 0.02:
                s = list(seq)
                s[rnd.randrange(len(s))] = "N"
                seq = "".join(s)
            if rnd.random() < 0.02:
                seq = seq.lower()
            if name == "lr.sam" and rnd.random() < 0.05:
                f = f[:11]
            f[5], f[9] = cig, seq
            res.append("\t".join(f) + "\n")
        open(p, "w").write("".join(res))
    size = {7: "10k", 9: "100k", 11: "1m"}[K]
    extra = []
    if rnd.random() < 0.4:
        extra += ["-p", str(rnd.choice([1, 2]))]
    if rnd.random() < 0.3:
        extra += ["-q", str(rnd.choice([0, 5, 20]))]
    if rnd.random() < 0.3:
        extra += ["-m", "3", "-x", "-6", "-g", "-5"]
    if with_long and rnd.random() < 0.3:
        extra += ["-n", str(rnd.choice([5, 12, 40]))]
    return (["-d", "draft.fa", "-r", "reads.fa", "-s", size, "-c", "30", "-b", "sr.sam"] + (["-B", "lr.sam"] if with_long else []) +
            ["-t", "1", "-i"] + extra), nc, with_long


if __name__ == "__main__":
    out, seed, G = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 11
    nc = int(sys.argv[sys.argv.index("--contigs") + 1]) if "--contigs" in sys.argv else 1
    if nc > 1:
        print("draft %d reads %d solid %d" % generate_multi(out, seed, G, "--long" in sys.argv, k, nc))
    else:
        print("draft %d reads %d solid %d" % generate(out, seed, G, "--long" in sys.argv, k))
