#!/usr/bin/env python3
"""Packs one end-to-end run of the REAL reference into the committed goldens
    <name>.manifest.json     generator arguments, md5 of every input file, md5 of the polished FASTA
    <name>.expected.fa.gz    the reference's polished FASTA
    <name>.regions.json.gz   one row per region of the reference's inspect dump:
                             [beg, end, type, n_internal, n_prefix, n_suffix, n_empty, crc32(arms joined by \\n), crc32(consensus)]
usage: make_e2e_golden.py <name> <run dir> <polished fasta> <seed> <G> <k> <size flag> [--long]
The run dir holds the inputs written by gen_e2e.py and the aux/inspect_ctg1.txt the reference binary (built per
SURVEY.md Appendix B with src/Hypo.cpp:262,265,271 enabled) wrote for them; build container only."""
import gzip
import hashlib
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))


def md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def regions(path):
    out = []
    with open(path) as f:
        f.readline(); f.readline()
        line = f.readline()
        while line:
            parts = line.rstrip("\n").split("\t")
            ni, np_, ns, ne = (int(x) for x in parts[2:6])
            beg, end = parts[0].strip("=()").split("-")
            f.readline()
            cons = f.readline().rstrip("\n")[3:]
            arms = []
            line = f.readline()
            while line and not line.startswith("=========="):
                arms.append(line.rstrip("\n"))
                line = f.readline()
            out.append([int(beg), int(end), parts[1], ni, np_, ns, ne, zlib.crc32("\n".join(arms).encode()), zlib.crc32(cons.encode())])
    return out


def main_messy():
    """make_e2e_golden.py --messy <name> <run dir> <polished fasta> <seed>: golden of one gen_e2e.generate_messy set."""
    import tempfile
    sys.path.insert(0, HERE)
    import gen_e2e
    name, d, outfa, seed = sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    with tempfile.TemporaryDirectory() as tmp:
        cmd, nc, is_long = gen_e2e.generate_messy(tmp, seed)
        ins = ["draft.fa", "sr.sam", "aux/solid_kmers.bvsd"] + (["lr.sam"] if is_long else [])
        for f in ins:
            assert md5(os.path.join(tmp, f)) == md5(os.path.join(d, f)), f"{f}: run dir does not hold generate_messy({seed})"
    man = {"generator": "tests/golden/gen_e2e.py", "args": {"messy": seed, "contigs": nc, "long": is_long},
           "command": "hypo " + " ".join(cmd), "inputs_md5": {f: md5(os.path.join(d, f)) for f in ins},
           "expected_fasta_md5": md5(os.path.join(d, outfa))}
    json.dump(man, open(os.path.join(HERE, name + ".manifest.json"), "w"), indent=1)
    with gzip.GzipFile(os.path.join(HERE, name + ".expected.fa.gz"), "wb", mtime=0) as f:
        f.write(open(os.path.join(d, outfa), "rb").read())
    reg = []
    for c in range(nc):
        reg += regions(os.path.join(d, "aux", f"inspect_ctg{c + 1}.txt"))
    with gzip.GzipFile(os.path.join(HERE, name + ".regions.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(reg, separators=(",", ":")).encode())
    print(name, len(reg), "regions")


def main():
    if sys.argv[1] == "--messy":
        return main_messy()
    name, d, outfa, seed, G, k, size = sys.argv[1:8]
    is_long = "--long" in sys.argv
    nc = int(sys.argv[sys.argv.index("--contigs") + 1]) if "--contigs" in sys.argv else 1
    extra = sys.argv[sys.argv.index("--extra") + 1] if "--extra" in sys.argv else ""        # e.g. "-p 2"
    ins = ["draft.fa", "sr.sam", "aux/solid_kmers.bvsd"] + (["lr.sam"] if is_long else [])
    args = {"seed": int(seed), "G": int(G), "k": int(k), "long": is_long}
    if "--read-len" in sys.argv:                          # HiFi-like reads (gen_e2e.generate(read_len=, read_sub=))
        args["read_len"] = int(sys.argv[sys.argv.index("--read-len") + 1])
        args["read_sub"] = float(sys.argv[sys.argv.index("--read-sub") + 1])
    if nc > 1:
        args["contigs"] = nc
    man = {"generator": "tests/golden/gen_e2e.py", "args": args,
           "command": f"hypo -d draft.fa -r reads.fa -s {size} -c 30 -b sr.sam" + (" -B lr.sam" if is_long else "") + " -t 1 -i" + (" " + extra if extra else ""),
           "size_flag": size, "inputs_md5": {f: md5(os.path.join(d, f)) for f in ins},
           "expected_fasta_md5": md5(os.path.join(d, outfa))}
    json.dump(man, open(os.path.join(HERE, name + ".manifest.json"), "w"), indent=1)
    with gzip.GzipFile(os.path.join(HERE, name + ".expected.fa.gz"), "wb", mtime=0) as f:
        f.write(open(os.path.join(d, outfa), "rb").read())
    reg = []
    for c in range(nc):                                  # contigs in draft order, like the region dump of this repo's host
        reg += regions(os.path.join(d, "aux", f"inspect_ctg{c + 1}.txt"))
    with gzip.GzipFile(os.path.join(HERE, name + ".regions.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(reg, separators=(",", ":")).encode())
    print(name, len(reg), "regions")


if __name__ == "__main__":
    main()
