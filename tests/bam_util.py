"""Minimal SAM-text -> BAM converter for tests (SAM spec sections 4.1 BGZF, 4.2 BAM): the `hypo` binary must give the
same FASTA whether its alignments arrive as SAM text or as BAM.  Only what the goldens' SAM files contain is encoded
(header @SQ lines, 11 mandatory fields, NM:i tags); qualities are written as 0xff ('*')."""
import struct
import zlib

_OPS = "MIDNSHP=X"
_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25                                       # total block size - 1
    return (b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def sam_to_bam(sam_path, bam_path, nm_type="i", extra_tags=False):
    text, refs, recs = [], [], []
    for line in open(sam_path):
        if line.startswith("@"):
            text.append(line)
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                refs.append((f["SN"], int(f["LN"])))
        elif line.strip():
            recs.append(line.rstrip("\n").split("\t"))
    tid = {n: i for i, (n, _) in enumerate(refs)}
    htxt = "".join(text).encode()
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(htxt)) + htxt + struct.pack("<i", len(refs)))
    for n, l in refs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\x00" + struct.pack("<i", l)
    for f in recs:
        qname, flag, rname, pos, mapq, cigar, seq = f[0], int(f[1]), f[2], int(f[3]) - 1, int(f[4]), f[5], f[9]
        ops, num, reflen = [], "", 0
        if cigar != "*":
            for ch in cigar:
                if ch.isdigit():
                    num += ch
                else:
                    n = int(num); num = ""
                    ops.append((n << 4) | _OPS.index(ch))
                    if ch in "MDN=X":
                        reflen += n
        seqb = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            seqb[i >> 1] |= _NT16.get(c, 15) << (4 if i % 2 == 0 else 0)
        tags = b""
        if extra_tags:                                            # one of every value type in front of NM: the reader must step over them
            tags += b"XAAx" + b"XZZsome text\x00" + b"XHH1AE301\x00" + b"Xff" + struct.pack("<f", 1.5) + b"Xcc" + struct.pack("<b", -3)
            tags += b"XBBs" + struct.pack("<i", 3) + struct.pack("<3h", 1, -2, 3) + b"XCBC" + struct.pack("<i", 2) + b"\x01\x02"
            tags += b"XIBI" + struct.pack("<i", 1) + struct.pack("<I", 7) + b"XSS" + struct.pack("<H", 65535) + b"Xii" + struct.pack("<i", -9)
        for t in f[11:]:
            if t.startswith("NM:i:"):
                v = int(t[5:])
                tags += b"NM" + {"i": b"i" + struct.pack("<i", v), "C": b"C" + struct.pack("<B", v & 0xff), "S": b"S" + struct.pack("<H", v & 0xffff)}[nm_type if v < (256 if nm_type == "C" else 65536) else "i"]
        body = struct.pack("<iiBBHHHiiii", tid.get(rname, -1), pos, len(qname) + 1, mapq, _reg2bin(pos, pos + max(reflen, 1)), len(ops), flag,
                           len(seq), -1, -1, 0)
        body += qname.encode() + b"\x00" + b"".join(struct.pack("<I", o) for o in ops) + bytes(seqb) + b"\xff" * len(seq) + tags
        out += struct.pack("<i", len(body)) + body
    with open(bam_path, "wb") as fh:
        for i in range(0, len(out), 0xff00):
            fh.write(_bgzf_block(bytes(out[i:i + 0xff00])))
        fh.write(_bgzf_block(b""))                                # BGZF end-of-file marker
