"""ctypes bindings of the TEST-ONLY CPU checkers (see oracle/hypo_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
  Oracle  -> oracle/_build/libhypo_oracle.so  (this repo's C restatement; built by oracle/Makefile)
  Ref     -> oracle/_ref/libhyporef.so        (the real reference classes; build container only)
"""
import ctypes as C
import os
import subprocess

import numpy as np

from hypo_amd import abi
from hypo_amd.batch import HostBatch

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libhypo_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libhyporef.so")

NW, LOV, ROV = 1, 3, 4


def build(ref: bool = True) -> None:
    subprocess.check_call(["make", "-s", "-C", HERE, "all" if ref else os.path.join(HERE, "_build", "libhypo_oracle.so")])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def batch_struct(b: HostBatch) -> abi.WindowBatch:
    s = abi.WindowBatch()
    s.n_windows = b.n_windows
    s.n_arms = b.n_arms
    s.windows = _ptr(b.windows)
    s.draft4 = _ptr(b.draft4)
    s.draft4_bytes = b.draft4.size
    s.arm_off = _ptr(b.arm_off)
    s.arm_len = _ptr(b.arm_len)
    s.arms2 = _ptr(b.arms2)
    s.arms2_bytes = b.arms2.size
    return s


class Oracle:
    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.oracle_poa_batch.restype = C.c_int
        self.lib.oracle_solid_scan.restype = C.c_int
        self.lib.oracle_replay.restype = C.c_int
        self.lib.oracle_num_threads.restype = C.c_int

    def num_threads(self) -> int:
        return int(self.lib.oracle_num_threads())

    def poa_batch(self, b: HostBatch, scores=abi.DEFAULT_SCORES, off=None, n_threads=0):
        """Returns (list of consensus strings, status array, dp_cells, n_alignments)."""
        sp = abi.ScoreParams(*scores)
        if off is None:
            off = b.slot_layout()
        n = b.n_windows
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        ins = batch_struct(b)
        out = abi.ConsensusBatch(_ptr(bases), _ptr(off), _ptr(ln), _ptr(st))
        cells = C.c_uint64(0)
        aligns = C.c_uint64(0)
        rc = self.lib.oracle_poa_batch(C.byref(sp), C.byref(ins), C.byref(out), C.c_int(n_threads),
                                       C.byref(cells), C.byref(aligns))
        if rc != 0:
            raise RuntimeError(f"oracle_poa_batch rc={rc}")
        cons = [bases[int(off[i]):int(off[i]) + int(ln[i])].tobytes().decode() if st[i] == 0 else None
                for i in range(n)]
        return cons, st, int(cells.value), int(aligns.value)

    def poa_batch_raw(self, b: HostBatch, scores=abi.DEFAULT_SCORES, off=None, n_threads=0):
        """Like poa_batch but returns the raw (bases, off, len, status) arrays."""
        sp = abi.ScoreParams(*scores)
        if off is None:
            off = b.slot_layout()
        n = b.n_windows
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        ins = batch_struct(b)
        out = abi.ConsensusBatch(_ptr(bases), _ptr(off), _ptr(ln), _ptr(st))
        cells = C.c_uint64(0)
        aligns = C.c_uint64(0)
        rc = self.lib.oracle_poa_batch(C.byref(sp), C.byref(ins), C.byref(out), C.c_int(n_threads),
                                       C.byref(cells), C.byref(aligns))
        if rc != 0:
            raise RuntimeError(f"oracle_poa_batch rc={rc}")
        return bases, off, ln, st, int(cells.value), int(aligns.value)

    def solid_scan(self, packed4: np.ndarray, n_bases: int, k: int, bits: np.ndarray, kids_cap=None):
        nw = (n_bases + 63) // 64
        words = np.zeros(max(nw, 1), dtype=np.uint64)
        if kids_cap is None:
            kids_cap = n_bases
        kids = np.zeros(max(kids_cap, 1), dtype=np.uint64)
        rank = np.zeros(nw + 1, dtype=np.uint64)
        ns = C.c_uint64(0)
        rc = self.lib.oracle_solid_scan(_ptr(packed4), C.c_uint64(n_bases), C.c_uint32(k), _ptr(bits),
                                        _ptr(words), _ptr(kids), C.c_uint64(kids_cap), _ptr(rank),
                                        C.byref(ns))
        if rc != 0:
            raise RuntimeError(f"oracle_solid_scan rc={rc}")
        n = int(ns.value)
        return words[:nw], kids[:min(n, kids_cap)], rank, n

    def replay(self, seqs, modes, scores=(5, -4, -8)):
        return _replay(self.lib.oracle_replay, seqs, modes, scores)


def _replay(fn, seqs, modes, scores):
    n = len(seqs)
    arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
    md = (C.c_int * n)(*modes)
    tot = sum(len(s) for s in seqs) + 8
    pairs = np.zeros(4 * tot, dtype=np.int32)
    rank = np.zeros(tot, dtype=np.int32)
    cons = np.zeros(tot, dtype=np.uint8)
    npairs, nn, cl = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = fn(C.c_int(scores[0]), C.c_int(scores[1]), C.c_int(scores[2]), C.c_int(n), arr, md,
            _ptr(pairs), C.c_int(2 * tot), C.byref(npairs), _ptr(rank), C.c_int(tot), C.byref(nn),
            _ptr(cons), C.c_int(tot), C.byref(cl))
    if rc != 0:
        return rc, None, None, None
    return 0, pairs[:2 * npairs.value].reshape(-1, 2).copy(), rank[:nn.value].copy(), \
        cons[:cl.value].tobytes().decode()


REF_ST_FILTERED = 0xF0      # hyporef_batch: a LONG window whose own arm filter dropped one of the arms it was handed (not comparable with the batch)


class Ref:
    """The real reference (hypo::Window + adapted spoa), SISD flavour."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (build it with `make -C oracle ref` where /root/reference exists)")
        self.lib = C.CDLL(path)
        self.lib.hyporef_new_engine.restype = C.c_int
        self.lib.hyporef_window.restype = C.c_int
        self.lib.hyporef_replay.restype = C.c_int
        self.lib.hyporef_pack_roundtrip.restype = C.c_int
        self._engines = {}

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def engine(self, scores) -> int:
        key = tuple(int(x) for x in scores)
        if key not in self._engines:
            sc = (C.c_int8 * 6)(*key)
            self._engines[key] = int(self.lib.hyporef_new_engine(sc))
        return self._engines[key]

    def window(self, w, scores=abi.DEFAULT_SCORES):
        """w: hypo_amd.TextWindow.  Returns (consensus, kept flags)."""
        e = self.engine(scores)
        mk = lambda xs: (C.c_char_p * max(len(xs), 1))(*[s.encode() for s in xs])
        narm = len(w.internal) + len(w.prefix) + len(w.suffix)
        cap = 4 * (len(w.draft) + sum(len(s) for s in w.internal + w.prefix + w.suffix)) + 64
        out = C.create_string_buffer(cap)
        kept = (C.c_ubyte * max(narm, 1))()
        r = self.lib.hyporef_window(C.c_int(e), C.c_int(1 if w.is_long else 0), w.draft.encode(),
                                    C.c_int(len(w.internal)), mk(w.internal),
                                    C.c_int(len(w.prefix)), mk(w.prefix),
                                    C.c_int(len(w.suffix)), mk(w.suffix),
                                    C.c_int(w.n_empty), out, C.c_int(cap), kept)
        if r < 0:
            raise RuntimeError("hyporef_window: output buffer too small")
        return out.raw[:r].decode(), list(kept)[:narm]

    def replay(self, seqs, modes, scores=(5, -4, -8)):
        return _replay(self.lib.hyporef_replay, seqs, modes, scores)

    def poa_batch_raw(self, b: HostBatch, scores=abi.DEFAULT_SCORES, off=None, n_threads=0):
        """The reference's POA phase (Window::generate_consensus in its OpenMP loop, src/Hypo.cpp:236-247) on a flattened
        batch.  Returns (bases, off, len, status, seconds of the consensus loop)."""
        if not hasattr(self.lib, "hyporef_batch"):
            raise RuntimeError("oracle/_ref/libhyporef.so predates hyporef_batch: rebuild it with `make -C oracle ref`")
        if off is None:
            off = b.slot_layout()
        n = b.n_windows
        bases = np.zeros(int(off[-1]) + 1, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8)
        ins = batch_struct(b)
        out = abi.ConsensusBatch(_ptr(bases), _ptr(off), _ptr(ln), _ptr(st))
        sec = C.c_double(0.0)
        sc = (C.c_int8 * 6)(*[int(x) for x in scores])
        rc = self.lib.hyporef_batch(sc, C.byref(ins), C.byref(out), C.c_int(n_threads), C.byref(sec))
        if rc != 0:
            raise RuntimeError(f"hyporef_batch rc={rc}")
        return bases, off, ln, st, float(sec.value)

    def pack_roundtrip(self, nb: int, text: str) -> str:
        out = C.create_string_buffer(len(text) + 8)
        r = self.lib.hyporef_pack_roundtrip(C.c_int(nb), text.encode(), out, C.c_int(len(text) + 8))
        return out.raw[:r].decode()


REF_SCAN_SO = os.path.join(HERE, "_ref", "libhyporef_scan.so")
REF_ARMS_SO = os.path.join(HERE, "_ref", "libhyporef_arms.so")


class RefScan:
    """The real Contig::find_solid_pos over the real suk::SolidKmers / sdsl bit vector (oracle/ref_scan_harness.cpp)."""

    def __init__(self, path: str = REF_SCAN_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (build it with `make -C oracle ref` where /root/reference exists)")
        self.lib = C.CDLL(path)
        self.lib.hyporef_solid_scan.restype = C.c_int

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SCAN_SO)

    def solid_scan(self, text: bytes, k: int, bits: np.ndarray):
        """text: ASCII contig; bits: 4^k-bit set as little-endian u64 words.  Returns (words, kids, rank, n_solid) like
        Oracle.solid_scan.  The bit set travels through a temporary .bvsd file because SolidKmers only exposes load()."""
        import tempfile
        n = len(text)
        nw = (n + 63) // 64
        words = np.zeros(max(nw, 1), dtype=np.uint64)
        kids = np.zeros(max(n, 1), dtype=np.uint64)
        rank = np.zeros(nw + 1, dtype=np.uint64)
        ns = C.c_uint64(0)
        with tempfile.NamedTemporaryFile(suffix=".bvsd", delete=False) as f:
            f.write(np.uint64(4 ** k).tobytes())
            f.write(np.ascontiguousarray(bits, dtype=np.uint64).tobytes())
            path = f.name
        try:
            rc = self.lib.hyporef_solid_scan(text, C.c_uint64(n), C.c_uint32(k), path.encode(), _ptr(words), _ptr(kids),
                                             C.c_uint64(n), _ptr(rank), C.byref(ns))
        finally:
            os.unlink(path)
        if rc != 0:
            raise RuntimeError(f"hyporef_solid_scan rc={rc}")
        return words[:nw], kids[:int(ns.value)], rank, int(ns.value)


_CIG = {c: i for i, c in enumerate("MIDNSHP=X")}


class RefArms:
    """The real short-read stage between "alignments loaded" and "windows filled" (src/Hypo.cpp:126-199) with the reference's own
    per-region dump as its output (oracle/ref_arms_harness.cpp)."""

    def __init__(self, path: str = REF_ARMS_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (build it with `make -C oracle ref` where /root/reference exists)")
        self.lib = C.CDLL(path)
        self.lib.hyporef_arms.restype = C.c_long

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_ARMS_SO)

    @staticmethod
    def sam_records(sam_path: str, contig: str, min_mapq: int = 2):
        """The records of `contig` the reference would turn into Alignment objects (flag and mapping-quality filter of
        src/Hypo.cpp:299-301), flattened: pos, cigar_off, cigar, seq_off, seq."""
        import re
        pos, coff, cig, soff, seq = [], [0], [], [0], []
        for line in open(sam_path):
            if line.startswith("@"):
                continue
            f = line.rstrip("\n").split("\t")
            if f[2] != contig or int(f[1]) & (4 | 256 | 512 | 1024) or int(f[4]) < min_mapq:
                continue
            pos.append(int(f[3]) - 1)
            for n, op in re.findall(r"(\d+)([MIDNSHP=X])", f[5]):
                cig.append(int(n) << 4 | _CIG[op])
            coff.append(len(cig))
            seq.append(f[9])
            soff.append(soff[-1] + len(f[9]))
        return (np.array(pos, dtype=np.uint32), np.array(coff, dtype=np.uint32), np.array(cig or [0], dtype=np.uint32),
                np.array(soff, dtype=np.uint64), "".join(seq).encode())

    _score_libs = {}

    def _lib_for_scores(self, scores, long_run=False):
        """The reference keeps its POA engines in statics that Window::prepare_for_poa only ever appends to (src/Window.cpp:28-42) and
        generate_consensus indexes from the front: a process that polishes with a SECOND score set would still use the first one's
        engines.  Every score set therefore runs in a private copy of the library (its own statics), like the long-read stage below."""
        # (... and Contig::set_no_long_reads() of a run without -B is sticky as well: runs with long reads get copies of their own)
        key = tuple(int(x) for x in scores) + (bool(long_run),)
        lib = RefArms._score_libs.get(key)
        if lib is None:
            import shutil
            import tempfile
            d = tempfile.mkdtemp(prefix="hyporef_scores_")
            dst = os.path.join(d, "libhyporef_arms_%d.so" % len(RefArms._score_libs))
            shutil.copy(REF_ARMS_SO, dst)
            lib = C.CDLL(dst)
            RefArms._score_libs[key] = lib
        return lib

    def fasta_file(self, draft_path: str, aln_path: str, k: int, bvsd_path: str, out_path: str, pick=None, min_mapq: int = 2,
                   scores=(5, -4, -8, 3, -5, -4), long_path=None, ned_th: int = 20, dump_dir=None) -> dict:
        """Whole files through the reference (rows T1 and N3 in place; oracle/ref_arms_harness.cpp, hyporef_fasta_bam / _sam): the
        short-read polish of the contigs `pick` (indices into the draft FASTA; None = all) with the records of a BAM or SAM file, decoded
        by the harness's own minimal reader and handed to the reference's Alignment constructor as bam1_t.  Returns the timers and counts
        of the reference's work (seconds: alignment objects, stage, POA loop, output; the decoder's own time apart)."""
        lib = self._lib_for_scores(scores, long_run=long_path is not None)
        bam = aln_path.endswith(".bam")
        if long_path is not None and not (bam and long_path.endswith(".bam")):
            raise RuntimeError("long reads (-B) go through the BAM entry point only")
        fn = getattr(lib, "hyporef_fasta_bam2" if bam else "hyporef_fasta_sam", None)
        if fn is None:
            raise RuntimeError("oracle/_ref/libhyporef_arms.so predates hyporef_fasta_bam2: make -C oracle ref")
        fn.restype = C.c_long
        pk = np.ascontiguousarray(sorted(set(int(x) for x in pick)) if pick is not None else [], dtype=np.uint32)
        sec = (C.c_double * 5)()
        cnt = (C.c_uint64 * 6)()
        # dump_dir: the reference's own per-region dump (type, arms, draft, consensus) of every polished contig goes to <dump_dir>/aux/inspect_<contig>.txt
        lib.hyporef_set_dump_dir.restype = None
        lib.hyporef_set_dump_dir(dump_dir.encode() if dump_dir else None)
        if bam:
            # (long_path: the long reads of a `-B` file through the SHORT-read constructor; rc -6 = a record the reference's NM filter would drop)
            rc = fn(draft_path.encode(), aln_path.encode(), long_path.encode() if long_path else None, C.c_uint32(k), bvsd_path.encode(), C.c_uint32(pk.size),
                    _ptr(pk) if pk.size else None, C.c_uint32(min_mapq), C.c_uint32(ned_th), out_path.encode(), (C.c_int8 * 6)(*scores), sec, cnt)
        else:
            rc = fn(draft_path.encode(), aln_path.encode(), C.c_uint32(k), bvsd_path.encode(), C.c_uint32(pk.size), _ptr(pk) if pk.size else None,
                    C.c_uint32(min_mapq), out_path.encode(), (C.c_int8 * 6)(*scores), sec, cnt)
        if rc < 0:
            raise RuntimeError(f"hyporef_fasta_{'bam' if aln_path.endswith('.bam') else 'sam'} rc={rc}")
        return {"contigs": int(cnt[0]), "draft_bases": int(cnt[1]), "alignments": int(cnt[2]), "invalid": int(cnt[3]), "regions": int(cnt[4]),
                "windows": int(cnt[5]), "decode_seconds": float(sec[0]), "alignment_object_seconds": float(sec[1]), "stage_seconds": float(sec[2]),
                "poa_seconds": float(sec[3]), "write_seconds": float(sec[4]), "threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))}

    def fasta(self, contig_seq: bytes, name: str, k: int, bvsd_path: str, records, out_path: str, scores=(5, -4, -8, 3, -5, -4)) -> int:
        """The polished FASTA record of one contig written by the reference's own `operator<<(Contig)` after its own short-read
        stage and its own Window::generate_consensus (hyporef_fasta, row A15 in place).  Returns the number of regions."""
        pos, coff, cig, soff, seq = records
        if not hasattr(self.lib, "hyporef_fasta"):
            raise RuntimeError("oracle/_ref/libhyporef_arms.so predates hyporef_fasta: make -C oracle ref")
        lib = self._lib_for_scores(scores)
        lib.hyporef_fasta.restype = C.c_long
        rc = lib.hyporef_fasta(contig_seq, C.c_uint64(len(contig_seq)), name.encode(), C.c_uint32(k), bvsd_path.encode(),
                                    C.c_uint32(len(pos)), _ptr(pos), _ptr(coff), _ptr(cig), _ptr(soff), seq, out_path.encode(),
                                    (C.c_int8 * 6)(*scores))
        if rc < 0:
            raise RuntimeError(f"hyporef_fasta rc={rc}")
        return int(rc)

    def regions_dump(self, contig_seq: bytes, k: int, bvsd_path: str, records, work_dir: str, long_records=None) -> str:
        """Runs the stage and returns the path of the reference's dump (aux/inspect_c.txt under work_dir).  long_records: the
        long reads of a `-B` run (same layout): the LONG-read stage follows (prepare_long_windows, find_long_arms, fill_long_windows
        with the real Filter); the caller guarantees none of them fails the reference's NM filter."""
        pos, coff, cig, soff, seq = records
        inv = C.c_uint64(0)
        if long_records is not None:
            # hypo::Contig::set_no_long_reads() (which the short-read entry point calls, as src/Hypo.cpp:230-232 does for a run without
            # -B) sets a static flag for the rest of the process and the reference has nothing that clears it: the long-read stage
            # runs in a second, private copy of the library (its own statics), whatever ran before in this process
            if getattr(self, "_long_lib", None) is None:
                import shutil
                import tempfile
                self._long_dir = tempfile.mkdtemp(prefix="hyporef_long_")
                dst = os.path.join(self._long_dir, "libhyporef_arms_long.so")
                shutil.copy(REF_ARMS_SO, dst)
                self._long_lib = C.CDLL(dst)
                self._long_lib.hyporef_arms_long.restype = C.c_long
            lpos, lcoff, lcig, lsoff, lseq = long_records
            rc = self._long_lib.hyporef_arms_long(contig_seq, C.c_uint64(len(contig_seq)), C.c_uint32(k), bvsd_path.encode(),
                                            C.c_uint32(len(pos)), _ptr(pos), _ptr(coff), _ptr(cig), _ptr(soff), seq,
                                            C.c_uint32(len(lpos)), _ptr(lpos), _ptr(lcoff), _ptr(lcig), _ptr(lsoff), lseq,
                                            work_dir.encode(), C.byref(inv))
            if rc < 0:
                raise RuntimeError(f"hyporef_arms_long rc={rc}")
            return os.path.join(work_dir, "aux", "inspect_c.txt")
        rc = self.lib.hyporef_arms(contig_seq, C.c_uint64(len(contig_seq)), C.c_uint32(k), bvsd_path.encode(),
                                   C.c_uint32(len(pos)), _ptr(pos), _ptr(coff), _ptr(cig), _ptr(soff), seq,
                                   work_dir.encode(), C.byref(inv))
        if rc < 0:
            raise RuntimeError(f"hyporef_arms rc={rc}")
        return os.path.join(work_dir, "aux", "inspect_c.txt")
