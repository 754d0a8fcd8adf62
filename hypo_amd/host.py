"""ctypes binding of libhypo_host.so: the C++ host mirror of the reference's object surface
(hypo_amd/csrc/host: PackedSeq, Filter, Window, Contig scan).  Used by tests; a C++ host links the mirror directly."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libhypo_host.so")


class HostMirror:
    def __init__(self):
        from . import capi
        capi.load_library()                   # torch + libhypo_gpu first: one HIP runtime per process
        self.lib = C.CDLL(LIB_PATH)

    @staticmethod
    def _arr(strings):
        return (C.c_char_p * max(len(strings), 1))(*[s.encode() for s in strings])

    def set_scores(self, scores):
        self.lib.hypo_host_set_scores((C.c_int8 * 6)(*scores))

    def filter(self, draft, arms):
        good = (C.c_ubyte * max(len(arms), 1))()
        self.lib.hypo_host_filter(draft.encode(), C.c_int(len(arms)), self._arr(arms), good)
        return [int(x) for x in good][:len(arms)]

    def read_fastx(self, path, line_by_line=False, cap=1 << 24):
        """[(name, sequence)] of a FASTA / FASTQ file through SeqIO.hpp's reader (the mapped, parallel one unless line_by_line)"""
        out = C.create_string_buffer(cap)
        r = self.lib.hypo_host_read_fastx(str(path).encode(), C.c_int(1 if line_by_line else 0), out, C.c_int(cap))
        if r < 0:
            return None
        return [tuple(l.split("\t")) for l in out.raw[:r].decode().split("\n")[:-1]]

    def pack_roundtrip(self, nb, text):
        out = C.create_string_buffer(len(text) + 8)
        r = self.lib.hypo_host_pack_roundtrip(C.c_int(nb), text.encode(), out, C.c_int(len(text) + 8))
        return out.raw[:r].decode()

    def windows(self, wins, batched=True):
        """wins: list of hypo_amd.TextWindow (LONG windows with UNFILTERED arms).  Returns (consensus list, kept flags)."""
        n = len(wins)
        arms = []
        for w in wins:
            arms += list(w.internal) + list(w.prefix) + list(w.suffix)
        iarr = lambda xs: (C.c_int * max(n, 1))(*xs)
        cap = 4 * sum(len(w.draft) for w in wins) + 2 * sum(len(a) for a in arms) + 1024
        out = C.create_string_buffer(cap)
        lens = (C.c_int * max(n, 1))()
        kept = (C.c_ubyte * max(len(arms), 1))()
        rc = self.lib.hypo_host_windows(C.c_int(n), iarr([1 if w.is_long else 0 for w in wins]),
                                        self._arr([w.draft for w in wins]), iarr([len(w.internal) for w in wins]),
                                        iarr([len(w.prefix) for w in wins]), iarr([len(w.suffix) for w in wins]),
                                        iarr([w.n_empty for w in wins]), self._arr(arms), out, C.c_long(cap), lens, kept,
                                        C.c_int(1 if batched else 0))
        if rc != 0:
            raise RuntimeError(f"hypo_host_windows rc={rc}")
        cons, o = [], 0
        for i in range(n):
            cons.append(out.raw[o:o + lens[i]].decode())
            o += lens[i]
        return cons, [int(x) for x in kept][:len(arms)]

    def contig_scan(self, seq, k, words, rank_q, sel_q):
        ns = C.c_uint64(0)
        kids = np.zeros(max(len(seq), 1), dtype=np.uint64)
        rq = np.asarray(rank_q, dtype=np.uint64); ra = np.zeros(max(rq.size, 1), dtype=np.uint64)
        sq = np.asarray(sel_q, dtype=np.uint64); sa = np.zeros(max(sq.size, 1), dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self.lib.hypo_host_contig_scan(seq.encode(), C.c_uint(k), p(words), C.c_uint64(words.size), C.byref(ns),
                                            p(kids), C.c_uint64(kids.size), p(rq), p(ra), C.c_int(rq.size), p(sq), p(sa),
                                            C.c_int(sq.size))
        if rc != 0:
            raise RuntimeError(f"hypo_host_contig_scan rc={rc}")
        n = int(ns.value)
        return n, kids[:n], ra[:rq.size], sa[:sq.size]
