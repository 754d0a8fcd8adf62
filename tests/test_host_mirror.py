"""The C++ host mirror of the reference's object surface (hypo_amd/csrc/host): PackedSeq and Filter on the
CPU against goldens of the real reference; Window / Contig through the device on the GPU box."""
import numpy as np
import pytest

from hypo_amd import abi, host, sim
from hypo_amd.batch import TextWindow
import golden_util as gu


@pytest.fixture(scope="module")
def mirror():
    return host.HostMirror()


@pytest.fixture(scope="module")
def device():
    """hypo_gpu_init for the tests whose C++ mirror calls the device (libhypo_host.so shares the process-wide context)."""
    from hypo_amd import capi
    return capi.HypoGpu(0)


def test_packedseq_roundtrip_goldens(mirror):
    for c in gu.load_json("packedseq_cases.json.gz"):
        assert mirror.pack_roundtrip(c["nb"], c["text"]) == c["unpacked"]


def test_filter_matches_reference_kept_flags(mirror):
    """Long-arm minimizer filter (include/Filter.hpp) against the kept flags the real Window recorded."""
    n = 0
    for r in gu.load_jsonl("windows_synth.jsonl.gz"):
        if not r["long"]:
            continue
        arms = r["internal"] + r["prefix"] + r["suffix"]
        assert mirror.filter(r["draft"], arms) == r["kept"], r["tag"]
        n += len(arms)
    assert n > 100


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [True, False])
def test_window_surface_goldens(mirror, device, batched):
    for name in gu.WINDOW_FILES:
        for scores, _ in gu.windows_by_scores(name).items():
            recs = [r for r in gu.load_jsonl(name) if tuple(r["scores"]) == scores]
            recs = recs if batched else recs[:40]
            wins = [gu.to_window(r, filtered=False) for r in recs]        # LONG: unfiltered arms, the mirror filters
            mirror.set_scores(scores)
            cons, kept = mirror.windows(wins, batched=batched)
            assert cons == [r["consensus"] for r in recs], (name, scores)
            k = 0
            for r in recs:
                narm = len(r["internal"]) + len(r["prefix"]) + len(r["suffix"])
                if r.get("kept") is not None:
                    assert kept[k:k + narm] == r["kept"]
                k += narm
    mirror.set_scores(abi.DEFAULT_SCORES)


@pytest.mark.gpu
def test_contig_scan_rank_select(mirror, device, oracle_lib):
    codes, p4 = sim.random_contig(100001, seed=4, n_frac=0.001)
    k = 11
    bits = sim.solid_bitset(codes, k)
    seq = np.frombuffer(b"ACGTN", dtype=np.uint8)[codes].tobytes().decode()
    ow, okids, orank, ons = oracle_lib.solid_scan(p4, codes.size, k, bits)
    pos = np.flatnonzero(np.unpackbits(ow.view(np.uint8), bitorder="little")[:codes.size])
    rq = [0, 1, 63, 64, 65, 5000, 100000, 100001]
    sq = [1, 2, ons // 2, ons]
    n, kids, ra, sa = mirror.contig_scan(seq, k, bits, rq, sq)
    assert n == ons and (kids == okids).all()
    assert ra.tolist() == [int((pos < q).sum()) for q in rq]
    assert sa.tolist() == [int(pos[i - 1]) for i in sq]


@pytest.mark.gpu
def test_window_beyond_the_table_driven_classes_is_polished(mirror, device, oracle_lib, capfd):
    """Rounds 2-5: a window with an arm longer than 1 021 bases kept its draft with a warning (HYPO_ST_CAPACITY).  Since round 6 it runs in
    size class 6 (hypo_amd/csrc/poa_giant.hpp) and the host mirror gets the reference's consensus for it, its neighbours as before; nothing
    is "kept unpolished"."""
    import random
    from hypo_amd.batch import build_batch
    rng = random.Random(5)
    truth = "".join(rng.choice("ACGT") for _ in range(40))
    draft = truth[:10] + "A" + truth[11:]
    deep = TextWindow(draft, [truth] * 1100)
    long_arm = TextWindow(draft, [truth] * 4 + ["".join(rng.choice("ACGT") for _ in range(1100))])
    normal = TextWindow(draft, [truth] * 8)
    wins = [normal, deep, long_arm, normal]
    cons, _ = mirror.windows(wins, batched=True)
    assert cons[0] == truth and cons[3] == truth
    assert cons[1] == truth                                   # deep, polished in the last table-driven class
    assert cons[2] == oracle_lib.poa_batch(build_batch([long_arm]))[0][0]
    assert "kept unpolished" not in capfd.readouterr().err


def test_draft_reader_mapped_and_line_by_line_agree(mirror, tmp_path):
    """The draft is read on all threads out of the mapped file (read_fasta_mapped: record starts found chunk by chunk) or line by line
    (gzip, FASTQ, pipes): the same records for wrapped sequences, CR LF line ends, empty lines, a header with a description, a record
    without a sequence, no line feed at the end — and what klib's kseq gives the reference for such files (src/Hypo.cpp:82-95)."""
    import random
    rng = random.Random(5)
    seq = lambda n: "".join(rng.choice("ACGTNacgt") for _ in range(n))
    recs = [("c1", seq(1000)), ("c2", seq(61)), ("empty", ""), ("c4", seq(5_000_00)), ("c5", seq(7))]

    def write(path, eol, wrap, blank, final_eol):
        with open(path, "wb") as f:
            f.write((eol * 2).encode())                                    # empty lines in front
            for i, (name, s) in enumerate(recs):
                f.write((">" + name + (" some description\tmore" if i % 2 else "") + eol).encode())
                lines = [s[j:j + wrap] for j in range(0, len(s), wrap)] if wrap else ([s] if s else [])
                for k, ln in enumerate(lines):
                    last = i == len(recs) - 1 and k == len(lines) - 1
                    f.write((ln + ("" if last and not final_eol else eol)).encode())
                    if blank and k % 3 == 1:
                        f.write(eol.encode())
    n = 0
    for eol in ("\n", "\r\n"):
        for wrap in (0, 60, 1):
            for blank in (False, True):
                for final_eol in (True, False):
                    if wrap == 1 and (blank or not final_eol):
                        continue
                    p = tmp_path / f"d{n}.fa"
                    write(p, eol, wrap, blank, final_eol)
                    a = mirror.read_fastx(p, line_by_line=False)
                    b = mirror.read_fastx(p, line_by_line=True)
                    assert a == b == [(nm, s) for nm, s in recs], (eol, wrap, blank, final_eol)
                    n += 1
    assert n >= 16
    fq = tmp_path / "r.fq"
    fq.write_text("@r1 x\nACGT\n+\nIIII\n@r2\nGG\n+\nII\n")
    assert mirror.read_fastx(fq) == mirror.read_fastx(fq, line_by_line=True) == [("r1", "ACGT"), ("r2", "GG")]
