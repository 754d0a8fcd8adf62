"""CPU: the scan restatement (oracle_solid_scan) and the host-side packing against G3 — the outputs of the REAL
hypo::Contig::find_solid_pos recorded in tests/golden/scan_cases.json.gz (made by tests/golden/make_scan_golden.py)."""
import base64
import gzip

import numpy as np

import golden_util as gu


def _arr(s, dtype=np.uint64):
    return np.frombuffer(gzip.decompress(base64.b64decode(s)), dtype=dtype)


def scan_cases():
    lut = np.full(256, 4, np.uint8)
    for c, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[c] = v
    for c in gu.load_json("scan_cases.json.gz")["cases"]:
        codes = lut[np.frombuffer(c["contig"].encode(), dtype=np.uint8)]
        pad = np.concatenate([codes, np.zeros((-codes.size) % 2, np.uint8)]).reshape(-1, 2)
        p4 = ((pad[:, 0] << 4) | pad[:, 1]).astype(np.uint8)
        if p4.size == 0:
            p4 = np.zeros(1, np.uint8)
        if "bitset" in c:
            bits = _arr(c["bitset"]).copy()
        else:                                   # sparse sets are stored as the sorted list of set k-mer ids
            ids = _arr(c["bitset_ids"], np.uint32).astype(np.uint64)
            bits = np.zeros(max((1 << (2 * c["k"])) // 64, 1), dtype=np.uint64)
            np.bitwise_or.at(bits, (ids >> np.uint64(6)).astype(np.int64), np.uint64(1) << (ids & np.uint64(63)))
        yield c, p4, bits, _arr(c["words"]), _arr(c["kids"]), _arr(c["rank"])


def test_oracle_scan_vs_reference_fixture(oracle_lib):
    n_cases = 0
    for c, p4, bits, words, kids, rank in scan_cases():
        ow, okids, orank, ons = oracle_lib.solid_scan(p4, c["n"], c["k"], bits)
        assert ons == c["n_solid"], (c["n"], c["k"])
        assert (ow == words[:ow.size]).all() and (okids == kids).all() and (orank == rank).all(), (c["n"], c["k"])
        n_cases += 1
    assert n_cases >= 20
