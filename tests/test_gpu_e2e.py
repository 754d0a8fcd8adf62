"""End to end on the MI355X: the `hypo` binary of this repo (host pipeline + libhypo_gpu.so) reproduces the REAL
reference's polished FASTA byte for byte, and its per-region arms and consensus, on the four committed goldens."""
import os

import pytest
import e2e_util as eu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", eu.CASES)
def test_e2e_gpu_matches_reference(name, tmp_path):
    if not os.path.exists(eu.BIN):           # the prebuilt binary travels with the snapshot; build only if it is absent
        eu.build_binary()
    man, p = eu.run_case(name, tmp_path, "gpu", threads=16)
    assert eu.check_outputs(name, tmp_path, man) > 0
    assert "POA of windows" in p.stdout
