"""CPU: the product library builds for gfx950, loads, and exports every symbol include/hypo_gpu.h
declares.  No compute call is made here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

from hypo_amd import abi, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        capi.build_library()
    return capi.load_library()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hypo_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hypo_gpu_[a-z_0-9]+)\s*\(", text)))


def test_exports_match_header(lib):
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"libhypo_gpu.so does not export {n}"
    assert sorted(capi.EXPORTS) == names


def test_abi_version_and_struct_sizes(lib):
    assert lib.hypo_gpu_abi_version() == abi.ABI_VERSION
    assert C.sizeof(abi.Window) == 40 and abi.WINDOW_DTYPE.itemsize == 40
    assert C.sizeof(abi.ScoreParams) == 6
    assert C.sizeof(abi.WindowBatch) == 64
    assert C.sizeof(abi.ConsensusBatch) == 32


def test_fails_loudly_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.hypo_gpu_init(None, 0) == abi.HYPO_E_NODEVICE
    assert b"no HIP device" in lib.hypo_gpu_last_error()
    sp = abi.ScoreParams(*abi.DEFAULT_SCORES)
    ins, out = abi.WindowBatch(), abi.ConsensusBatch()
    assert lib.hypo_gpu_poa_batch(C.byref(sp), C.byref(ins), C.byref(out)) == abi.HYPO_E_NOTINIT
    with pytest.raises(capi.HypoGpuError):
        capi.HypoGpu(0)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(capi.HypoGpuError):
        capi.load_library(str(tmp_path / "nope.so"))


def test_build_id_matches_sources(lib):
    """hypo_gpu_build_id() = first 16 hex digits of sha256 over the library's sources (sorted paths), embedded by the
    Makefile: a prebuilt libhypo_gpu.so that does not come from the sources in the tree fails here."""
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "hypo_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) +
                   [os.path.join(csrc, "..", "..", "include", "hypo_gpu.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    assert lib.hypo_gpu_build_id().decode() == h.hexdigest()[:16]


def test_library_asks_for_hardware_queues_when_loaded():
    """the host does not have to set GPU_MAX_HW_QUEUES; a value it does set is kept."""
    import subprocess
    import sys
    if not os.path.exists(capi.LIB_PATH):
        capi.build_library()
    code = ("import ctypes, os, sys\n"
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "before = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "ctypes.CDLL(sys.argv[1])\n"
            "print(before, libc.getenv(b'GPU_MAX_HW_QUEUES'))\n")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code, capi.LIB_PATH], env=env, capture_output=True, text=True, check=True).stdout
    assert out.strip() == "None b'8'", out
    out = subprocess.run([sys.executable, "-c", code, capi.LIB_PATH], env=dict(env, GPU_MAX_HW_QUEUES="2"), capture_output=True, text=True, check=True).stdout
    assert out.strip() == "b'2' b'2'", out
