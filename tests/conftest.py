import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_unavailable_reason():
    """None when libhypo_gpu.so loads and sees a HIP device, else why not (checked once per session)."""
    try:
        import ctypes
        so = os.path.join(ROOT, "hypo_amd", "_build", "libhypo_gpu.so")
        if not os.path.exists(so):
            return "hypo_amd/_build/libhypo_gpu.so is not built"
        ctypes.CDLL(so)
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        rc = hip.hipGetDeviceCount(ctypes.byref(n))
        if rc != 0 or n.value < 1:
            return "no HIP device visible"
    except OSError as e:  # pragma: no cover - depends on the box
        return "cannot load the HIP runtime / libhypo_gpu.so: %s" % e
    return None


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on the GPU box must FAIL loudly when the device or the library is missing (the driver runs exactly that); a plain
    # `pytest tests` on a CPU box skips the gpu-marked tests instead of drowning the CPU results in errors.
    if "gpu" in (config.getoption("-m") or ""):
        return
    reason = None
    checked = False
    for item in items:
        if "gpu" in item.keywords:
            if not checked:
                reason, checked = _gpu_unavailable_reason(), True
            if reason:
                item.add_marker(pytest.mark.skip(reason="gpu test: " + reason))


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def ref_lib():
    import oracle
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref not built (the real reference only exists in the build container)")
    return oracle.Ref()
