"""hypo_amd — MI355X-native implementation of HyPo's per-window polishing hot path.

Python here is plumbing (ctypes over the C-ABI in include/hypo_gpu.h, torch for device memory,
streams and torch.distributed); the product is hypo_amd/csrc (HIP kernels + C-ABI + C++ host mirror).
"""
from . import abi  # noqa: F401
from .batch import TextWindow, HostBatch, build_batch  # noqa: F401

__all__ = ["abi", "TextWindow", "HostBatch", "build_batch"]
