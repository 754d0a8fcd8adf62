"""Host-side flattening of Window objects into the C-ABI batch layout (include/hypo_gpu.h).

Packing is byte-identical to the reference's PackedSeq<NB> (src/PackedSeq.cpp:58-89): MSB-first,
4-bit codes A0 C1 G2 T3 other=4 for drafts, 2-bit codes for arms; every sequence starts on a byte
boundary.  This is plumbing for tests / bench; the C++ host mirror does the same from PackedSeq data.
"""
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from . import abi

_NT4 = np.full(256, 4, dtype=np.uint8)
for _c, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
    _NT4[ord(_c)] = _v
    _NT4[ord(_c.lower())] = _v
_NT4[0:4] = (0, 1, 2, 3)  # cNt4Table row 0, include/globalDefs.hpp:160-178


def pack2(text: str) -> np.ndarray:
    """PackedSeq<2>(std::string) bytes."""
    codes = _NT4[np.frombuffer(text.encode("ascii"), dtype=np.uint8)]
    if codes.size and codes.max() > 3:
        raise ValueError("non-ACGT base cannot be packed in 2 bits (PackedSeq.cpp:75-79)")
    pad = (-codes.size) % 4
    c = np.concatenate([codes, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return ((c[:, 0] << 6) | (c[:, 1] << 4) | (c[:, 2] << 2) | c[:, 3]).astype(np.uint8)


def pack4(text: str) -> np.ndarray:
    """PackedSeq<4>(std::string) bytes."""
    codes = _NT4[np.frombuffer(text.encode("ascii"), dtype=np.uint8)]
    pad = (-codes.size) % 2
    c = np.concatenate([codes, np.zeros(pad, np.uint8)]).reshape(-1, 2)
    return ((c[:, 0] << 4) | c[:, 1]).astype(np.uint8)


def unpack2(data: np.ndarray, n: int) -> str:
    d = np.asarray(data, dtype=np.uint8)
    c = np.stack([(d >> 6) & 3, (d >> 4) & 3, (d >> 2) & 3, d & 3], axis=1).reshape(-1)[:n]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[c].tobytes().decode()


def unpack4(data: np.ndarray, n: int) -> str:
    d = np.asarray(data, dtype=np.uint8)
    c = np.stack([(d >> 4) & 15, d & 15], axis=1).reshape(-1)[:n]
    c = np.minimum(c, 4)
    return np.frombuffer(b"ACGTN", dtype=np.uint8)[c].tobytes().decode()


@dataclass
class TextWindow:
    """A hypo::Window as text (include/Window.hpp:123-135)."""
    draft: str
    internal: List[str] = field(default_factory=list)
    prefix: List[str] = field(default_factory=list)
    suffix: List[str] = field(default_factory=list)
    n_empty: int = 0
    is_long: bool = False


@dataclass
class HostBatch:
    """numpy arrays laid out exactly as HypoWindowBatch expects (host memory)."""
    windows: np.ndarray      # WINDOW_DTYPE [n_windows]
    draft4: np.ndarray       # u8
    arm_off: np.ndarray      # u64 [n_arms]
    arm_len: np.ndarray      # u32 [n_arms]
    arms2: np.ndarray        # u8

    @property
    def n_windows(self) -> int:
        return int(self.windows.shape[0])

    @property
    def n_arms(self) -> int:
        return int(self.arm_len.shape[0])

    def slot_layout(self) -> np.ndarray:
        """Same rule as hypo_gpu_poa_slot_layout: 1.5*max(draft, longest arm)+24 rounded up to 8."""
        n = self.n_windows
        longest = self.windows["draft_len"].astype(np.int64).copy()
        if self.n_arms:
            w = self.windows
            cnt = (w["n_internal"] + w["n_prefix"] + w["n_suffix"]).astype(np.int64)
            owner = np.repeat(np.arange(n), cnt)
            # arms are stored contiguously per window in first_arm order in batches built here
            first = w["first_arm"].astype(np.int64)
            idx = np.concatenate([np.arange(f, f + c) for f, c in zip(first, cnt)]) if n else np.zeros(0, np.int64)
            if idx.size:
                np.maximum.at(longest, owner, self.arm_len[idx].astype(np.int64))
        slot = (longest + longest // 2 + 24 + 7) // 8 * 8
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(slot, out=off[1:])
        return off

    def algorithmic_bytes(self, cons_len: np.ndarray) -> int:
        """A_poa = ceil(Ld/2) + sum ceil(La/4) + Lcons + 16 + 8*(1+n_arms) per window."""
        w = self.windows
        narm = (w["n_internal"] + w["n_prefix"] + w["n_suffix"]).astype(np.int64)
        a = ((w["draft_len"].astype(np.int64) + 1) // 2).sum()
        a += ((self.arm_len.astype(np.int64) + 3) // 4).sum()
        a += int(np.asarray(cons_len, dtype=np.int64).sum())
        a += 16 * self.n_windows + 8 * int((1 + narm).sum())
        return int(a)


def _pack_many(texts: Sequence[str], bases_per_byte: int):
    """Packs many sequences at once, each starting on a byte boundary.  Returns (bytes, byte_off, len)."""
    n = len(texts)
    lens = np.fromiter((len(t) for t in texts), dtype=np.int64, count=n)
    nbytes = (lens + bases_per_byte - 1) // bases_per_byte
    byte_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nbytes, out=byte_off[1:])
    codes = _NT4[np.frombuffer("".join(texts).encode("ascii"), dtype=np.uint8)]
    if bases_per_byte == 4 and codes.size and codes.max() > 3:
        raise ValueError("non-ACGT base cannot be packed in 2 bits (PackedSeq.cpp:75-79)")
    starts = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    pos = np.arange(codes.size, dtype=np.int64) - np.repeat(starts[:-1], lens)
    dest = np.repeat(byte_off[:-1], lens) * bases_per_byte + pos
    slots = np.zeros(int(byte_off[-1]) * bases_per_byte, dtype=np.uint8)
    slots[dest] = codes
    c = slots.reshape(-1, bases_per_byte)
    if bases_per_byte == 4:
        out = (c[:, 0] << 6) | (c[:, 1] << 4) | (c[:, 2] << 2) | c[:, 3]
    else:
        out = (c[:, 0] << 4) | c[:, 1]
    return out.astype(np.uint8), byte_off[:-1].astype(np.uint64), lens.astype(np.uint32)


def build_batch(windows: Sequence[TextWindow]) -> HostBatch:
    n = len(windows)
    wd = np.zeros(n, dtype=abi.WINDOW_DTYPE)
    arms: List[str] = []
    first = np.zeros(n, dtype=np.uint32)
    for i, w in enumerate(windows):
        first[i] = len(arms)
        arms.extend(w.internal)
        arms.extend(w.prefix)
        arms.extend(w.suffix)
    draft4, doff, dlen = _pack_many([w.draft for w in windows], 2)
    arms2, aoff, alen = _pack_many(arms, 4)
    wd["type"] = [abi.WIN_LONG if w.is_long else abi.WIN_SHORT for w in windows]
    wd["draft_len"] = dlen
    wd["draft_off"] = doff
    wd["first_arm"] = first
    wd["n_internal"] = [len(w.internal) for w in windows]
    wd["n_prefix"] = [len(w.prefix) for w in windows]
    wd["n_suffix"] = [len(w.suffix) for w in windows]
    wd["n_empty"] = [w.n_empty for w in windows]
    return HostBatch(wd, draft4, aoff, alen, arms2)
