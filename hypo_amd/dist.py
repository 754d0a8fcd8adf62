"""Multi-GPU plumbing of the path: windows are independent, so a batch is sharded over
ranks (one process per GPU) by estimated cost, each rank polishes its shard locally, and ONE exchange step
— an all-gather of consensus lengths and bytes over RCCL (backend "nccl" on ROCm) / gloo in CPU tests —
gives every rank the consensus of every window for contig re-assembly (src/Contig.cpp:345-366).
No collective sits inside the data path itself."""
from typing import List, Sequence, Tuple

import numpy as np


def window_costs(windows: np.ndarray, arm_len: np.ndarray) -> np.ndarray:
    """c_w = (L_w + 2) * sum_arms (L_arm + 1), SURVEY.md §8e (x2 for LONG windows)."""
    n = windows.shape[0]
    narm = (windows["n_internal"] + windows["n_prefix"] + windows["n_suffix"]).astype(np.int64)
    owner = np.repeat(np.arange(n), narm)
    first = windows["first_arm"].astype(np.int64)
    idx = np.repeat(first, narm) + (np.arange(int(narm.sum())) - np.repeat(np.cumsum(narm) - narm, narm))
    s = np.bincount(owner, weights=arm_len[idx].astype(np.float64) + 1.0, minlength=n)
    c = (windows["draft_len"].astype(np.float64) + 2.0) * s
    return np.where(windows["type"] != 0, 2.0 * c, c)


def shard_contiguous(costs: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous, cost-balanced ranges [begin, end) per rank (keeps contig locality)."""
    n = costs.shape[0]
    csum = np.concatenate([[0.0], np.cumsum(costs)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csum, total * r / world)))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.minimum(cuts, n))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def take_windows(batch, begin: int, end: int, compact: bool = False):
    """Sub-batch of windows [begin, end) (first_arm re-based).  compact=False shares the parent's packed buffers;
    compact=True also cuts draft4 / arms2 down to the range's bytes and re-bases draft_off / arm_off (what a rank uploads
    when one batch is sharded over several GPUs).  Assumes the layout the simulator and the host pipeline produce: windows,
    arms and packed bytes in the same order."""
    from .batch import HostBatch
    w = batch.windows[begin:end].copy()
    if end > begin:
        a0 = int(w["first_arm"][0])
        last = w[-1]
        a1 = int(last["first_arm"]) + int(last["n_internal"]) + int(last["n_prefix"]) + int(last["n_suffix"])
    else:
        a0 = a1 = 0
    w["first_arm"] -= a0
    arm_off, arm_len = batch.arm_off[a0:a1].copy(), batch.arm_len[a0:a1].copy()
    if not compact or end <= begin:
        return HostBatch(w, batch.draft4, arm_off, arm_len, batch.arms2)
    d0 = int(w["draft_off"][0])
    d1 = int(w["draft_off"][-1]) + (int(w["draft_len"][-1]) + 1) // 2
    w["draft_off"] -= d0
    if a1 > a0:
        b0 = int(arm_off[0])
        b1 = int(arm_off[-1]) + (int(arm_len[-1]) + 3) // 4
        arm_off -= np.uint64(b0)
    else:
        b0 = b1 = 0
    return HostBatch(w, batch.draft4[d0:d1].copy(), arm_off, arm_len, batch.arms2[b0:b1].copy())


def gather_consensus(bases, lens, max_bytes: int, max_windows: int, group=None):
    """all-gather of one rank's consensus buffer (torch uint8 tensor `bases`, int32 `lens`).
    Every rank contributes buffers padded to (max_bytes, max_windows) — sizes agreed beforehand with one
    all_reduce(MAX) — so the exchange is two fixed-size collectives per batch.
    Returns (all_bases [world, max_bytes], all_lens [world, max_windows])."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = bases.device
    sb = torch.zeros(max_bytes, dtype=torch.uint8, device=dev)
    sb[:min(bases.numel(), max_bytes)].copy_(bases[:max_bytes])
    sl = torch.zeros(max_windows, dtype=torch.int32, device=dev)
    sl[:lens.numel()].copy_(lens)
    ab = torch.empty(world * max_bytes, dtype=torch.uint8, device=dev)
    al = torch.empty(world * max_windows, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(al, sl, group=group)
    dist.all_gather_into_tensor(ab, sb, group=group)
    return ab.view(world, max_bytes), al.view(world, max_windows)


class ConsensusExchange:
    """The same exchange with nothing allocated or filled per batch: one send buffer [lens as bytes | consensus bytes],
    one receive buffer, ONE all-gather per batch (the two collectives of gather_consensus cost their launch latency twice).
    Sizes come from agree_sizes.  gather() returns views into the receive buffer, valid until the next gather()."""

    def __init__(self, max_bytes: int, max_windows: int, device, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.max_bytes, self.max_windows = max_bytes, max_windows
        self.len_bytes = (4 * max_windows + 255) // 256 * 256
        self.slot = self.len_bytes + max_bytes
        self.send = torch.zeros(self.slot, dtype=torch.uint8, device=device)
        self.recv = torch.empty(self.world * self.slot, dtype=torch.uint8, device=device)

    def gather(self, bases, lens):
        import torch
        import torch.distributed as dist
        nb = min(bases.numel(), self.max_bytes)
        self.send[self.len_bytes:self.len_bytes + nb].copy_(bases[:nb])
        self.send[:4 * lens.numel()].copy_(lens.contiguous().view(torch.uint8))
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        r = self.recv.view(self.world, self.slot)
        return r[:, self.len_bytes:], r[:, :4 * self.max_windows].contiguous().view(torch.int32).view(self.world, self.max_windows)


def agree_sizes(n_bytes: int, n_windows: int, device, group=None) -> Tuple[int, int]:
    import torch
    import torch.distributed as dist
    t = torch.tensor([n_bytes, n_windows], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return (int(t[0].item()) + 255) // 256 * 256, int(t[1].item())


def reassemble(all_bases: np.ndarray, all_lens: np.ndarray, offs: Sequence[np.ndarray],
               ranges: Sequence[Tuple[int, int]]) -> List[str]:
    """Consensus strings of all windows in global window order from the gathered per-rank buffers."""
    out: List[str] = []
    for r, (b, e) in enumerate(ranges):
        off = offs[r]
        for i in range(e - b):
            a = int(off[i])
            out.append(all_bases[r, a:a + int(all_lens[r, i])].tobytes().decode())
    return out
