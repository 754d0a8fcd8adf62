"""CPU, world_size 2 on gloo: the N > 1 path — cost-balanced sharding of a window batch over ranks, the
all-gather exchange of consensus bytes/lengths, and re-assembly in global window order.  The per-rank
polishing is done by the oracle here (no GPU in this container); the exchange code is the one bench.py and a
multi-GPU host use with the nccl (RCCL) backend."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hypo_amd import dist as hd
from hypo_amd import sim


def offs_of(batch, ranges, r):
    return hd.take_windows(batch, *ranges[r]).slot_layout()


def _worker(rank, world, port, n_windows, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    orc = oracle.Oracle()
    batch = sim.window_batch(n_windows, seed=5)                       # same batch on every rank
    ranges = hd.shard_contiguous(hd.window_costs(batch.windows, batch.arm_len), world)
    b, e = ranges[rank]
    mine = hd.take_windows(batch, b, e)
    off = mine.slot_layout()
    bases, _, ln, st, _, _ = orc.poa_batch_raw(mine, off=off, n_threads=2)
    assert (st == 0).all()
    dev = torch.device("cpu")
    max_bytes, max_windows = hd.agree_sizes(int(off[-1]), mine.n_windows, dev)
    ab, al = hd.gather_consensus(torch.from_numpy(bases), torch.from_numpy(ln.view(np.int32)), max_bytes, max_windows)
    # the preallocated single-collective form bench.py uses must hand back the same bytes, batch after batch
    ex = hd.ConsensusExchange(max_bytes, max_windows, dev)
    for _ in range(2):
        xb, xl = ex.gather(torch.from_numpy(bases), torch.from_numpy(ln.view(np.int32)))
        assert torch.equal(xl, al)
        for r in range(world):
            nbytes = int(offs_of(batch, ranges, r)[-1])
            assert torch.equal(xb[r, :nbytes], ab[r, :nbytes])
    # every rank needs the slot layouts of the other ranks to cut the gathered buffers: they are a pure
    # function of the shared batch (slot_layout), so no extra communication
    offs = [hd.take_windows(batch, *ranges[r]).slot_layout() for r in range(world)]
    cons = hd.reassemble(ab.numpy(), al.numpy(), offs, ranges)
    if rank == 0:
        ret["cons"] = cons
        ret["ranges"] = ranges
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    import oracle
    n = 600
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
    batch = sim.window_batch(n, seed=5)
    want, st, _, _ = oracle.Oracle().poa_batch(batch)
    assert list(ret["cons"]) == want
    (b0, e0), (b1, e1) = ret["ranges"]
    assert b0 == 0 and e0 == b1 and e1 == n and 0 < e0 < n


def test_sharding_is_cost_balanced_and_contiguous():
    batch = sim.window_batch(5000, seed=9)
    costs = hd.window_costs(batch.windows, batch.arm_len)
    for world in (2, 4, 8):
        rs = hd.shard_contiguous(costs, world)
        assert rs[0][0] == 0 and rs[-1][1] == 5000
        assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        per = np.array([costs[b:e].sum() for b, e in rs])
        assert per.max() / per.mean() < 1.05
    sub = hd.take_windows(batch, 100, 200)
    assert sub.n_windows == 100 and int(sub.windows["first_arm"][0]) == 0
