#!/usr/bin/env python3
"""Where does a host-pointer POA call spend its time?  begin() / end() durations, pageable vs page-locked buffers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hypo_amd import capi, sim
from hypo_amd.batch import HostBatch
gpu = capi.HypoGpu(0)
batch = sim.window_batch(97078, seed=1000)
off = batch.slot_layout(); n = batch.n_windows
pin = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).pin_memory().numpy()
for tag, b, poff, mk in (("pageable", batch, off, lambda k, dt: np.zeros(k, dt)),
                         ("pinned", HostBatch(pin(batch.windows).view(batch.windows.dtype), pin(batch.draft4), batch.arm_off, pin(batch.arm_len).view(np.uint32), pin(batch.arms2)),
                          pin(off).view(np.uint64), lambda k, dt: torch.zeros(k * np.dtype(dt).itemsize, dtype=torch.uint8).pin_memory().numpy().view(dt))):
    outs = [(mk(int(off[-1]) + 16, np.uint8), mk(n, np.uint32), mk(n, np.uint8)) for _ in range(2)]
    for depth in (1, 2):
        for rep in range(2):
            pending, tb, te = [], [], []
            t0 = time.perf_counter()
            for i in range(8):
                if len(pending) == depth:
                    t = time.perf_counter(); gpu.poa_batch_end(pending.pop(0)[0]); te.append(time.perf_counter() - t)
                t = time.perf_counter(); pending.append(gpu.poa_batch_begin(b, poff, *outs[i % 2], no_arm_off=True)); tb.append(time.perf_counter() - t)
            while pending:
                t = time.perf_counter(); gpu.poa_batch_end(pending.pop(0)[0]); te.append(time.perf_counter() - t)
            dt = (time.perf_counter() - t0) / 8
        print(f"{tag:9s} depth {depth}: {dt * 1e3:.2f} ms/call  begin {np.mean(tb[2:]) * 1e3:.2f} ms  end {np.mean(te[2:]) * 1e3:.2f} ms", flush=True)
