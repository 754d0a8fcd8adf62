#!/usr/bin/env python3
"""A/B of the bench's step (scan of the 5 Mbp contig + POA of the C2 batch, two batches alternating) between two libraries.
usage: r03_scan_ab.py <libA.so> <libB.so>"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(lib):
    import torch
    from hypo_amd import capi, sim
    gpu = capi.HypoGpu(0, path=lib)
    codes, p4 = sim.random_contig(5_000_000, seed=2000, n_frac=0.0)
    bits = sim.solid_bitset(codes, 11)
    ds = gpu.device_scan(p4, 5_000_000, 11, bits, kids_cap=2_500_000)
    dbs = [gpu.device_batch(sim.window_batch(97078, seed=s)) for s in (1000, 5000)]
    for i in range(6):
        ds.run(); dbs[i % 2].run()
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(40):
            ds.run(); dbs[i % 2].run()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 40 * 1e3)
    t0 = time.perf_counter()
    for i in range(200):
        ds.run()
    torch.cuda.synchronize()
    sc = (time.perf_counter() - t0) / 200 * 1e6
    print(f"{os.path.basename(lib)}: step ms {[round(x, 3) for x in out]}; scan alone back to back {sc:.1f} us per call", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for rep in range(3):
            for lib in sys.argv[1:3]:
                subprocess.run([sys.executable, __file__, "--child", lib], stderr=subprocess.DEVNULL)
