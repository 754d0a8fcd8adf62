#!/bin/bash
# How much of the time at 1 % / 5 % read error is the concurrent schedule's sharing of the chip?  Per-kernel times alone
# (HYPO_POA_SEQUENTIAL=1) against the concurrent schedule, with and without the polling class-3 launch.
cd "$(dirname "$0")/../.."
L=hypo_amd/_build/libhypo_gpu.so
cat > /tmp/err1.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from hypo_amd import capi, sim
gpu = capi.HypoGpu(0, path=sys.argv[1])
for sub in [float(x) for x in sys.argv[2].split(",")]:
    db = gpu.device_batch(sim.window_batch(97078, seed=1000, read_sub=sub))
    db.run(); db.run(); torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); db.run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    gpu.profile_begin(4)
    for _ in range(4): db.run()
    torch.cuda.synchronize()
    prof = gpu.profile_read()
    st = db.stats()
    print(f"sub={sub} calls ms {[round(t,2) for t in ts]} classes={st['n_class'][:5]}")
    for p in prof: print("    kernels ms [plan, c0..c5, call]", [round(float(x), 2) for x in p])
PY
for cfg in "" "HYPO_POA_SEQUENTIAL=1" "HYPO_POA_POLL=0" "HYPO_POA_CAPS=3,3,4" "HYPO_POA_CAPS=5,5,5" "HYPO_POA_ORDER=210" "HYPO_POA_ORDER=012"; do
  echo "== $cfg"
  env $cfg python /tmp/err1.py $L 0.01,0.05 2>&1 | grep -v amdgpu
done
