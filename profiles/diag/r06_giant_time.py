import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from hypo_amd import capi
from hypo_amd.batch import build_batch
from test_giant import giant_windows
rng = np.random.default_rng(606)
wins = giant_windows(rng, deep_arms=20000)
gpu = capi.HypoGpu(0, path=os.path.join("hypo_amd", "_build", "libhypo_gpu_prof.so"))
for i, w in enumerate(wins):
    b = build_batch([w])
    t = time.time(); r = gpu.poa_batch(b); dt = time.time() - t
    print("window", i, "status", r[3], f"{dt:.2f} s", flush=True)
