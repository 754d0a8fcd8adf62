"""The N > 1 code on the one GPU a test box has: `bench.py --gpus 2` launched exactly as the driver launches it (torch.distributed.run, one
process per rank) with both ranks on device 0 and a gloo group (HYPO_BENCH_SHARE_GPU / HYPO_BENCH_BACKEND: RCCL itself needs two devices) —
cost-balanced sharding of ONE window batch, every rank's scans + POA of its range through libhypo_gpu, the fixed-size all-gather of the
consensus, max-over-ranks timing — and the gathered, re-assembled consensus of the whole batch compared with the oracle's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_device_gathers_the_oracle_consensus():
    env = dict(os.environ, HYPO_BENCH_SHARE_GPU="1", HYPO_BENCH_BACKEND="gloo", HYPO_BENCH_C3_WINDOWS="60000", HYPO_BENCH_CHECK_GATHER="1",
               MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="16")
    port = 29600 + os.getpid() % 300
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-e2e", "--no-extras", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["windows_total"] == 60000
    assert "bit-exact vs oracle" in j["parity_gathered"] and j["parity"] and "MISMATCH" not in j["parity"]
    assert j["imbalance"]["planned_cost_max_over_mean"] < 1.05 and j["value"] > 0
