#!/usr/bin/env python3
"""Row T1 end to end (north_star: 3 Gbp / 30x short reads on one MI355X): bench.py's e2e_t1 leg on its own.
usage: t1_run.py [contigs=3000] [batchings=10,50]   ->  one JSON line (what `bench.py --t1-contigs N` puts under "e2e_t1")."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
pbs = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (10, 50)
print(json.dumps(bench.end_to_end_t1_leg(n, batchings=pbs)))
