#!/bin/bash
# long_grid.sh — grid, LDS and registers of the LONG-window kernel as launched for a 2 400-window batch (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python $R/profiles/long_rate.py ${1:-200} > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "639" in r["Kernel_Name"]:
        print("grid", r["Grid_Size_X"], "wg", r["Workgroup_Size_X"], "lds", r["LDS_Block_Size"], "vgpr", r.get("VGPR_Count"), "agpr", r.get("Accum_VGPR_Count"), "sgpr", r.get("SGPR_Count"),
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
PY
grep -v "^[EW]2026" /tmp/kt.log | tail -8; ls /tmp/kt
