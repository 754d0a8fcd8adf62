#!/bin/bash
# Which kernels does an end-to-end run spend its device time in?  rocprofv3 --kernel-trace --stats around the hypo binary on the C3 set
# (100 x 1 Mbp, SAM, k = 13, -p 10) and on a 250 x 1 Mbp BAM set at k = 17 (the T1 shape: a random genome marks 40 % of its positions).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
run() {   # name, generator args..., then hypo args after --
  local name=$1; shift
  local D=/dev/shm/e2ek_$$_$name; rm -rf $D; mkdir -p $D
  local gen=(); while [ "$1" != "--" ]; do gen+=("$1"); shift; done; shift
  $R/tests/_build/gen_e2e_fast $D "${gen[@]}" > /dev/null
  (cd $D && GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -o t -- $R/hypo_amd/_build/hypo "$@" -o out.fa > run.log 2> run.err) || { tail -5 $D/run.err; return 1; }
  echo "== $name: $(grep Overall $D/run.log | sed 's/.*TIME= //')"
  local f=$(find $D/prof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"   device time in kernels: {tot / 1e9:.3f} s over {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    n = r["Name"]
    n = n[:n.index("(")] if "(" in n else n
    n = n.replace("void hypo::", "").replace("hypo::", "")
    if len(n) > 90: n = n[:87] + "..."
    print(f"   {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:9.1f} us  {n}")
PY
  rm -rf $D
}
cd /tmp
run c3 31 100 1000000 13 -- -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p 10
run k17_250m 91 250 1000000 17 30 150 2000 --bam --fast-hash -- -d draft.fa -r reads.fa -s 3g -c 30 -b sr.bam -t 64 -i -p 50
