#!/bin/bash
# Phase sums of the C3 end-to-end run (100 x 1 Mbp, -p 10) on the GPU box: where the wall time goes after N1 / N2 moved to the device.
set -e
D=/tmp/c3run; rm -rf $D; mkdir -p $D
tests/_build/gen_e2e_fast $D 31 100 1000000 13 > /dev/null
cd $D
export GPU_MAX_HW_QUEUES=8
for p in 10 100; do
HYPO_HOST_TIMING=1 $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p $p > run.log 2> run.err
echo "== -p $p"
grep "RESOURCES" run.log | python3 -c "
import sys,re,collections
acc=collections.OrderedDict()
for l in sys.stdin:
    m=re.search(r'\(\[Hypo:Hypo\]: (.*?)\. \): TIME= ([0-9.e+-]+)',l)
    if m: acc[m.group(1)]=acc.get(m.group(1),0)+float(m.group(2))
for k,v in acc.items(): print(f'{v:8.3f} s  {k}')
"
grep -c timing run.err || true
grep timing run.err | grep -v find_solid_pos | sed -n 1,24p | cut -c1-200
done
