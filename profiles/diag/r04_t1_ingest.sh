#!/bin/bash
# Where the 1 Gbp run (row T1, BAM) spends its time at -t 64 / -t 128: phase sums + the reader's own split (waiting / parse / collect).
# env: K (15), SZ (1g), P (10), THREADS ("64 128"), VARIANTS="X=1|Y=2" VAR=value settings to run one after the other on the same files; e.g. K=17 SZ=3g P=50 THREADS=64 r04_t1_ingest.sh 500 = the shape of the 3 Gbp row in small
set -e
N=${1:-1000}; K=${K:-15}; SZ=${SZ:-1g}; P=${P:-10}; THREADS=${THREADS:-"64 128"}
D=/dev/shm/t1ing_$$; rm -rf $D; mkdir -p $D
tests/_build/gen_e2e_fast $D 91 $N 1000000 $K 30 150 2000 --bam --fast-hash > /dev/null
cd $D
export GPU_MAX_HW_QUEUES=8
IFS="|" read -ra VARR <<< "${VARIANTS:-A=1}"
for V in "${VARR[@]}"; do
for T in $THREADS; do
  echo "-- $V"
  env $V HYPO_HOST_TIMING=1 $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s $SZ -c 30 -b sr.bam -t $T -i -p $P -o out.fa > run.log 2> run.err
  echo "== -t $T: $(grep Overall run.log | sed 's/.*TIME= //')  md5 $(md5sum out.fa | cut -c1-32)"
  grep "RESOURCES" run.log | python3 -c "
import sys,re,collections
acc=collections.OrderedDict()
for l in sys.stdin:
    m=re.search(r'\(\[Hypo:Hypo\]: (.*?)\. \): TIME= ([0-9.e+-]+)',l)
    if m: acc[m.group(1)]=acc.get(m.group(1),0)+float(m.group(2))
print('  '.join(f'{k}: {v:.2f}' for k,v in acc.items()))"
  grep "create_alignments_flat" run.err | awk '{w+=$6; p+=$10; c+=$15} END {printf "  reader: waiting for records %.2f s, parse %.2f s, into the batch %.2f s\n", w, p, c}'
  grep "upload_reads" run.err | awk '{f+=$8; u+=$17} END {printf "  upload_reads: flatten %.2f s, upload %.2f s\n", f, u}'
  grep "support_kmers:" run.err | sed 's/.*whole call //' | awk '{w+=$1} END {printf "  support_kmers: whole calls %.2f s\n", w}'
  grep "support_minimizers:" run.err | awk '{t+=$4; k+=$7} END {printf "  support_minimizers: tables %.2f s, device call %.2f s\n", t, k}'
  grep "device arms: flatten" run.err | awk '{f+=$5; b+=$8; p+=$13} END {printf "  short arms: tables %.2f s, hypo_gpu_arms_build %.2f s, prune %.2f s\n", f, b, p}'
  grep "hypo_gpu_arms_poa: device" run.err | sed 's/.*kernels //' | awk '{k+=$1} END {printf "  hypo_gpu_arms_poa: kernels %.2f s\n", k / 1000}'
done
done
rm -rf $D
