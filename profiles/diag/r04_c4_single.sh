#!/bin/bash
# Config C4 at size on the GPU box: ONE contig of <pieces> Mbp (gen_e2e_fast --join), 30x short reads with coverage gaps + 40x noisy long
# reads (-B), k from -s; phase sums, peak RSS, FASTA md5.  usage: r04_c4_single.sh [pieces=250] [size_flag=250m] [k=15]
set -e
N=${1:-250}; SZ=${2:-250m}; K=${3:-15}
D=/dev/shm/c4single_$$; rm -rf $D; mkdir -p $D
/usr/bin/env time -v true 2>/dev/null || true
S=$(date +%s.%N)
tests/_build/gen_e2e_fast $D 77 $N 1000000 $K 30 150 2000 --bam --fast-hash --join --long 40 8000 --gaps 100000 1500
echo "generated in $(awk -v a=$S -v b=$(date +%s.%N) "BEGIN{print b-a}") s"; ls -la $D
cd $D
export GPU_MAX_HW_QUEUES=8
S=$(date +%s.%N)
HYPO_HOST_TIMING=1 HYPO_DUMP_LONG=${DUMP_LONG:-} $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s $SZ -c 30 -b sr.bam -B lr.bam -t 64 -i -o out.fa > run.log 2> run.err || { tail -20 run.log run.err; exit 1; }
echo "process wall $(awk -v a=$S -v b=$(date +%s.%N) "BEGIN{print b-a}") s"
md5sum out.fa
grep "RESOURCES" run.log | sed 's/RESOURCES (\[Hypo:Hypo\]: //; s/\. ): TIME=/:/; s/sec.*PEAK RSS (so far)=/s, RSS/'
grep "Info:" run.log | head -30
grep timing run.err | head -40
# the BAM reader alone over the long reads (inflate + record cutting, nothing parsed)
g++ -O2 -fopenmp -std=c++17 -o /tmp/bgzf_bench_$$ $GRAFT_REPO_ROOT/profiles/diag/r04_bgzf_bench.cpp -lz -ldl && for t in 32 64; do echo "BAM reader alone, lr.bam, $t threads:"; OMP_WAIT_POLICY=passive /tmp/bgzf_bench_$$ lr.bam $t; done
cd /; rm -rf $D /tmp/bgzf_bench_$$
