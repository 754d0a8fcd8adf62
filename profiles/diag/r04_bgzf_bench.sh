#!/bin/bash
# How fast do the BGZF reader's threads inflate + cut a BAM on the GPU box's host?  (T1: "Loaded alignments" is 40 of 58 s.)
set -e
D=/dev/shm/bgzfb_$$; mkdir -p $D
g++ -O2 -fopenmp -std=c++17 -o /tmp/bgzf_bench_$$ profiles/diag/r04_bgzf_bench.cpp -lz -ldl
tests/_build/gen_e2e_fast $D 77 250 1000000 15 30 150 2000 --bam --fast-hash > /dev/null
ls -la $D/sr.bam
for t in 1 8 32 64 128; do echo "threads $t:"; OMP_WAIT_POLICY=passive /tmp/bgzf_bench_$$ $D/sr.bam $t; done
echo "active wait policy, 64:"; /tmp/bgzf_bench_$$ $D/sr.bam 64
cp $D/sr.bam /tmp/sr_disk.bam; echo "from the overlay disk, 64:"; /tmp/bgzf_bench_$$ /tmp/sr_disk.bam 64
rm -rf $D /tmp/sr_disk.bam
