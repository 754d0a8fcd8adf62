set -e
D=/tmp/c3run; rm -rf $D; mkdir -p $D
tests/_build/gen_e2e_fast $D 31 100 1000000 13 > /dev/null
cd $D
export GPU_MAX_HW_QUEUES=8
H=$GRAFT_REPO_ROOT/hypo_amd/_build/hypo
run() { "$@" | grep Overall | sed "s/RESOURCES (\[Hypo:Hypo\]: Overall. ): //"; }
for t in 8 16 24 32 48; do echo -n "t=$t "; run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t $t -i -p 10; done
echo -n "t=64 passive "; OMP_WAIT_POLICY=passive run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p 10
echo -n "t=64 bind "; OMP_PROC_BIND=close OMP_PLACES=cores run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p 10
echo -n "t=64 arena4 "; MALLOC_ARENA_MAX=4 run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p 10
echo -n "t=32 p25 "; run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 32 -i -p 25
echo -n "t=32 p100 "; run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 32 -i -p 100
HYPO_HOST_TIMING=1 $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 32 -i -p 10 > run.log 2> run.err
grep "create_alignments\|device arms" run.err | sed -n 1,6p; grep "RESOURCES" run.log | sed -n 12,20p | cut -c1-100; lscpu | grep -i "socket\|numa\|model name" | head -6
