D=/tmp/c3run; rm -rf $D; mkdir -p $D
tests/_build/gen_e2e_fast $D 31 100 1000000 13 > /dev/null
cd $D
export GPU_MAX_HW_QUEUES=8
for p in 10 10; do
s=$(date +%s.%N)
HYPO_HOST_TIMING=1 $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p $p > run.log 2> run.err
e=$(date +%s.%N)
echo "== -p $p wall $(echo "$e - $s" | bc)"
grep Overall run.log; grep "main:" run.err
done
