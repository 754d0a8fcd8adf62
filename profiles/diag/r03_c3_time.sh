set -e
D=/tmp/c3run; rm -rf $D; mkdir -p $D
tests/_build/gen_e2e_fast $D 31 100 1000000 13 > /dev/null
cd $D
export GPU_MAX_HW_QUEUES=8
HYPO_HOST_TIMING=1 $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p 10 > run.log 2> run.err
grep "Overall" run.log; grep "timing" run.err | sed -n 1,12p; grep "RESOURCES" run.log | sed -n 12,24p | cut -c1-110
for t in 32 128; do $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t $t -i -p 10 | grep Overall | sed "s/^/t=$t /"; done
$GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p 25 | grep Overall | sed "s/^/p=25 /"
