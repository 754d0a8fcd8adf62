# round 5: would two ranks, each reading its half of the alignments, beat one process on ONE box + ONE GPU?  Two independent 500 Mbp sets
# (k = 15) polished one after the other by one process each (-t 64), then side by side (-t 64 each, -t 32 each), all on device 0.
cd $GRAFT_REPO_ROOT; mkdir -p /dev/shm/pa /dev/shm/pb
./tests/_build/gen_e2e_fast /dev/shm/pa 91 500 1000000 15 30 150 2000 --bam --fast-hash > /dev/null
./tests/_build/gen_e2e_fast /dev/shm/pb 92 500 1000000 15 30 150 2000 --bam --fast-hash > /dev/null
H=$GRAFT_REPO_ROOT/hypo_amd/_build/hypo
run() { (cd /dev/shm/$1 && HYPO_REQUIRE_DEVICE=1 $H -d draft.fa -r reads.fa -s 1g -c 30 -b sr.bam -t $2 -i -p 50 -o out.fa > run.log 2>&1); }
t0=$(date +%s.%N); run pa 64; t1=$(date +%s.%N); run pb 64; t2=$(date +%s.%N)
echo "one after the other, -t 64: $(echo "$t1 $t0 $t2" | awk '{printf "%.2f s + %.2f s", $1-$2, $3-$1}')  (Overall timers: $(grep Overall /dev/shm/pa/run.log | sed 's/.*TIME= //;s/ sec.*//') / $(grep Overall /dev/shm/pb/run.log | sed 's/.*TIME= //;s/ sec.*//'))"
for t in 64 32; do
  t0=$(date +%s.%N); run pa $t & run pb $t & wait; t1=$(date +%s.%N)
  echo "side by side, -t $t each: $(echo "$t1 $t0" | awk '{printf "%.2f s", $1-$2}') wall  (Overall timers: $(grep Overall /dev/shm/pa/run.log | sed 's/.*TIME= //;s/ sec.*//') / $(grep Overall /dev/shm/pb/run.log | sed 's/.*TIME= //;s/ sec.*//'))"
done
rm -rf /dev/shm/pa /dev/shm/pb
