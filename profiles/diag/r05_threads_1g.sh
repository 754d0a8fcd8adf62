# round 5: host threads of the 1 Gbp run (reader-bound): -t 64 (the bench's choice) against 96 / 128 on the 2 x 64-core box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out /dev/shm/t1g
./tests/_build/gen_e2e_fast /dev/shm/t1g 91 1000 1000000 15 30 150 2000 --bam --fast-hash > /dev/null
cd /dev/shm/t1g
for t in 64 96 128 64 96 128; do
  HYPO_REQUIRE_DEVICE=1 $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 1g -c 30 -b sr.bam -t $t -i -p 50 -o out_$t.fa > run_$t.log 2>&1
  echo "-t $t: $(grep 'Overall' run_$t.log | sed 's/.*TIME= //') ; md5 $(md5sum out_$t.fa | cut -c1-8)"
done
rm -rf /dev/shm/t1g
