set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i /dev/shm/c5
./tests/_build/gen_e2e_fast /dev/shm/c5 207 250 1000000 17 50 15000 1000 --bam --fast-hash > gpurun_out/r05i/gen.json
cd /dev/shm/c5
HYPO_HOST_TIMING=1 HYPO_REQUIRE_DEVICE=1 $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 3g -c 50 -b sr.bam -t 64 -i -p 50 > $GRAFT_REPO_ROOT/gpurun_out/r05i/c5_run.log 2> $GRAFT_REPO_ROOT/gpurun_out/r05i/c5_run.err
md5sum hypo_draft.fasta > $GRAFT_REPO_ROOT/gpurun_out/r05i/c5_md5.txt
cd /tmp && export TMPDIR=/tmp
cd /dev/shm/c5 && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05i/trace -o trace -- $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 3g -c 50 -b sr.bam -t 64 -i -p 50 -o prof.fa > /dev/null 2>&1 || true
rm -rf /dev/shm/c5
