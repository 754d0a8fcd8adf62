D=/tmp/c3run; rm -rf $D; mkdir -p $D
tests/_build/gen_e2e_fast $D 31 100 1000000 13 > /dev/null
cd $D
for m in 1 2 0; do
echo "== HYPO_DIAG_PARSE=$m"
HYPO_DIAG_PARSE=$m HYPO_HOST_TIMING=1 $GRAFT_REPO_ROOT/hypo_amd/_build/hypo -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t 64 -i -p 100 2>&1 | grep -E "create_alignments|Loaded alignments"
done
