set -e
D=/tmp/c3run; rm -rf $D; mkdir -p $D
tests/_build/gen_e2e_fast $D 31 100 1000000 13 > /dev/null
cd $D
H=$GRAFT_REPO_ROOT/hypo_amd/_build/hypo
run() { "$@" | grep Overall | sed "s/RESOURCES (\[Hypo:Hypo\]: Overall. ): //"; }
for t in 16 32 64 128; do echo -n "passive t=$t p=10 "; run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t $t -i -p 10; done
for t in 32 64; do echo -n "passive t=$t p=100 "; run $H -d draft.fa -r reads.fa -s 100m -c 30 -b sr.sam -t $t -i -p 100; done
md5sum hypo_draft.fasta
D2=/tmp/c2run; rm -rf $D2; mkdir -p $D2; cd $GRAFT_REPO_ROOT; python - <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("gen_e2e", "tests/golden/gen_e2e.py"); g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
g.generate("/tmp/c2run", 11, 5000000, False, 11)
PY
cd $D2; for t in 16 32 64; do echo -n "C2 passive t=$t "; run $H -d draft.fa -r reads.fa -s 5m -c 30 -b sr.sam -t $t -i; done; md5sum hypo_draft.fasta
