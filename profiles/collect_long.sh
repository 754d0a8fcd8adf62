#!/bin/bash
# collect_long.sh <tag> — the LONG-window part of collect_round.sh alone (rates, phase table, counters, C4 mix, registers, bench line)
TAG=${1:-r02_long}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
make -s -C hypo_amd/csrc prof >/dev/null 2>&1
(python profiles/long_rate.py 12; python profiles/long_rate.py 400; HYPO_GPU_LIB=hypo_amd/_build/libhypo_gpu_prof.so python profiles/long_rate.py 200) 2>&1 | grep -v "amdgpu.ids\|^CPU oracle" > $OUT/long_rate.txt
python profiles/c4_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/c4_rate.txt
PMC_CMD="python $R/profiles/long_rate.py 200" bash profiles/run_pmc.sh ${TAG}_long > /dev/null 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_${TAG}_long | grep "^kernel\|131072" > $OUT/pmc_long.csv
bash profiles/kernel_registers.sh > $OUT/kernel_registers.csv 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
