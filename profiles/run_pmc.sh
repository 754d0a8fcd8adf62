#!/bin/bash
# PMC passes for the bench's kernels (run on the GPU box: gpurun -- bash profiles/run_pmc.sh <tag>).
# Each counter set is collected in its own rocprofv3 run with --kernel-trace only (never with sys/hip traces).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-e2e --no-extras"}      # PMC_CMD: another workload (e.g. "python profiles/long_rate.py 200")
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq1 -o sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -o tcc -- $CMD > $OUT/tcc.log 2>&1
find $OUT -name "*.csv" | head -30
