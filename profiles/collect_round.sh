#!/bin/bash
# collect_round.sh <tag> — everything the round's DESIGN.md quotes, produced on the GPU box in one go:
#   gpurun -- bash profiles/collect_round.sh r02
# writes gpurun_out/<tag>/ (scratch); copy what should be judged into profiles/ (tracked).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
make -s -C hypo_amd/csrc prof >/dev/null 2>&1   # the diagnostic build must match the sources (it is not built by the default target)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# kernel trace + counters of the bench's kernels (concurrent schedule = the product; sequential = every class alone)
bash profiles/run_pmc.sh $TAG > /dev/null 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_$TAG > $OUT/pmc_summary.csv
python profiles/summarize_trace.py gpurun_out/pmc_$TAG/trace/trace_kernel_trace.csv > $OUT/kernel_launches.csv
cp gpurun_out/pmc_$TAG/trace/trace_kernel_stats.csv $OUT/kernel_stats.csv
HYPO_POA_SEQUENTIAL=1 bash profiles/run_pmc.sh ${TAG}_seq > /dev/null 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_${TAG}_seq > $OUT/pmc_summary_sequential.csv
python profiles/summarize_trace.py gpurun_out/pmc_${TAG}_seq/trace/trace_kernel_trace.csv > $OUT/kernel_launches_sequential.csv
# per-phase cycle counters of the diagnostic build
for sub in 0.002 0.01; do
  python profiles/phase_profile.py 97078 $sub 2>&1 | grep -v "amdgpu.ids\|Warning\|print(" > $OUT/phase_profile_concurrent_$sub.txt
  HYPO_POA_SEQUENTIAL=1 python profiles/phase_profile.py 97078 $sub 2>&1 | grep -v "amdgpu.ids\|Warning\|print(" > $OUT/phase_profile_sequential_$sub.txt
done
# rates on other shapes
python profiles/err_rate.py hypo_amd/_build/libhypo_gpu.so 5 2>&1 | grep -v amdgpu.ids > $OUT/err_rate.txt
python profiles/dense_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/dense_rate.txt
python profiles/hifi_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/hifi_rate.txt
python profiles/wide_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/wide_rate.txt
(python profiles/long_rate.py 12; python profiles/long_rate.py 400; HYPO_GPU_LIB=hypo_amd/_build/libhypo_gpu_prof.so python profiles/long_rate.py 200) 2>&1 | grep -v "amdgpu.ids\|^CPU oracle" > $OUT/long_rate.txt
(python profiles/scan_rate.py 5000000 11; python profiles/scan_rate.py 100000000 13; python profiles/scan_rate.py 250000000 15; python profiles/scan_rate.py 512000000 17) 2>&1 | grep -v amdgpu.ids > $OUT/scan_rate.txt
python profiles/diag/host_api_timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/host_api_timeline.txt
python profiles/c4_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/c4_rate.txt
PMC_CMD="python $R/profiles/long_rate.py 200" bash profiles/run_pmc.sh ${TAG}_long > /dev/null 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_${TAG}_long | grep "^kernel\|131072" > $OUT/pmc_long.csv
bash profiles/kernel_registers.sh > $OUT/kernel_registers.csv 2>/dev/null
# the scan's counters at the sizes the rate table quotes (round 5: FETCH_SIZE / TCC hit + miss / SQ_WAIT_ANY of the 100 / 250 / 512 Mbp launches)
for spec in "5000000 11" "100000000 13" "250000000 15" "512000000 17"; do
  set -- $spec
  PMC_CMD="python $R/profiles/scan_rate.py $1 $2" bash profiles/run_pmc.sh ${TAG}_scan$2 > /dev/null 2>&1
  python profiles/summarize_pmc.py gpurun_out/pmc_${TAG}_scan$2 | grep "^kernel\|scan_fused" > $OUT/pmc_scan_k$2.csv
done
# round 6: the 8 % cells of the SURVEY 8(d) grid under several schedules (profiles/diag/r06_grid_cell.py)
(for c in "32 6" "64 6" "100 6" "16 50" "32 12"; do echo "== cell $c 0.08"; python profiles/diag/r06_grid_cell.py $c 0.08 2>&1 | grep -v amdgpu.ids; done) > $OUT/grid_cells.txt
ls -la $OUT
