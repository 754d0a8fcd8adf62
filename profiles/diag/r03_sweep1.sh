mkdir -p gpurun_out
O=gpurun_out/r03_c_sweep.txt; : > $O
P="python profiles/diag/err_point.py"
for s in 0.002 0.005 0.01 0.05; do $P $s >> $O 2>&1; done
for s in 0.002 0.01; do HYPO_POA_POLL=0 $P $s >> $O 2>&1; done
for c in 5,4,4 4,5,4 5,3,5 3,4,5 4,3,5 4,4,4; do for s in 0.002 0.01; do HYPO_POA_CAPS=$c $P $s >> $O 2>&1; done; done
for w in 64 256 512; do for s in 0.005 0.01; do HYPO_POA_POLL_WAVES=$w $P $s >> $O 2>&1; done; done
grep -v amdgpu.ids $O
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o trace -- python $GRAFT_REPO_ROOT/profiles/diag/err_point.py 0.002 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python profiles/summarize_trace.py /tmp/prof_c2/*/trace_kernel_trace.csv > gpurun_out/r03_c_launches.csv 2>&1 || python profiles/summarize_trace.py $(find /tmp/prof_c2 -name "*kernel_trace.csv" | head -1) > gpurun_out/r03_c_launches.csv 2>&1; head -40 gpurun_out/r03_c_launches.csv
