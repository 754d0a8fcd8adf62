mkdir -p gpurun_out
O=gpurun_out/r03_e_sweep.txt; : > $O
P="python profiles/diag/err_point.py"
for s in 0.002 0.005 0.01 0.02 0.05; do $P $s >> $O 2>&1; done
for s in 0.005 0.01; do $P $s >> $O 2>&1; done
for c in 4,4,4 5,4,4; do for s in 0.01 0.05; do HYPO_POA_CAPS=$c $P $s >> $O 2>&1; done; done
for w in 512 1024 2048; do for s in 0.02 0.05; do HYPO_POA_POLL_WAVES=$w $P $s >> $O 2>&1; done; done
grep -v amdgpu.ids $O
for f in dense_rate hifi_rate wide_rate; do timeout 300 python profiles/$f.py 2>&1 | grep -v "amdgpu.ids\|^CPU" | tail -2; done
