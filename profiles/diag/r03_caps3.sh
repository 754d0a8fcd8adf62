P="python profiles/diag/err_point.py"
L=hypo_amd/_build/libhypo_gpu.so; A=hypo_amd/_build/libhypo_gpu_alt832.so
for rep in 1 2; do
$P 0.002 20 $L 2>&1 | grep -v amdgpu
$P 0.002 20 $A 2>&1 | grep -v amdgpu
HYPO_POA_CAPS=3,4,5 $P 0.002 20 $A 2>&1 | grep -v amdgpu
HYPO_POA_CAPS=4,3,5 $P 0.002 20 $A 2>&1 | grep -v amdgpu
HYPO_POA_CAPS=4,4,4 $P 0.002 20 $A 2>&1 | grep -v amdgpu
done
HYPO_POA_CAPS=3,4,5 $P 0.01 10 $A 2>&1 | grep -v amdgpu
$P 0.01 10 $A 2>&1 | grep -v amdgpu
