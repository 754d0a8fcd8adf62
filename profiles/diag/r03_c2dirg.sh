#!/bin/bash
# Class 2 with its direction codes in HBM scratch (-DHYPO_C2_DIRG=1: 7.4 KB of LDS per window instead of 14.1) against the product.
cd "$(dirname "$0")/../.."
A=hypo_amd/_build/libhypo_gpu.so; B=hypo_amd/_build/libhypo_gpu_c2dirg.so
for sub in 0.002 0.01; do
python profiles/ab_rate.py $A $sub 2>&1 | grep -v amdgpu
python profiles/ab_rate.py $B $sub 2>&1 | grep -v amdgpu
for caps in 4,4,7 4,4,9 4,3,10 3,3,10; do
  HYPO_POA_CAPS=$caps python profiles/ab_rate.py $B $sub 2>&1 | grep -v amdgpu | grep concurrent | sed "s/^/caps=$caps /"
done
done
