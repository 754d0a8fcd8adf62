#!/bin/bash
# caps_fit_sweep.sh — caps whose LDS footprints fit one CU together (7.6 / 15.8 / 14.1 KB per wave of classes 0 / 1 / 2, 160 KB):
# the split between the three concurrent kernels then does not depend on which of them the dispatcher serves first.  Both
# warm-up stream modes (= two different stream -> hardware queue mappings) per setting.
for caps in 5,5,5 4,4,4 5,4,4 3,5,4 3,4,5 5,3,5 4,3,5 2,4,5 4,5,3 6,4,3 3,3,6 6,3,4 4,4,5; do
  for w in 1 0; do
    echo -n "caps=$caps warmup_stream=$w: "; HYPO_WARMUP_STREAM=$w HYPO_POA_CAPS=$caps HYPO_AB_CHILD=1 python profiles/ab_rate.py hypo_amd/_build/libhypo_gpu.so ${1:-0.002} 2>&1 | grep -v amdgpu | cut -c48-150
  done
done
