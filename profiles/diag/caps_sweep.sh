#!/bin/bash
# caps_sweep.sh [read_sub] — waves-per-CU caps of the three concurrent class kernels x submission order, two runs each
# (the balance between the kernels also depends on which hardware queues the streams land on: see DESIGN.md 3.1)
SUB=${1:-0.002}
for caps in 5,5,5 6,6,6 4,5,5 5,5,6 5,6,5 6,5,5 4,6,5 5,4,6 6,6,5 5,6,6 7,7,7; do
  for ord in 201 102 012; do
    for rep in 1 2; do
      echo -n "caps=$caps order=$ord: "; HYPO_POA_CAPS=$caps HYPO_POA_ORDER=$ord HYPO_AB_CHILD=1 python profiles/ab_rate.py hypo_amd/_build/libhypo_gpu.so $SUB 2>&1 | grep -v amdgpu | cut -c50-150
    done
  done
done
