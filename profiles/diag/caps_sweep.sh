for caps in 4,4,6 4,5,6 4,5,5 3,5,6 5,5,5 4,6,5 5,5,6 3,6,6 6,6,6 4,4,7 5,4,6; do
  for ord in 201 102; do
    echo -n "caps=$caps order=$ord: "; HYPO_POA_CAPS=$caps HYPO_POA_ORDER=$ord HYPO_AB_CHILD=1 python profiles/ab_rate.py hypo_amd/_build/libhypo_gpu.so 0.002 2>&1 | grep -v amdgpu | cut -c50-150
  done
done
