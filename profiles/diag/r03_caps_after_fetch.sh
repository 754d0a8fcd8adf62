# caps of the three concurrent class kernels after Poa::fetch_next / the one-pass staging (the kernels' balance changed)
L=hypo_amd/_build/libhypo_gpu.so
for caps in 4,4,5 5,4,5 5,3,5 4,3,6 5,3,6 6,3,5 4,4,6 6,4,5 5,4,6 3,4,6 4,5,5; do
  for s in 0.002 0.005; do
    echo -n "caps $caps  "; HYPO_POA_CAPS=$caps HYPO_AB_CHILD=1 python profiles/ab_rate.py $L $s 2>&1 | grep libhypo | cut -c1-150
  done
done
