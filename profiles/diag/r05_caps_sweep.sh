# round 5: wave shares of the three concurrent LDS-class kernels after threading along the guide (class 1 is now the longest kernel of the C2 step)
B=hypo_amd/_build
run() { echo -n "caps $1 sub $2  "; HYPO_POA_CAPS=$1 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/libhypo_gpu.so $2 2>&1 | grep libhypo | cut -c26-150; }
for s in 0.002 0.01; do
for c in 5,5,6 5,6,5 4,6,6 5,6,6 4,7,5 5,7,5 6,6,5 4,6,5 6,6,4 5,7,4 6,7,4; do run $c $s; done
done
