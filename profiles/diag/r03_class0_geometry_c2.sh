# C2 mix: class 0 as four 16-lane groups per wave (the dense-batch geometry) against two 32-lane groups, after class 1 went wave-wide
B=hypo_amd/_build
run() { echo -n "class0=$1 caps $2  "; HYPO_POA_CLASS0=$1 HYPO_POA_CAPS=$2 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/libhypo_gpu.so $3 2>&1 | grep libhypo | cut -c26-160; }
for s in 0.002 0.005; do
run 32 5,5,6 $s
for c in 4,5,6 5,5,6 3,5,6 3,6,6 6,5,5 2,5,6 3,5,7; do run 16 $c $s; done
done
