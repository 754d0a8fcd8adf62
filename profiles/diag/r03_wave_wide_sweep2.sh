B=hypo_amd/_build
run() { echo -n "$1 caps $2  "; HYPO_POA_CAPS=$2 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/$1 $3 2>&1 | grep libhypo | cut -c26-150; }
for s in 0.002 0.005 0.01; do
for c in 4,6,6 4,5,6 3,6,6 5,6,6 4,7,6 4,5,7 4,6,7 3,5,7 5,5,6; do run libhypo_gpu_v1.so $c $s; done
for c in 6,6,6 5,6,6 7,6,6 6,5,6 8,6,6 6,5,7 6,7,6 8,5,6 6,4,7; do run libhypo_gpu_v2.so $c $s; done
done
