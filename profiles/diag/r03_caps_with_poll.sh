# caps of the first passes when the polling class-3 launch runs beside them (it holds 16.6 KB of LDS per wave)
B=hypo_amd/_build
run() { echo -n "caps $1  "; HYPO_POA_CAPS=$1 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/libhypo_gpu.so $2 2>&1 | grep libhypo | cut -c26-160; }
for s in 0.005 0.01 0.02; do for c in 5,5,6 4,5,5 5,4,5 4,4,6 5,5,5 4,5,6 4,4,5 3,5,6; do run $c $s; run $c $s; done; done
