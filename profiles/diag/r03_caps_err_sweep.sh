# fixed wave shares of the three concurrent class kernels at 1 % and 2 % read error, each twice (the landscape the adaptive shares walk on)
B=hypo_amd/_build
run() { echo -n "caps $1  "; HYPO_POA_CAPS=$1 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/libhypo_gpu.so $2 2>&1 | grep libhypo | cut -c37-150; }
for s in 0.01 0.02; do for c in 5,5,6 4,6,6 4,6,5 3,6,6 4,5,6 3,7,5 4,7,5 3,6,5 4,5,5; do run $c $s; run $c $s; done; done
echo "== dense / HiFi: what the adaptive shares are"
HYPO_POA_ADAPT_LOG=1 python profiles/dense_rate.py 2>&1 | grep "dense-SR\|wave shares" | uniq -c
HYPO_POA_ADAPT_LOG=1 python profiles/hifi_rate.py 2>&1 | grep "HiFi\|wave shares" | uniq -c
