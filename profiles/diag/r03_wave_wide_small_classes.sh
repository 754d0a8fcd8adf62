# classes 0 / 1 with ONE window per wave (64 lanes x 2 columns) instead of two 32-lane groups:
#   hipcc ... -DHYPO_C1_GW=64 -DHYPO_C1_CPL=2 [-DHYPO_C0W_GW=64] -o hypo_amd/_build/libhypo_gpu_v1.so | _v2.so
B=hypo_amd/_build
run() { echo -n "$1 caps $2  "; HYPO_POA_CAPS=$2 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/$1 $3 2>&1 | grep libhypo | cut -c26-160; }
run libhypo_gpu.so 4,4,5 0.002
for c in 4,8,5 4,6,5 4,7,5 3,8,5 4,8,4 4,6,6; do run libhypo_gpu_v1.so $c 0.002; done
for c in 8,8,5 6,6,5 8,6,5 6,8,4 5,6,6 8,8,4 6,6,6 10,6,5; do run libhypo_gpu_v2.so $c 0.002; done
run libhypo_gpu.so 4,4,5 0.01
run libhypo_gpu_v1.so 4,8,5 0.01
run libhypo_gpu_v1.so 4,6,5 0.01
run libhypo_gpu_v2.so 8,8,5 0.01
run libhypo_gpu_v2.so 6,6,5 0.01
