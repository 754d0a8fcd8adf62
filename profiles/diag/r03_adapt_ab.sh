# wave shares of the three concurrent class kernels: fixed {5,5,6} (HYPO_POA_ADAPT=0) against shares picked from the last call's wave-time
B=hypo_amd/_build
for a in 0 1; do
  echo "== HYPO_POA_ADAPT=$a"
  HYPO_POA_ADAPT=$a HYPO_POA_ADAPT_LOG=1 python profiles/err_rate.py $B/libhypo_gpu.so 8 2>&1 | grep -v "amdgpu.ids" | uniq -c | grep -v "^ *[12] \[hypo_gpu\]"
  HYPO_POA_ADAPT=$a python profiles/dense_rate.py 2>&1 | grep "dense-SR"
  HYPO_POA_ADAPT=$a python profiles/hifi_rate.py 2>&1 | grep "HiFi"
done
echo "== 5 % with the concurrent schedule forced, adaptive shares"
HYPO_POA_SEQUENTIAL=0 HYPO_POA_ADAPT=1 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/libhypo_gpu.so 0.05 2>&1 | grep libhypo | cut -c26-160
HYPO_POA_SEQUENTIAL=0 HYPO_POA_ADAPT=1 HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/libhypo_gpu.so 0.03 2>&1 | grep libhypo | cut -c26-160
