cd /root/repo
for cfg in "HYPO_POA_SEQUENTIAL=1" "HYPO_POA_SERIAL=1 HYPO_POA_POLL=0"; do
  echo "== $cfg"
  env $cfg python profiles/err_rate.py hypo_amd/_build/libhypo_gpu.so 10 2>&1 | grep -v amdgpu | grep read_sub | cut -c1-75
done
