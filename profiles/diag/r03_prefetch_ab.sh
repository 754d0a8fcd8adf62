# A/B of Poa::fetch_next (HYPO_PREFETCH_NEXT): build the variant without it first:
#   cd hypo_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHYPO_PREFETCH_NEXT=0 -DHYPO_BUILD_ID=\"nopf\" -shared -o ../_build/libhypo_gpu_nopf.so *.hip
B=hypo_amd/_build
for l in $B/libhypo_gpu_nopf.so $B/libhypo_gpu.so; do
  for s in 0.002 0.005 0.01; do python profiles/ab_rate.py $l $s 2>&1 | grep libhypo; done
  HYPO_GPU_LIB=$l python profiles/dense_rate.py 2>&1 | grep windows | head -1
  HYPO_GPU_LIB=$l python profiles/hifi_rate.py 2>&1 | grep windows | head -1
done
