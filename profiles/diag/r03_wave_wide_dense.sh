B=hypo_amd/_build
d() { echo -n "$1 caps=$2 class0=$3  "; HYPO_GPU_LIB=$B/$1 HYPO_POA_CAPS=$2 HYPO_POA_CLASS0=$3 python profiles/dense_rate.py 2>&1 | grep "windows/s" | head -1 | cut -c1-110; echo -n "     hifi "; HYPO_GPU_LIB=$B/$1 HYPO_POA_CAPS=$2 HYPO_POA_CLASS0=$3 python profiles/hifi_rate.py 2>&1 | grep "windows/s" | head -1 | cut -c1-100; }
d libhypo_gpu.so 7,4,5 16
d libhypo_gpu_v1.so 7,4,5 16
d libhypo_gpu_v1.so 7,6,5 16
d libhypo_gpu_v1.so 7,6,4 16
d libhypo_gpu_v1.so 8,5,4 16
d libhypo_gpu_v2.so 12,4,4 32
d libhypo_gpu_v2.so 10,6,4 32
d libhypo_gpu_v2.so 8,6,5 32
d libhypo_gpu_v2.so 14,3,3 32
