B=hypo_amd/_build
export HYPO_GPU_LIB=$B/libhypo_gpu.so
for i in 1 2 3; do python profiles/poa_rate.py $HYPO_GPU_LIB 97078 20 2>&1 | tail -1; done
python profiles/hifi_rate.py 2>&1 | tail -1; python profiles/dense_rate.py 2>&1 | grep -i "dense-SR"
python profiles/err_rate.py $HYPO_GPU_LIB 2>&1 | grep read_sub
python -m pytest tests/test_gpu_poa.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
