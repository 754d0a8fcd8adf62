# how many new nodes the plan expects per differing arm byte (HYPO_PLAN_GROW_Q quarters): windows it sends to class 3 at once start there at t = 0 instead of arriving re-queued
B=hypo_amd/_build
run() { echo -n "$1  "; HYPO_AB_CHILD=1 python profiles/ab_rate.py $B/$1 $2 2>&1 | grep libhypo | cut -c26-160; }
for s in 0.002 0.005 0.01 0.02; do for l in libhypo_gpu.so libhypo_gpu_g3.so libhypo_gpu_g4.so; do run $l $s; done; done
