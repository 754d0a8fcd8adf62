mkdir -p gpurun_out
export HYPO_HOST_TIMING=1
python profiles/e2e_time.py 32 3 > gpurun_out/e2e_time.txt 2>&1
grep "timing\|Short arms\|POA of\|Overall" gpurun_out/e2e_time.txt
# kernel trace of one run
python - <<'PY'
import sys, os, shlex
sys.path.insert(0, "tests")
import e2e_util as eu
os.makedirs("/tmp/e2e5m", exist_ok=True)
man = eu.make_inputs("e2e_5m_s11", "/tmp/e2e5m")
open("/tmp/e2e5m/cmd.txt", "w").write(man["command"])
PY
cd /tmp/e2e5m && export TMPDIR=/tmp
ARGS=$(cut -d' ' -f2- cmd.txt | sed 's/-t 1 /-t 32 /')
rocprofv3 --kernel-trace --stats -d /tmp/e2eprof -o e2e -- /root/repo/hypo_amd/_build/hypo $ARGS > /tmp/e2eprof.log 2>&1
cd /root/repo
find /tmp/e2eprof -type f | head; tail -5 /tmp/e2eprof.log
F=$(find /tmp/e2eprof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/n2_e2e_kernel_stats.csv && head -30 gpurun_out/n2_e2e_kernel_stats.csv | cut -c1-160
