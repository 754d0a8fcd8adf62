#!/bin/bash
# The host pipeline's DEVICE path (flat reads, staging, region tables, resident batches) under AddressSanitizer + UBSan on the GPU box:
# the hypo binary rebuilt with -fsanitize=address,undefined against the real libhypo_gpu.so, run on the C4-in-small set (SAM and BAM,
# -n 7 and default) and on a multi-batch BAM set.  The FASTA must keep its md5 and the sanitizers must stay silent.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
B=/tmp/asan_$$; mkdir -p $B
cd $R/hypo_amd/csrc
g++ -std=c++17 -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -o $B/hypo_asan host/main.cpp host/Window.cpp host/Contig.cpp host/Alignment.cpp host/Hypo.cpp \
    host/DeviceArms.cpp host/ReadBatch.cpp host/host_capi.cpp -L$R/hypo_amd/_build -lhypo_gpu -lz -ldl -Wl,-rpath,$R/hypo_amd/_build
export HYPO_ALLOW_DUP_DEVICES=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1 HYPO_NO_REEXEC=1 OMP_WAIT_POLICY=passive
run() {  # name, expected md5 or -, generator args -- hypo args
  local name=$1 want=$2; shift 2
  local D=/dev/shm/asan_$$_$name; rm -rf $D; mkdir -p $D
  local gen=(); while [ "$1" != "--" ]; do gen+=("$1"); shift; done; shift
  $R/tests/_build/gen_e2e_fast $D "${gen[@]}" > /dev/null
  (cd $D && $B/hypo_asan "$@" -o out.fa > run.log 2> run.err) || { echo "$name: FAILED"; tail -20 $D/run.err; return 1; }
  local got=$(md5sum $D/out.fa | cut -c1-32)
  local n=$(grep -c "ERROR: AddressSanitizer\|runtime error" $D/run.err || true)
  echo "$name: md5 $got (expected $want), sanitizer reports: $n, device paths: $(grep -c 'on the device' $D/run.log)"
  grep -m3 "ERROR: AddressSanitizer\|runtime error" $D/run.err || true
  grep -o "[a-z -]* on the device\|[a-z -]* on the host" $D/run.log | sort | uniq -c | sed 's/^/      /'
  rm -rf $D
}
run c4s_sam_n7 ce099e4541083bb19aa895b68bb87bfe 55 5 1000000 11 30 150 2000 --join --long 40 8000 --gaps 100000 1500 -- -d draft.fa -r reads.fa -s 5m -c 30 -b sr.sam -B lr.sam -t 16 -i -n 7
run c4s_bam 97a72d1d3ac343f137e692fae6911ead 55 5 1000000 11 30 150 2000 --join --long 40 8000 --gaps 100000 1500 --bam -- -d draft.fa -r reads.fa -s 5m -c 30 -b sr.bam -B lr.bam -t 16 -i
run c4s_bam_3ctx 97a72d1d3ac343f137e692fae6911ead 55 5 1000000 11 30 150 2000 --join --long 40 8000 --gaps 100000 1500 --bam -- -d draft.fa -r reads.fa -s 5m -c 30 -b sr.bam -B lr.bam -t 16 -i --devices 0,0,0
run batches_bam - 31 12 500000 13 30 150 2000 --bam --fast-hash -- -d draft.fa -r reads.fa -s 100m -c 30 -b sr.bam -t 16 -i -p 5
rm -rf $B
