#!/bin/bash
# GPU-box runner for the decode-engine micro harness:  gpurun --timeout T -- 'bash tools/engine_session.sh TAG "[bin:]args1" "args2" ...'
# every run is wrapped in its own timeout; output under gpurun_out/<TAG>_engine_<i>.txt.  An argument "ag1:26 100 ..." runs tools/micro/engine_bench_ag1.
OUT=gpurun_out; mkdir -p $OUT; TAG=$1; shift; i=0
for a in "$@"; do i=$((i+1)); bin=engine_bench; case "$a" in *:*) bin=engine_bench_${a%%:*}; a=${a#*:};; esac
  echo "=== $bin $a"; timeout 180 tools/micro/$bin $a > $OUT/${TAG}_engine_$i.txt 2>&1; echo "rc=$?"; cat $OUT/${TAG}_engine_$i.txt; done
