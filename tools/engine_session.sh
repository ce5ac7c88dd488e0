#!/bin/bash
# GPU-box runner for the decode-engine micro harness:  gpurun --timeout T -- 'bash tools/engine_session.sh TAG "args1" "args2" ...'
# every run is wrapped in its own timeout; output under gpurun_out/<TAG>_engine_<i>.txt
OUT=gpurun_out; mkdir -p $OUT; TAG=$1; shift; i=0
for a in "$@"; do i=$((i+1)); echo "=== engine_bench $a"; timeout 180 tools/micro/engine_bench $a > $OUT/${TAG}_engine_$i.txt 2>&1; echo "rc=$?"; cat $OUT/${TAG}_engine_$i.txt; done
