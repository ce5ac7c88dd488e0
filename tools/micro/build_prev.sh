#!/bin/bash
# A/B helper: builds tools/micro/engine_bench_prev from the COMMITTED (git HEAD) vox_engine.hip next to the working-tree build, so both run in one gpurun call
# on the same box:  gpurun -- 'bash tools/engine_session.sh TAG "26 100 40 -1 640 50" "prev:26 100 40 -1 640 50"'
cd "$(dirname "$0")"; B=../../voxtral-mini-realtime-rs_amd/build; C=../../voxtral-mini-realtime-rs_amd/csrc
git show HEAD:voxtral-mini-realtime-rs_amd/csrc/vox_engine.hip > $C/_prev_engine.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $C/_prev_engine.hip -o $B/_prev_engine.o 2>&1 | grep -E "error" ; rm -f $C/_prev_engine.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o engine_bench_prev engine_bench.o $B/vox_kernels.o $B/_prev_engine.o 2>&1 | grep -v "argument unused"; echo "built engine_bench_prev"
