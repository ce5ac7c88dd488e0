#!/bin/bash
# builds tools/micro/engine_bench_<tag> from the working-tree vox_engine.hip with extra compiler flags (code-layout experiments):
#   bash tools/micro/build_variants.sh al64 "-falign-loops=64" al256 "-falign-loops=256" ...
cd "$(dirname "$0")"; B=../../voxtral-mini-realtime-rs_amd/build; C=../../voxtral-mini-realtime-rs_amd/csrc
while [ $# -ge 2 ]; do tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -x hip -c $C/vox_engine.hip -o $B/_var_$tag.o 2>&1 | grep -E "error|unknown|unsupported" | head -3
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -o engine_bench_$tag engine_bench.o $B/vox_kernels.o $B/_var_$tag.o 2>&1 | grep -v "argument unused"; echo "built engine_bench_$tag ($flags)"
done
