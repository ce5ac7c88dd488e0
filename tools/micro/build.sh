#!/bin/bash
# Cross-compiles the micro-benchmarks for gfx950 (no GPU needed); the binaries travel to the GPU box with the gpurun snapshot.
cd "$(dirname "$0")"
for f in chain_floor gridbar_bench overlap_chain atomic_reduce edge_pingpong edge_fanin; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $f $f.hip 2>&1 | grep -v "argument unused" ; echo "built $f"
done
for f in mfma_fp8_dot mfma_i8_dot; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $f $f.hip 2>&1 | grep -v "argument unused"; echo "built $f"; done
# engine_bench links the product's kernel objects (python voxtral-mini-realtime-rs_amd/build.py first)
B=../../voxtral-mini-realtime-rs_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -o engine_bench.o engine_bench.hip 2>&1 | grep -v "argument unused"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o engine_bench engine_bench.o $B/vox_kernels.o $B/vox_engine.o 2>&1 | grep -v "argument unused"; echo "built engine_bench"
# engine_b16_bench: the batched decode-layer engine against a CPU restatement (OpenMP on the host side)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fopenmp -c -o engine_b16_bench.o engine_b16_bench.hip 2>&1 | grep -v "argument unused"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fopenmp -o engine_b16_bench engine_b16_bench.o $B/vox_kernels.o $B/vox_engine.o $B/vox_engine_b16.o 2>&1 | grep -v "argument unused"; echo "built engine_b16_bench"
# engine_b32_bench: the two-group launch + cache-slice indirection against the one-group launch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -o engine_b32_bench.o engine_b32_bench.hip 2>&1 | grep -v "argument unused"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o engine_b32_bench engine_b32_bench.o $B/vox_kernels.o $B/vox_engine.o $B/vox_engine_b16.o 2>&1 | grep -v "argument unused"; echo "built engine_b32_bench"
