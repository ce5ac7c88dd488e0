#!/bin/bash
# Cross-compiles the micro-benchmarks for gfx950 (no GPU needed); the binaries travel to the GPU box with the gpurun snapshot.
cd "$(dirname "$0")"
for f in chain_floor gridbar_bench overlap_chain atomic_reduce; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $f $f.hip 2>&1 | grep -v "argument unused" ; echo "built $f"
done
