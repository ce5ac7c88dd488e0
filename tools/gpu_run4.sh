#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --timeout 600 --durations=5 -k "not model_shapes_random or 38 or 3072" 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -14 gpurun_out/pytest_gpu.log
timeout 300 python tools/torch_arena_check.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/torch_arena.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench4.json 2> gpurun_out/bench4.err
tail -3 gpurun_out/bench4.err; cat gpurun_out/bench4.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -o fetch -- python $REPO/tools/gemv_traffic.py > $REPO/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/pmc_write -o write -- python $REPO/tools/gemv_traffic.py > $REPO/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof4 -o r01 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof4_bench.json 2> $REPO/gpurun_out/prof4.err
cd $REPO; ls gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof4 | head -30
python - <<'PY'
import csv, glob, collections
for tag in ("fetch", "write"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection.csv")
    if not fs: print(tag, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        k = (row.get("Kernel_Name", "")[:60], row.get("Counter_Name"))
        agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
    for k, (n, v) in sorted(agg.items()):
        if "gemv" in k[0]: print(tag, k, "launches", n, "avg", v / n)
PY
