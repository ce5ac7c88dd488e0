#!/bin/bash
# round-1 evidence run: GPU suite, smoke, bench line (with cpu_baseline + batch extra), rocprofv3 kernel stats of the bench command,
# PMC HBM traffic of the dominant GEMV (separate FETCH_SIZE / WRITE_SIZE passes)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -o fetch -- python $REPO/tools/gemv_traffic.py > $REPO/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/pmc_write -o write -- python $REPO/tools/gemv_traffic.py > $REPO/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_bench -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch 0 > $REPO/gpurun_out/bench_prof.log 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/pmc_gemv_traffic.txt
import csv, glob, collections, json
res = {}
for tag in ("fetch", "write"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection.csv")
    if not fs: print(tag, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        k = (row.get("Kernel_Name", ""), row.get("Counter_Name"))
        agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
    for k, (n, v) in sorted(agg.items()):
        if "gemv" in k[0]:
            name = k[0].replace("void vox::", "").split("(")[0]
            print(tag, name, k[1], "launches", n, "avg", round(v / n, 1))
            res.setdefault(name, {})[k[1]] = v / n
out = {}
for name, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # counter unit = KB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads
        out[name] = int(2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024)
json.dump({"source": "profiles/r01_pmc_gemv_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 FETCH correction)",
           "hbm_bytes_per_launch": out}, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_batch -o b16 -- env VOX_BATCH_NO_GRAPH=1 python $REPO/tools/batch_prof.py 16 > $REPO/gpurun_out/prof_batch.log 2>&1
cd $REPO
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -2 gpurun_out/bench_n1.err; cat gpurun_out/bench_n1.json
