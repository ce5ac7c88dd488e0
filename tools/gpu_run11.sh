#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
python tools/skinny_pmc.py 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $REPO/gpurun_out/pmc_sk1 -o sk1 -- python $REPO/tools/skinny_pmc.py > $REPO/gpurun_out/pmc_sk1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sk2 -o sk2 -- python $REPO/tools/skinny_pmc.py > $REPO/gpurun_out/pmc_sk2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_sk3 -o sk3 -- python $REPO/tools/skinny_pmc.py > $REPO/gpurun_out/pmc_sk3.log 2>&1
cd $REPO
for f in gpurun_out/pmc_sk1.log gpurun_out/pmc_sk2.log gpurun_out/pmc_sk3.log; do grep -i "error\|invalid\|not" $f | head -3; done
python - <<'PY'
import csv, glob, collections
for tag in ("sk1", "sk2", "sk3"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection.csv")
    if not fs: print(tag, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        if "skinny" not in row["Kernel_Name"]: continue
        k = (row["Kernel_Name"][10:40], row["Grid_Size"], row["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for k, (n, v) in sorted(agg.items()): print(tag, k, n, round(v / n, 1))
PY
