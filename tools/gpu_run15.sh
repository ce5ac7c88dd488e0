#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
python tools/gemm_pmc.py 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_g1 -o g1 -- python $REPO/tools/gemm_pmc.py > $REPO/gpurun_out/pmc_g1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $REPO/gpurun_out/pmc_g2 -o g2 -- python $REPO/tools/gemm_pmc.py > $REPO/gpurun_out/pmc_g2.log 2>&1
cd $REPO
for f in gpurun_out/pmc_g1.log gpurun_out/pmc_g2.log; do grep -i "error\|invalid" $f | head -3; done
python - <<'PY' | tee gpurun_out/pmc_gemm_big.txt
import csv, glob, collections
for tag in ("g1", "g2"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection.csv")
    if not fs: print(tag, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        if "gemm_big" not in row["Kernel_Name"]: continue
        k = (row["Grid_Size"], row["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for k, (n, v) in sorted(agg.items()): print(tag, k, n, round(v / n, 1))
PY
