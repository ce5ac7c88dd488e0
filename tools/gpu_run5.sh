#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
for v in "" NOCONSUME NOREDUCE; do
  if [ -n "$v" ]; then export VOX_LIB=$REPO/voxtral-mini-realtime-rs_amd/abl_$v.so; else unset VOX_LIB; fi
  echo "== variant ${v:-default}"
  timeout 300 python tools/warm_cold.py 2>&1 | grep -v "amdgpu\|bench\]"
done | tee gpurun_out/ablation.log
unset VOX_LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $REPO/gpurun_out/pmc_sq1 -o sq1 -- python $REPO/tools/gemv_traffic.py > $REPO/gpurun_out/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmc_sq2 -o sq2 -- python $REPO/tools/gemv_traffic.py > $REPO/gpurun_out/pmc_sq2.log 2>&1
cd $REPO
tail -3 gpurun_out/pmc_sq1.log gpurun_out/pmc_sq2.log
python - <<'PY'
import csv, glob, collections
for tag in ("sq1", "sq2"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection.csv")
    if not fs: print(tag, "no counter csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        k = (row.get("Kernel_Name", "")[10:48], row.get("Counter_Name"))
        agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
    for k, (n, v) in sorted(agg.items()):
        if "gemv" in k[0] and ("2, 2, 2, 2" in k[0] or "2, 2, 1, 4" in k[0]): print(tag, k, "launches", n, "avg", round(v / n, 1))
PY
