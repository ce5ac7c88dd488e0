#!/bin/bash
# One parametrised GPU-box runner (replaces round 1's gpu_run1..19.sh):  gpurun --timeout T -- 'bash tools/gpu_session.sh step [step ...]'
# Every step is wrapped in its own `timeout`, logs under gpurun_out/ (merged back by gpurun); steps never abort the session.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${VOX_TAG:-r05}
step_tests()     { timeout 900 python -m pytest tests -m gpu -x -q ${VOX_PYTEST_ARGS:-} > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/${TAG}_pytest_gpu.log; }
step_tests_all() { timeout 900 python -m pytest tests -m gpu -q -rA ${VOX_PYTEST_ARGS:-} > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|full-size|batch" $OUT/${TAG}_pytest_gpu.log | tail -30; }
step_smoke()     { timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/${TAG}_smoke.log; }
step_f32golden() { timeout 900 python tests/golden/make_fullsize_f32_golden.py > $OUT/${TAG}_f32golden.log 2>&1; echo "f32golden rc=$?"; tail -4 $OUT/${TAG}_f32golden.log; }
step_micro()     { # tools/micro/build.sh first (cross-compiles here, the binaries travel with the snapshot)
                   timeout 300 tools/micro/chain_floor > $OUT/${TAG}_chain_floor.txt 2>&1; echo "micro rc=$?"; cat $OUT/${TAG}_chain_floor.txt
                   for b in overlap_chain atomic_reduce; do [ -x tools/micro/$b ] && timeout 120 tools/micro/$b | tee $OUT/${TAG}_$b.txt; done; }
step_timeline()  { timeout 300 python tools/timeline.py > $OUT/${TAG}_timeline.txt 2>&1; echo "timeline rc=$?"; cat $OUT/${TAG}_timeline.txt; }
step_bench()     { timeout 900 python bench.py --steps ${VOX_BENCH_STEPS:-10} --warmup 3 ${VOX_BENCH_ARGS:-} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err; }
step_benchq()    { timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch ${VOX_BENCH_BATCH:-16} ${VOX_BENCH_ARGS:---no-f32 --fleurs-clips 0} > $OUT/${TAG}_benchq.json 2> $OUT/${TAG}_benchq.err; echo "benchq rc=$?"; python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_benchq.json").read().strip().splitlines()[-1])
r = d["roofline"]; print("value", d["value"], "ms/step", d["ms_per_step"], "stage", d["stage_ms"], "decode step ms", r["decode_step_measured_ms"])
print({k: (v["avg_us"], v["GBps"]) for k, v in r["all_decode_gemvs"].items()}); print("batch", d.get("batch")); print("f32", d.get("f32")); print("fleurs", d.get("fleurs_like"))
PY
}
step_prof()      { cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --fleurs-clips 0 --batch ${VOX_BENCH_BATCH:-0} ${VOX_BENCH_ARGS:-} > $OUT/${TAG}_prof.log 2>&1; echo "prof rc=$?"; cd $REPO
                   f=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv && head -24 $f | cut -c1-160; }
step_profbatch() { cd /tmp; VOX_BATCH_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_profb -o p -- python $REPO/tools/batch_prof.py 16 > $OUT/${TAG}_profb.log 2>&1; echo "profbatch rc=$?"; cd $REPO
                   f=$(find $OUT/${TAG}_profb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_batch16_kernel_stats.csv && head -16 $f | cut -c1-160; }
step_prefill()   { timeout 300 python tools/prefill_bench.py 2>&1 | tee $OUT/${TAG}_prefill.txt | tail -6; }
step_pmc()       { # per-kernel PMC averages of the batched decode step (separate --pmc passes, kernel trace only: no other trace domains)
                   cd /tmp; i=0
                   for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
                     i=$((i+1)); VOX_BATCH_NO_GRAPH=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o p -- python $REPO/${VOX_PMC_CMD:-tools/batch_prof.py 16} > $OUT/${TAG}_pmc$i.log 2>&1; echo "pmc pass $i rc=$?"
                   done; cd $REPO; python tools/pmc_summary.py $OUT/${TAG}_pmc* > $OUT/${TAG}_pmc_summary.txt 2>&1; cat $OUT/${TAG}_pmc_summary.txt | head -60; }
step_traffic()   { # HBM bytes per launch of the dominant decode GEMV: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes over a short target
                   cd /tmp
                   timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python $REPO/tools/gemv_traffic.py > $OUT/${TAG}_pmc_fetch.log 2>&1; echo "fetch rc=$?"
                   timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python $REPO/tools/gemv_traffic.py > $OUT/${TAG}_pmc_write.log 2>&1; echo "write rc=$?"
                   cd $REPO; python tools/traffic_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write | tee $OUT/${TAG}_pmc_gemv_traffic.txt; }
step_ablate()    { # what each part of q4_gemv_kernel costs: product build, then every abl_* measurement build present
                   python tools/gemv_ablate.py 2>/dev/null | tail -1 | tee $OUT/${TAG}_gemv_ablate.txt
                   for l in voxtral-mini-realtime-rs_amd/libvoxtral_hip_abl_*.so; do VOX_LIB=$REPO/$l timeout 120 python tools/gemv_ablate.py 2>/dev/null | tail -1 | tee -a $OUT/${TAG}_gemv_ablate.txt; done; }
step_continuous() { timeout 600 python tools/continuous_sweep.py ${VOX_WORLD:-8} ${VOX_CLIPS:-647} > $OUT/${TAG}_continuous_sweep.log 2>&1; echo "continuous rc=$?"; grep -v "continuous batch:" $OUT/${TAG}_continuous_sweep.log | tail -12; }
step_batch()     { timeout 300 python tools/batch_prof.py ${VOX_BENCH_BATCH:-16} 2>&1 | tail -4; }
for s in "$@"; do echo "=== $s"; t0=$(date +%s); step_$s; echo "--- $s took $(( $(date +%s) - t0 )) s"; done
