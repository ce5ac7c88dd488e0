#!/bin/bash
# Counter evidence for the batched decode engines and the wide step: separate rocprofv3 --pmc passes (kernel trace only) over three targets --
#   batch16 : tools/batch_prof.py 16          -> decode_engine_b16_kernel<1, false> (one 16-row group per launch)
#   share81 : tools/share_prof.py 8 0         -> decode_engine_b16_kernel<2, true> / <1, true> (continuous batch of a rank's share: two slot groups per launch)
#   wide64  : tools/wide_probe.py 4 64 4        -> q4_wide_kernel / wide_finish_kernel (four slot groups) and the forked skinny chains it replaces
#   gpurun -- 'bash tools/pmc_engines.sh [tag]'     (summaries -> gpurun_out/<tag>_pmc_<target>.txt)
TAG=${1:-r06}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp VOX_BATCH_NO_GRAPH=1
declare -A CMD=( [batch16]="tools/batch_prof.py 16" [share81]="tools/share_prof.py 8 0" [wide64]="tools/wide_probe.py 4 64 4" )
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
for tgt in ${TARGETS:-batch16 share81 wide64}; do
  dirs=""; i=0
  for set in "${SETS[@]}"; do
    i=$((i+1)); d=$OUT/${TAG}_pmc_${tgt}_$i
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python $REPO/${CMD[$tgt]} > $d.log 2>&1; echo "$tgt pass $i rc=$?"
    dirs="$dirs $d"
  done
  python $REPO/tools/pmc_summary.py $dirs > $OUT/${TAG}_pmc_${tgt}.txt 2>&1
  grep -A6 -E "^== .*(decode_engine|q4_wide|wide_finish)" $OUT/${TAG}_pmc_${tgt}.txt | head -60
  rm -rf $dirs      # (the raw counter CSVs are tens of MB: only the summaries travel back)
done
