REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out; cd /tmp; export TMPDIR=/tmp VOX_BATCH_NO_GRAPH=1
for tgt in batch16 share81; do
  if [ $tgt = batch16 ]; then cmd="tools/batch_prof.py 16"; else cmd="tools/share_prof.py 8 0"; fi
  d=$OUT/r06_pmc_dram_$tgt
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $d -o p -- python $REPO/$cmd > $d.log 2>&1; echo "$tgt rc=$?"
  VOX_PMC_TOP=80 python $REPO/tools/pmc_summary.py $d | tee $OUT/r06_pmc_dram_$tgt.txt | grep -A2 -E "^== .*decode_engine" ; rm -rf $d
done
