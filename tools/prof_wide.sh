# per-kernel durations of the wide decode step (rocprofv3 kernel trace, eager launches):  gpurun -- 'bash tools/prof_wide.sh [G] [tag]'
G=${1:-4}; TAG=${2:-r06}; cd /tmp; export TMPDIR=/tmp
VOX_BATCH_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_wide$G -o p -- python $GRAFT_REPO_ROOT/tools/wide_probe.py $G 70 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_wide$G.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_wide$G -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_wide${G}_kernel_stats.csv
grep -E "wide|attn_decode_gqa|argmax_embed_slots" $f | cut -c1-170
grep -E "^wide|^chains|ids identical" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_wide$G.log
cd $GRAFT_REPO_ROOT; timeout 300 python tools/wide_probe.py $G 70 2>&1 | grep -E "^wide|^chains|ids identical"
