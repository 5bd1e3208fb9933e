# round 3, call 30: mask-head backward as two waves per strip
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_mask_head.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/c30_pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c30 -o heads -- python $GRAFT_REPO_ROOT/tools/prof_heads.py > $GRAFT_REPO_ROOT/gpurun_out/c30_heads.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/prof_c30/*.db gpurun_out/prof_c30/*kernel_trace.csv gpurun_out/prof_c30/*agent_info.csv
tail -3 gpurun_out/c30_pytest.log; grep -h "mask_head\|layernorm" gpurun_out/prof_c30/*kernel_stats.csv | cut -c1-160
