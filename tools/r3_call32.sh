# round 3, call 32: rows per wave of the mask-head backward (A/B builds)
cd /tmp && export TMPDIR=/tmp
for lib in product mbrows12 mbrows16 mbrows24; do
  if [ $lib = product ]; then unset VNX_HIP_LIB; else export VNX_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/$lib/libvnext_hip.so; fi
  ( cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_mask_head.py -m gpu -x -q 2>&1 | tail -1 )
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c32 -o heads_$lib -- python $GRAFT_REPO_ROOT/tools/prof_heads.py > $GRAFT_REPO_ROOT/gpurun_out/c32_$lib.log 2>&1
  echo $lib; grep -h "mask_head_bwd" $GRAFT_REPO_ROOT/gpurun_out/prof_c32/heads_${lib}_kernel_stats.csv | awk -F'",' '{print $2}'
done
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_c32/*.db $GRAFT_REPO_ROOT/gpurun_out/prof_c32/*kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/prof_c32/*agent_info.csv
