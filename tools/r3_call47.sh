# re-trace of the tile2 forward (the refresh run recorded 97 us for it; 64.6 us in kbench)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_shapes
K=$GRAFT_REPO_ROOT/tools/kbench.bin
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for name in enc360_M_tile2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $name -- $K --shape enc360 --dist M --op fwd --variants 720 --inner 8 --reps 7 --cold-only > $OUT/$name.log 2> $OUT/$name.err
done
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
grep tile2 $OUT/enc360_M_tile2_kernel_stats.csv | cut -c1-120
