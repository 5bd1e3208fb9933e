# round 3, final profile refresh: bench line + trace + PMC, every other shape + heads, model step
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh r03 > gpurun_out/prof_bench.log 2>&1
bash tools/prof_shapes.sh r03 > gpurun_out/prof_shapes.log 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_model
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
VNX_PROF_DELAY=40 rocprofv3 --kernel-trace --stats --collection-period 40:300:1 --output-format csv -d $OUT -o model -- python $GRAFT_REPO_ROOT/tools/prof_model_step.py > $OUT/model.log 2> $OUT/model.err
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
tail -c 300 $GRAFT_REPO_ROOT/gpurun_out/prof_r03/bench_line.json
