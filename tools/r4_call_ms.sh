cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
VNX_CUDNN_BENCHMARK=0 python tools/prof_model_step.py 2>&1 | tail -2
VNX_CUDNN_BENCHMARK=1 python tools/prof_model_step.py 2>&1 | tail -2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04_model; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
VNX_PROF_STEPS=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o model -- python $GRAFT_REPO_ROOT/tools/prof_model_step.py > $OUT/model.log 2>&1
tail -2 $OUT/model.log
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
ls $OUT
