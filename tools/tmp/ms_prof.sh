cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04_model; rm -rf $OUT; mkdir -p $OUT
VNX_PROF_DELAY=45 timeout 600 rocprofv3 --kernel-trace --stats --collection-period 45:200:1 --output-format csv -d $OUT -o step -- python $GRAFT_REPO_ROOT/tools/prof_model_step.py > $OUT/step.log 2> $OUT/step.err
tail -2 $OUT/step.log
rm -f $OUT/*.db $OUT/*kernel_trace.csv
ls $OUT
