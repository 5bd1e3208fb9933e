cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_model
mkdir -p $OUT
echo skip > gpurun_out/c18_pytest.log
cd /tmp && export TMPDIR=/tmp
VNX_PROF_DELAY=40 rocprofv3 --kernel-trace --stats --collection-period 40:300:1 --output-format csv -d $OUT -o model -- python $GRAFT_REPO_ROOT/tools/prof_model_step.py > $OUT/model.log 2> $OUT/model.err
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
cd $GRAFT_REPO_ROOT
./tools/kbench.bin --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --inner 8 > gpurun_out/c18_kbench.log 2>&1
tail -3 gpurun_out/c18_pytest.log
