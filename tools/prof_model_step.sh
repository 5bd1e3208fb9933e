# rocprofv3 --kernel-trace --stats of 12 SeqFormer-R50 training steps (tools/prof_model_step.py), collection starting after the
# warm-up (MIOpen's kernel search runs in the first steps):   gpurun -- 'bash tools/prof_model_step.sh r05'   then
#   python tools/summarize_model_step.py gpurun_out/prof_r05_model/model_kernel_stats.csv 12 profiles/r05_model_step_top_kernels.csv
set -x
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${R}_model${VNX_PROF_AUTOCAST:+_$VNX_PROF_AUTOCAST}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VNX_PROF_DELAY=${VNX_PROF_DELAY:-170} VNX_PROF_STEPS=6
rocprofv3 --kernel-trace --stats --collection-period ${VNX_PROF_DELAY}:600:1 --collection-period-unit sec --output-format csv \
  -d $OUT -o model -- python $GRAFT_REPO_ROOT/tools/prof_model_step.py > $OUT/model.log 2> $OUT/model.err
tail -3 $OUT/model.log
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
ls $OUT
