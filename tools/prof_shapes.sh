# rocprofv3 --kernel-trace --stats for the shapes and kernel families the headline trace does not cover
# (encoder 360p / 720p, decoder 720p, B = 10, model-like locations; mask head / reid / tracker kernels).
# Every traced MSDA launch reads cold inputs (tools/kbench.hip --cold-only, rotating > 320 MiB of input sets).
#   gpurun -- 'bash tools/prof_shapes.sh r02'   then   python tools/summarize_shapes.py gpurun_out/prof_r02_shapes profiles/r02_shapes
set -x
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${R}_shapes
mkdir -p $OUT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
cd /tmp && export TMPDIR=/tmp
run() {   # name, kbench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $name -- $K "$@" --cold-only > $OUT/$name.log 2> $OUT/$name.err
}
run dec360_U      --shape dec360 --dist U --op both --variants 0 --inner 24 --reps 9
run dec360_M      --shape dec360 --dist M --op both --variants 0 --inner 24 --reps 9
run dec360_U_B10  --shape dec360 --dist U --B 10 --op both --variants 0 --inner 12 --reps 9
run dec720_U      --shape dec720 --dist U --op both --variants 0 --inner 8 --reps 9
run enc360_M      --shape enc360 --dist M --op both --variants 0 --inner 8 --reps 7
run enc360_M_tile --shape enc360 --dist M --op fwd --variants 700 --inner 8 --reps 7
run enc720_M      --shape enc720 --dist M --op both --variants 0 --inner 4 --reps 5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o heads -- python $GRAFT_REPO_ROOT/tools/time_heads.py > $OUT/heads.log 2> $OUT/heads.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o heads_pmc -- python $GRAFT_REPO_ROOT/tools/time_heads.py > /dev/null 2> $OUT/heads_pmc.err
rm -f $OUT/*.db $OUT/*kernel_trace.csv
ls $OUT
