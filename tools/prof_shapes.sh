# rocprofv3 --kernel-trace --stats for the shapes and kernel families the headline trace does not cover
# (encoder 360p / 720p, decoder 720p, B = 10, model-like locations, bf16; mask head / reid / tracker / add_norm kernels),
# and FETCH_SIZE / WRITE_SIZE PMC passes for the encoder and decoder-720p shapes.
# Every traced MSDA launch reads cold inputs (tools/kbench.hip --cold-only, rotating > 320 MiB of input sets).
#   gpurun -- 'bash tools/prof_shapes.sh r03'   then   python tools/summarize_shapes.py gpurun_out/prof_r03_shapes profiles/r03_shapes
set -x
R=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${R}_shapes
mkdir -p $OUT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
cd /tmp && export TMPDIR=/tmp
run() {   # name, kbench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $name -- $K "$@" --cold-only > $OUT/$name.log 2> $OUT/$name.err
}
pmc() {   # name, kbench args...: separate FETCH_SIZE and WRITE_SIZE passes (MI355X_MICROARCH.md, HBM / PMC slots)
  name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o ${name}_pmcfetch -- $K "$@" --cold-only --inner 2 --reps 2 > /dev/null 2> $OUT/${name}_pmcfetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o ${name}_pmcwrite -- $K "$@" --cold-only --inner 2 --reps 2 > /dev/null 2> $OUT/${name}_pmcwrite.err
}
run dec360_U      --shape dec360 --dist U --op both --variants 0 --inner 24 --reps 9
run dec360_M      --shape dec360 --dist M --op both --variants 0 --inner 24 --reps 9
run dec360_U_B10  --shape dec360 --dist U --B 10 --op both --variants 0 --inner 12 --reps 9
run dec720_U      --shape dec720 --dist U --op both --variants 0 --inner 8 --reps 9
run dec720_U_bf16 --shape dec720 --dtype bf16 --dist U --op both --variants 0 --inner 8 --reps 9
run enc360_M      --shape enc360 --dist M --op both --variants 0 --inner 8 --reps 7
run enc360_M_rec  --shape enc360 --dist M --op bwd --variants 430 --inner 8 --reps 7
run enc360_M_bf16 --shape enc360 --dtype bf16 --dist M --op both --variants 0 --inner 8 --reps 7
run enc360_M_tile --shape enc360 --dist M --op fwd --variants 700 --inner 8 --reps 7
run enc360_M_tile2 --shape enc360 --dist M --op fwd --variants 720 --inner 8 --reps 7
run enc720_M      --shape enc720 --dist M --op both --variants 0 --inner 4 --reps 5
run enc720_M_B2   --shape enc720 --B 2 --dist M --op both --variants 0 --inner 4 --reps 7
run enc360_M_fused --shape enc360 --dist M --op fbwd --variants 0 --inner 8 --reps 7
run enc720_M_B2_bf16 --shape enc720 --dtype bf16 --B 2 --dist M --op both --variants 0 --inner 4 --reps 5
pmc enc360_M      --shape enc360 --dist M --op both --variants 0
pmc enc720_M      --shape enc720 --dist M --op both --variants 0
pmc dec720_U      --shape dec720 --dist U --op both --variants 0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o heads -- python $GRAFT_REPO_ROOT/tools/prof_heads.py > $OUT/heads.log 2> $OUT/heads.err
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
ls $OUT
