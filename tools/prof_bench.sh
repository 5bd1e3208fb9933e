# The command sequence behind profiles/rNN_bench_*: the bench line, rocprofv3 --kernel-trace --stats over the same
# command (cold launches only), then separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (MI355X_MICROARCH.md, HBM).
#   gpurun -- 'bash tools/prof_bench.sh r02'   then   python tools/summarize_prof.py gpurun_out/prof_r02 profiles/r02_bench
set -x
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -c 400 $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-model --no-warm --no-cases --no-spans --steps 50 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o pmc_write -- $B > /dev/null 2> $OUT/pmc_write.err
ls $OUT
rm -f $OUT/*.db $OUT/*kernel_trace.csv
