set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r01b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -c 600 $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-model --no-warm --steps 50 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o pmc_write -- $B > /dev/null 2> $OUT/pmc_write.err
ls $OUT
rm -f $OUT/*.db $OUT/*kernel_trace.csv
