cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c5; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- $K --shape enc360 --dist M --op bwd --variants 0 --inner 8 --reps 5 --cold-only > $OUT/kt.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kt_kernel_stats.csv")):
    print("%-70s calls %6s avg %9.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
