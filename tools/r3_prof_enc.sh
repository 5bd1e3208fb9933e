# per-kernel times + PMC of the encoder-shape backward, records path (430) vs tile path (0)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_enc_${1:-a}
mkdir -p $OUT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
SHAPE=${2:-enc360}
VARS=${3:-430 0}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt_$v -- $K --shape $SHAPE --dist M --op bwd --variants $v --inner 8 --reps 5 --cold-only > $OUT/kt_$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT -o pmc_$v -- $K --shape $SHAPE --dist M --op bwd --variants $v --inner 2 --reps 2 --cold-only > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pf_$v -- $K --shape $SHAPE --dist M --op bwd --variants $v --inner 2 --reps 2 --cold-only > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o pw_$v -- $K --shape $SHAPE --dist M --op bwd --variants $v --inner 2 --reps 2 --cold-only > /dev/null 2>&1
done
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for f in sorted(glob.glob(out + "/*kernel_stats.csv")):
    print(os.path.basename(f))
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "msda" in n:
            print("   %-60s calls %5s avg %9.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
for f in sorted(glob.glob(out + "/p*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "msda" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:50], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    print(os.path.basename(f))
    for (kn, cn), (v, n) in sorted(agg.items()):
        print("   %-50s %-22s %14.1f per launch (%d)" % (kn, cn, v / n, n))
PY
