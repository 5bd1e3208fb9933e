# PMC of the encoder-shape forward kernels: gather (0), tiled v1 (700), tiled v2 (720)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_fwd_${1:-a}
mkdir -p $OUT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
SHAPE=${2:-enc360}
VARS=${3:-0 700 720}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt_$v -- $K --shape $SHAPE --dist M --op fwd --variants $v --inner 8 --reps 5 --cold-only > $OUT/kt_$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT -o pmc_$v -- $K --shape $SHAPE --dist M --op fwd --variants $v --inner 2 --reps 2 --cold-only > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -o pmc2_$v -- $K --shape $SHAPE --dist M --op fwd --variants $v --inner 2 --reps 2 --cold-only > /dev/null 2>&1
done
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for f in sorted(glob.glob(out + "/*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "msda" in n:
            print("%-24s %-60s calls %5s avg %9.2f us" % (os.path.basename(f)[:24], n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
for f in sorted(glob.glob(out + "/p*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "msda" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:44], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    print(os.path.basename(f))
    for (kn, cn), (v, n) in sorted(agg.items()):
        print("   %-44s %-22s %14.1f per launch (%d)" % (kn, cn, v / n, n))
PY
