cd /tmp && export TMPDIR=/tmp
K=$GRAFT_REPO_ROOT/tools/kbench.bin
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_w
rm -rf $O; mkdir -p $O
$K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 9 --check
$K --shape dec360 --dist M --op bwd --variants 0 --inner 24 --reps 9 --check
$K --shape enc360 --dist M --op bwd --variants 0 --inner 8 --reps 7 --check
$K --shape dec720 --dist U --op bwd --variants 0 --inner 8 --reps 7 --check
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o w_0 -- $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 3 --cold-only > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o f_0 -- $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 3 --cold-only > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t_0 -- $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 9 --cold-only > /dev/null 2>&1
rm -f $O/*.db
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_w"
for f in sorted(glob.glob(O+"/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "vnx" in r["Kernel_Name"]: acc[(r["Kernel_Name"][10:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(os.path.basename(f)[:6], k, round(sum(v)/len(v)), min(v), max(v), len(v))
for f in sorted(glob.glob(O+"/t_*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "vnx" in r["Name"]: print(r["Name"][10:60], r["Calls"], float(r["AverageNs"])/1e3)
PY
