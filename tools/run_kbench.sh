cd /tmp && export TMPDIR=/tmp
K=$GRAFT_REPO_ROOT/tools/kbench.bin
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_nt
mkdir -p $O
for lib in lib lib/expl; do
n=$(basename $lib)
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/$lib rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o w_$n -- $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 5 --cold-only > /dev/null 2>&1
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/$lib rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o f_$n -- $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 5 --cold-only > /dev/null 2>&1
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t_$n -- $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 9 --cold-only 2>/dev/null | grep variant
done
rm -f $O/*.db
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_nt"
for f in sorted(glob.glob(O+"/*_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(os.path.basename(f)[:8], k, round(sum(v)/len(v)), len(v))
for f in sorted(glob.glob(O+"/t_*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        print(os.path.basename(f)[:8], r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3)
PY
