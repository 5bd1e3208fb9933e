cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mh_prof; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o heads -- python $GRAFT_REPO_ROOT/tools/prof_heads.py > $OUT/heads.log 2> $OUT/heads.err
python3 - <<'P'
import csv, glob, collections, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/mh_prof/*kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "mask_head" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    v.sort()
    print(k, "calls", len(v), "avg %.2f us" % (sum(v) / len(v) / 1e3), "median %.2f" % (v[len(v) // 2] / 1e3))
P
find $OUT -name "*kernel_trace.csv" -delete
