cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mh_pmc; rm -rf $OUT; mkdir -p $OUT
PMC_A="SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU"
PMC_B="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $PMC_A --output-format csv -d $OUT -o A -- python $GRAFT_REPO_ROOT/tools/tmp/mh_only.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc $PMC_B --output-format csv -d $OUT -o B -- python $GRAFT_REPO_ROOT/tools/tmp/mh_only.py > /dev/null 2>&1
python3 - <<'P'
import csv, glob, collections, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/mh_pmc/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "mask_head" not in r["Kernel_Name"]: continue
        key = r.get("Grid_Size") or r.get("Grid_Size_X")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_WAIT_ANY"): cnt[key] += 1
    for k, d in acc.items():
        print(os.path.basename(f), "grid", k, "dispatches", cnt[k], {n: round(v / max(cnt[k], 1)) for n, v in d.items()})
P
rm -f $OUT/*.db
