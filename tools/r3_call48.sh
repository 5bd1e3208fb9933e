# round 3, call 48: instruction accounting of the three headline kernels (PMC)
cd /tmp && export TMPDIR=/tmp
K=$GRAFT_REPO_ROOT/tools/kbench.bin
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c48; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT -o a -- $K --shape dec360 --dist U --op both --variants 0 --inner 4 --reps 2 --cold-only > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d $OUT -o b -- $K --shape dec360 --dist U --op both --variants 0 --inner 4 --reps 2 --cold-only > /dev/null 2>&1
rm -f $OUT/*.db $OUT/*agent_info.csv
python3 - <<PY
import csv, glob, os, collections
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        kk = "fwd" if "fwd_d32" in n else "K1" if "bwd_d32" in n else "K2" if "gv_sel" in n else None
        if kk is None: continue
        agg[(kk, r["Counter_Name"])][0] += float(r["Counter_Value"]); agg[(kk, r["Counter_Name"])][1] += 1
    for kk in ("fwd", "K1", "K2"):
        print(os.path.basename(f)[:1], kk, " ".join("%s=%.0f" % (c[3:], v / n) for (k2, c), (v, n) in sorted(agg.items()) if k2 == kk))
PY
