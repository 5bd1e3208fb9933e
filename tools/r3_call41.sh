# round 3, call 39: why is the record-fed grad_value kernel slower for bf16 values?  PMC, decoder-720p, fp32 vs bf16
cd /tmp && export TMPDIR=/tmp
K=$GRAFT_REPO_ROOT/tools/kbench.bin
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c41; rm -rf $OUT; mkdir -p $OUT
for dt in f32 bf16; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT -o a_$dt -- $K --shape enc360 --dtype $dt --dist M --op bwd --variants 0 --inner 2 --reps 2 --cold-only > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT --output-format csv -d $OUT -o b_$dt -- $K --shape enc360 --dtype $dt --dist M --op bwd --variants 0 --inner 2 --reps 2 --cold-only > /dev/null 2>&1
done
rm -f $OUT/*.db $OUT/*agent_info.csv
python3 - <<PY
import csv, glob, os, collections
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "gv_tiles" not in r["Kernel_Name"] and "bwd_d32" not in r["Kernel_Name"]: continue
        kk = "K2" if "gv_tiles" in r["Kernel_Name"] else "K1"
        agg[kk + ":" + r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[kk + ":" + r["Counter_Name"]][1] += 1
    print(os.path.basename(f), " ".join("%s=%.0f" % (k, v / n) for k, (v, n) in sorted(agg.items())))
PY
