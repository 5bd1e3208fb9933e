# PMC of the BACKWARD kernels (VERDICT r3 "What's missing" item 4): msda_bwd_d32_kernel (grad_loc / grad_attn),
# msda_bwd_gv_sel_kernel (record-fed grad_value), msda_bwd_gv_tiles_kernel (tile-fed grad_value) at the headline shape
# (decoder-360p, B = 5, uniform locations) and at encoder-360p (B = 5, model-like locations), and the mask head's
# backward kernel at the training shape.  Per case: one `--kernel-trace --stats` pass (durations) and two `--pmc` passes
# (separate runs: gpurun refuses counters combined with traces).  Cold launches only (tools/kbench.hip rotates inputs).
#   tools/prof_backward_pmc.sh r04   ->  gpurun_out/prof_r04_bwd/ ;  tools/summarize_backward_pmc.py writes profiles/r04_backward_pmc.csv
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_bwd
rm -rf $OUT; mkdir -p $OUT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
cd /tmp && export TMPDIR=/tmp
PMC_A="SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU"
PMC_B="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for c in "dec360 U bwd" "enc360 M both"; do      # (encoder shape: the forward too -- msda_fwd_slab_kernel, round 5)
  set -- $c
  ARGS="--shape $1 --dist $2 --op $3 --variants 0 --cold-only"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt_$1 -- $K $ARGS --inner 8 --reps 5 > $OUT/kt_$1.log 2>&1
  timeout 300 rocprofv3 --pmc $PMC_A --output-format csv -d $OUT -o pmcA_$1 -- $K $ARGS --inner 2 --reps 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $PMC_B --output-format csv -d $OUT -o pmcB_$1 -- $K $ARGS --inner 2 --reps 2 > /dev/null 2>&1
done
# the mask head's backward at the training shape (5 frames x 24 matched instances, 360p): tools/prof_heads.py
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt_heads -- python $GRAFT_REPO_ROOT/tools/prof_heads.py > $OUT/kt_heads.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --pmc $PMC_A --output-format csv -d $OUT -o pmcA_heads -- python $GRAFT_REPO_ROOT/tools/prof_heads.py > /dev/null 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --pmc $PMC_B --output-format csv -d $OUT -o pmcB_heads -- python $GRAFT_REPO_ROOT/tools/prof_heads.py > /dev/null 2>&1 )
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
# keep only the rows of this library's kernels in the counter files (the heads run carries thousands of ATen dispatches)
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "vnx" in r["Kernel_Name"] or "msda" in r["Kernel_Name"] or "dynamic_mask" in r["Kernel_Name"]]
    if rows:
        w = csv.DictWriter(open(f, "w", newline=""), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
python3 $GRAFT_REPO_ROOT/tools/summarize_backward_pmc.py $OUT $GRAFT_REPO_ROOT/gpurun_out/${TAG}_backward_pmc.csv
