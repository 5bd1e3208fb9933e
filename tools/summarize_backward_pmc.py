"""gpurun_out/prof_rNN_bwd (tools/prof_backward_pmc.sh) -> profiles/rNN_backward_pmc.csv: per backward kernel and shape the
rocprofv3 duration, the raw counters per launch (summed over the device) and the derived columns of
profiles/r03_forward_encoder360_pmc.csv (VALU issue share, LDS busy, bank-conflict share, wait share, instructions per wave)."""
import collections
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
KERNELS = ("msda_fwd_slab_kernel", "msda_bwd_slab_kernel", "msda_fwd_d32_kernel", "msda_bwd_pair_kernel", "msda_bwd_d32_kernel", "msda_bwd_gv_direct_kernel", "msda_bwd_gv_sel_kernel", "msda_bwd_gv_tiles_kernel", "gv_split_finish_kernel", "dynamic_mask_head_bwd_kernel",
           "dynamic_mask_head_runs_kernel", "zero3_kernel")
SAMPLES = {"dec360": 128 * 5 * 300, "enc360": 128 * 5 * 5100}
GHZ = 2.0          # the clock the derived shares assume (as in r03_forward_encoder360_pmc.csv)


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


dur = {}
for f in glob.glob(os.path.join(src, "kt_*kernel_stats.csv")):
    case = os.path.basename(f).split("_")[1]
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            tot, calls = dur.get((case, k), (0.0, 0))
            dur[(case, k)] = (tot + float(r["TotalDurationNs"]), calls + int(r["Calls"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
for f in glob.glob(os.path.join(src, "pmc*_counter_collection.csv")):
    case = os.path.basename(f).split("_")[1]
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            c = cnt[(case, k)][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1].add(r["Dispatch_Id"])
names = sorted({c for v in cnt.values() for c in v})
rows = []
for key in sorted(cnt):
    case, k = key
    per = {c: (v[0] / max(1, len(v[1]))) for c, v in cnt[key].items()}
    d = dur.get(key)
    us = d[0] / d[1] / 1e3 if d and d[1] else None
    row = {"case": case, "kernel": k, "duration_us_rocprofv3": "%.2f" % us if us else "", "launches_timed": d[1] if d else ""}
    row.update({c: "%.1f" % per.get(c, float("nan")) for c in names})
    valu, waves = per.get("SQ_INSTS_VALU"), per.get("SQ_WAVES")
    if valu and waves:
        row["valu_per_wave"] = "%.0f" % (valu / waves)
    if valu and case in SAMPLES:
        row["valu_per_sample"] = "%.2f" % (valu / SAMPLES[case])
    if valu and us:
        row["valu_issue_frac"] = "%.3f" % (valu * 4 / 1024 / (us * 1e3 * GHZ))
    if us and per.get("SQ_LDS_IDX_ACTIVE") is not None:
        row["lds_busy"] = "%.3f" % (per["SQ_LDS_IDX_ACTIVE"] / 256 / (us * 1e3 * GHZ))
    if per.get("SQ_LDS_IDX_ACTIVE"):
        row["conflict_frac"] = "%.3f" % (per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"])
    if per.get("SQ_WAVE_CYCLES") and per.get("SQ_WAIT_ANY") is not None:
        row["wait_frac"] = "%.3f" % (per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"])
    rows.append(row)
cols = ["case", "kernel", "duration_us_rocprofv3", "launches_timed"] + names + \
       ["valu_per_wave", "valu_per_sample", "valu_issue_frac", "lds_busy", "conflict_frac", "wait_frac"]
with open(dst, "w", newline="") as fh:
    fh.write("# backward kernels, cold inputs (tools/prof_backward_pmc.sh: rocprofv3 --kernel-trace --stats, then two --pmc passes per case); "
             "counters are per launch, summed over the device\n")
    fh.write("# cases: dec360 = the headline call (B=5, Lq=300, 360p, uniform locations); enc360 = encoder-360p (B=5, Lq=S=5100, model-like "
             "locations); heads = tools/prof_heads.py (mask head: training shape 5 frames x 24 instances, and 300-instance inference frames)\n")
    fh.write("# derived: valu_per_wave = SQ_INSTS_VALU / SQ_WAVES; valu_issue_frac = SQ_INSTS_VALU * 4 clk / 1024 SIMDs / (duration * %.1f GHz); "
             "lds_busy = SQ_LDS_IDX_ACTIVE / 256 CUs / (duration * %.1f GHz); conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
             "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES\n" % (GHZ, GHZ))
    w = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore")
    w.writeheader()
    w.writerows(rows)
print(open(dst).read())
