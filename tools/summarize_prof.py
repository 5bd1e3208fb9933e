"""Condense rocprofv3 outputs (gpurun_out/prof_rNN) into the small files committed under profiles/.

python tools/summarize_prof.py gpurun_out/prof_r01 profiles/r01
"""
import csv, collections, json, os, re, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def short(name):
    m = re.search(r"vnx::((?:\w+::)*\w+)(<[^(]*>)?", name)
    if m:
        return "vnx::" + m.group(1) + (m.group(2) or "")
    return name[:60]


rows = list(csv.DictReader(open(os.path.join(src, "trace_kernel_stats.csv"))))
with open(dst + "_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    r["MinNs"], r["MaxNs"], r["StdDev"]])

# kernel -> average microseconds of the traced command (what bench.py's roofline figures must agree with)
avg = {short(r["Name"]): {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])} for r in rows if "vnx::" in r["Name"]}
json.dump(avg, open(dst + "_kernel_avg_us.json", "w"), indent=1, sort_keys=True)

pmc = {}
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    path = os.path.join(src, f"{tag}_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "vnx::" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc.setdefault(k, {})[counter] = {"dispatches": len(v), "mean_KiB": sum(v) / len(v),
                                          "min_KiB": min(v), "max_KiB": max(v)}
for k, d in pmc.items():
    f_kib = d.get("FETCH_SIZE", {}).get("mean_KiB")
    w_kib = d.get("WRITE_SIZE", {}).get("mean_KiB")
    if f_kib is not None and w_kib is not None:
        # MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for
        # 16-B-per-lane loads -> double it; WRITE_SIZE is taken as reported (uncalibrated).
        d["hbm_bytes_per_launch_corrected"] = (2 * f_kib + w_kib) * 1024
        d["hbm_bytes_per_launch_raw"] = (f_kib + w_kib) * 1024
json.dump(pmc, open(dst + "_pmc_hbm.json", "w"), indent=1, sort_keys=True)
print(open(dst + "_kernel_stats.csv").read()[:1500])
print(json.dumps(pmc, indent=1)[:2500])
