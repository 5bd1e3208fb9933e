"""Condense tools/prof_shapes.sh output into profiles/rNN_shapes_kernel_stats.csv (+ .json):
one row per (case, vnx kernel): calls, average / min / max microseconds by rocprofv3 --kernel-trace --stats.

python tools/summarize_shapes.py gpurun_out/prof_r02_shapes profiles/r02_shapes
"""
import csv, glob, json, os, re, sys

src, dst = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"vnx::((?:\w+::)*\w+)(<[^(]*>)?", name)
    return ("vnx::" + m.group(1) + (m.group(2) or "")) if m else name[:60]


rows, table = [], {}
for path in sorted(glob.glob(os.path.join(src, "*_kernel_stats.csv"))):
    case = os.path.basename(path)[: -len("_kernel_stats.csv")]
    log = os.path.join(src, case + ".log")
    note = ""
    if os.path.exists(log):
        head = [l for l in open(log) if l.startswith("shape=")]
        note = head[0].strip() if head else ""
    for r in csv.DictReader(open(path)):
        if "vnx::" not in r["Name"]:
            continue
        k = short(r["Name"])
        rows.append([case, k, r["Calls"], f'{float(r["AverageNs"]) / 1e3:.3f}', f'{float(r["MinNs"]) / 1e3:.3f}',
                     f'{float(r["MaxNs"]) / 1e3:.3f}', note])
        table.setdefault(case, {})[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
with open(dst + "_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["case", "kernel", "calls", "avg_us", "min_us", "max_us", "workload"])
    w.writerows(rows)
# HBM traffic per launch of the MSDA kernels at the non-headline shapes: FETCH_SIZE / WRITE_SIZE, separate passes
# (*_pmcfetch / *_pmcwrite).  Units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes:
# both counters are in KiB; FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 -> doubled.
pmc = {}
for path in sorted(glob.glob(os.path.join(src, "*_pmc*_counter_collection.csv"))):
    case = os.path.basename(path).split("_pmc")[0]
    acc = {}
    for r in csv.DictReader(open(path)):
        if "vnx::" in r["Kernel_Name"]:
            acc.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        mean = sum(v) / len(v)
        e = pmc.setdefault(case, {}).setdefault(k, {})
        e[c + "_KiB_raw"] = mean
        e["dispatches"] = len(v)
        if c == "FETCH_SIZE":
            e["read_MB"] = 2 * mean * 1024 / 1e6
        elif c == "WRITE_SIZE":
            e["write_MB"] = mean * 1024 / 1e6
for case in pmc.values():
    for e in case.values():
        if "read_MB" in e and "write_MB" in e:
            e["traffic_MB"] = e["read_MB"] + e["write_MB"]
json.dump({"kernel_trace": table, "pmc_hbm": pmc}, open(dst + "_kernel_avg_us.json", "w"), indent=1, sort_keys=True)
print(open(dst + "_kernel_stats.csv").read())
print(json.dumps(pmc, indent=1, sort_keys=True))
