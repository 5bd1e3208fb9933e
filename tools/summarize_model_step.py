"""python tools/summarize_model_step.py <dir with model_kernel_stats.csv> <steps> <dst csv>: top kernels of the training
step by time per step (rocprofv3 --kernel-trace --stats of tools/prof_model_step.py)."""
import csv, sys

src, steps, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = []
for r in csv.DictReader(open(src)):
    rows.append((float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"]))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
with open(dst, "w") as f:
    w = csv.writer(f)
    w.writerow(["ms_per_step", "calls_per_step", "avg_us", "kernel"])
    for ms, calls, avg, name in rows[:60]:
        w.writerow([f"{ms:.3f}", f"{calls:.1f}", f"{avg:.2f}", name[:140]])
    w.writerow([f"{total:.3f}", f"{sum(r[1] for r in rows):.0f}", "", "TOTAL (all kernels)"])
print(open(dst).read())
