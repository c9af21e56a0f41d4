"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals, launches per step, and how busy the GPU was inside the
traced window (union of kernel intervals / window)  ->  python tools/prof_stats.py <kernel_trace.csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t0, t1 = iv[0][0], max(e for _, e in iv)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for r in rows:
    n = r["Kernel_Name"]
    a = tot.setdefault(n, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ksum = sum(a[1] for a in tot.values())
print("window %.2f ms, union of kernel intervals %.2f ms (%.1f %% busy), sum of kernel durations %.2f ms; %d launches (%.1f per step over %d steps)"
      % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), ksum / 1e6, len(rows), len(rows) / steps, steps))
for n, a in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-100s calls %6d (%.1f/step) total_ms %9.3f avg_us %8.1f %5.1f%%" % (n[:100], a[0], a[0] / steps, a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / ksum))
