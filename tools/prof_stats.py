"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (csv output)."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time (ms): %.3f over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-72s calls %6s  total_ms %9.3f  avg_us %9.1f  %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                  float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
