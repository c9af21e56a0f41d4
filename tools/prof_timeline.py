"""Timeline of ONE graph replay from a rocprofv3 --kernel-trace CSV of `bench.py` (graph mode): start, duration, how many other kernels
were in flight when it started, name.  Shows which stream's chain is the critical path and where a stream runs alone.
    python tools/prof_timeline.py <kernel_trace.csv> [step]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ticks = [i for i, r in enumerate(rows) if "step_tick" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ticks) // 2
seg = rows[ticks[k]:ticks[k + 1]]
t0 = int(seg[0]["Start_Timestamp"])
ends = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    live = sum(1 for x in ends if x > s)
    ends.append(e)
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:64]
    print("%9.1f %8.1f  +%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, live, name))
print("window %.1f us" % ((int(rows[ticks[k + 1]]["Start_Timestamp"]) - t0) / 1e3))
