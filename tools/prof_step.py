"""Per-step view of a rocprofv3 --kernel-trace CSV of `bench.py` in graph-replay mode: the steps are delimited by the graph's first
node (step_tick_kernel); prints launches per step, the window, how much of it at least one kernel was running, the sum of kernel
durations (> window: the two encoder streams overlap) and the per-kernel breakdown per step.
    python tools/prof_step.py <kernel_trace.csv> [first_step] [nsteps]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ticks = [i for i, r in enumerate(rows) if "step_tick" in r["Kernel_Name"]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, len(ticks) - first - 1)
n = min(n, len(ticks) - first - 1)
seg = rows[ticks[first]:ticks[first + n]]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[ticks[first + n]]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
# how many kernels are in flight, as shares of the window (sweep over start / end events)
ev = sorted([(s_, 1) for s_, _ in iv] + [(e_, -1) for _, e_ in iv])
depth, last, share = 0, t0, {}
for t_, d_ in ev:
    t_ = min(max(t_, t0), t1)
    share[depth] = share.get(depth, 0) + (t_ - last)
    last, depth = t_, depth + d_
share[depth] = share.get(depth, 0) + (t1 - last)
tot = {}
for r in seg:
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:72]
    a = tot.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ks = sum(a[1] for a in tot.values())
print("%d graph replays: %.1f launches per step, window %.3f ms per step, >= 1 kernel running %.1f %% of it, sum of kernel durations %.3f ms per step"
      % (n, len(seg) / n, (t1 - t0) / n / 1e6, 100.0 * busy / (t1 - t0), ks / n / 1e6))
print("kernels in flight (share of the window): " + ", ".join("%d: %.1f %%" % (d, 100.0 * v / (t1 - t0)) for d, v in sorted(share.items()) if v > 0))
for k, a in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-74s %5.1f/step %8.1f us/step  avg %7.1f us  %4.1f%%" % (k, a[0] / n, a[1] / n / 1e3, a[1] / a[0] / 1e3, 100.0 * a[1] / ks))
