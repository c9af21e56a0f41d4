#!/bin/bash
# Per-kernel durations of tools/bench_attn.py (rocprofv3 kernel trace): bash tools/attn_trace.sh  (through gpurun, from the repo root)
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_attn -o attn -- python3 $ROOT/tools/bench_attn.py --iters 20 > /tmp/attn.out 2> /tmp/attn.err
f=$(find /tmp/prof_attn -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print("%-84s %5s %9.1f us" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
