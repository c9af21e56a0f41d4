#!/bin/bash
# Per-kernel durations of tools/bench_rowkernels.py (rocprofv3 kernel trace): bash tools/row_trace.sh  (through gpurun, from the repo root)
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_row -o row -- python3 $ROOT/tools/bench_rowkernels.py > /tmp/row.out 2> /tmp/row.err
f=$(find /tmp/prof_row -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:30]:
    print("%-84s %5s %9.1f us" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
