#!/bin/bash
# Config 5 (4 microphones x 10 s, T = 624), single stream: rocprofv3 kernel trace -> per-kernel step breakdown.
#   bash tools/collect_c5_profile.sh r06        (through gpurun, from the repo root)
set -u
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SARSSL_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_${TAG}_c5 -o c5 -- python3 $ROOT/bench.py --workload config5 --steps 12 --warmup 3 --no-cpu-baseline --no-product-loop --no-other-mode > $OUT/${TAG}_c5_1stream_line.json 2> $OUT/${TAG}_c5.err
cd $ROOT
TRACE=$(find $OUT/prof_${TAG}_c5 -name "*kernel_trace.csv" | head -1)
[ -n "$TRACE" ] && python3 tools/prof_step.py $TRACE > $OUT/${TAG}_c5_1stream_step_breakdown.txt 2>&1
rm -rf $OUT/prof_${TAG}_c5/*/*.db 2>/dev/null
head -40 $OUT/${TAG}_c5_1stream_step_breakdown.txt | cut -c1-150
