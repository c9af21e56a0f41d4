#!/bin/bash
# Collects the round's profile evidence on an MI355X box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r04
# -> gpurun_out/<tag>_*: rocprofv3 kernel trace + stats of bench.py's graph replays, the per-step breakdown, the PMC counter passes
#    (separate rocprofv3 --pmc runs, tools/prof_counters.py), the default bench line.  Copy what is to be judged into profiles/.
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/prof_counters.py --build-marker
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$TAG -o b64 -- python3 $ROOT/bench.py --steps 22 --warmup 3 --no-cpu-baseline --no-product-loop --no-other-mode \
    > $OUT/${TAG}_bench_b64_line_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
cd $ROOT
TRACE=$(find $OUT/prof_$TAG -name "*kernel_trace.csv" | head -1)
STATS=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$TRACE" ] && python3 tools/prof_step.py $TRACE > $OUT/${TAG}_bench_b64_step_breakdown.txt 2>&1
[ -n "$STATS" ] && cp $STATS $OUT/${TAG}_bench_b64_kernel_stats.csv
python3 tools/prof_counters.py --out $OUT/${TAG}_kernel_counters.json --scratch $OUT/pmc_$TAG > $OUT/${TAG}_counters.log 2>&1
python3 bench.py > $OUT/${TAG}_bench_default_line.json 2> $OUT/${TAG}_bench_default.err
SARSSL_PROF_SHAPES=1 python3 tools/step_gemm_table.py --precision hybrid > $OUT/${TAG}_step_gemm_table.txt 2>&1
SARSSL_PROF_SHAPES=1 python3 tools/step_gemm_table.py --precision fp16 > $OUT/${TAG}_step_gemm_table_fp16.txt 2>&1
python3 tools/step_stamps.py --precision hybrid > $OUT/${TAG}_two_stream_timeline.txt 2>&1
python3 tools/step_stamps.py --precision fp16 > $OUT/${TAG}_two_stream_timeline_fp16.txt 2>&1
rm -rf $OUT/prof_$TAG/*/*.db 2>/dev/null
du -sh $OUT/prof_$TAG $OUT/pmc_$TAG 2>/dev/null
head -c 600 $OUT/${TAG}_bench_default_line.json; echo
head -5 $OUT/${TAG}_bench_b64_step_breakdown.txt
