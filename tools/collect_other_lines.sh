#!/bin/bash
# The round's other bench lines + the single-stream kernel trace (run through gpurun from the repo root):
#   bash tools/collect_other_lines.sh r06
# -> gpurun_out/<tag>_other_bench_lines.jsonl: --precision fp16 (the fast mode), fp32 (the reference CLI's default without --use-amp), bf16,
#    --workload config5 (hybrid and fp16), the data-parallel step over a ONE-rank RCCL process group (segmented replay / native exchange);
#    gpurun_out/<tag>_1stream_step_breakdown.txt: tools/prof_step.py on a rocprofv3 kernel trace with SARSSL_TWO_STREAMS=0 (every launch alone)
set -u
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
L=$OUT/${TAG}_other_bench_lines.jsonl
: > $L
B="--no-cpu-baseline --no-product-loop --no-other-mode"
python3 bench.py --precision fp16 --steps 100 --warmup 5 $B >> $L 2>$OUT/ol_fp16.err
python3 bench.py --precision fp32 --steps 30 --warmup 3 $B >> $L 2>$OUT/ol_fp32.err
python3 bench.py --precision bf16 --steps 100 --warmup 5 $B >> $L 2>$OUT/ol_bf16.err
python3 bench.py --workload config5 --steps 50 --warmup 5 $B >> $L 2>$OUT/ol_c5.err
python3 bench.py --workload config5 --precision fp16 --steps 50 --warmup 5 $B >> $L 2>$OUT/ol_c5_fp16.err
SARSSL_DIST_FORCE=1 python3 bench.py --steps 100 --warmup 5 $B >> $L 2>$OUT/ol_dp.err
SARSSL_DIST_FORCE=1 SARSSL_NATIVE_RCCL=1 python3 bench.py --steps 100 --warmup 5 $B >> $L 2>$OUT/ol_dpn.err
cd /tmp
SARSSL_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_${TAG}_1s -o s1 -- python3 $ROOT/bench.py --steps 22 --warmup 3 $B > $OUT/${TAG}_1stream_line.json 2> $OUT/${TAG}_1s.err
cd $ROOT
TRACE=$(find $OUT/prof_${TAG}_1s -name "*kernel_trace.csv" | head -1)
[ -n "$TRACE" ] && python3 tools/prof_step.py $TRACE > $OUT/${TAG}_1stream_step_breakdown.txt 2>&1
rm -rf $OUT/prof_${TAG}_1s/*/*.db 2>/dev/null
cut -c1-260 $L
head -3 $OUT/${TAG}_1stream_step_breakdown.txt
