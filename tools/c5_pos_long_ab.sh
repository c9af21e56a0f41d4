#!/bin/bash
# Config 5 (T = 624): the forward's in-kernel positional score per 256-key block against the score-GEMM launch, both 16-bit modes, same box.
#   bash tools/c5_pos_long_ab.sh
B="--workload config5 --steps 40 --warmup 3 --no-cpu-baseline --no-product-loop --no-other-mode"
for round in 1 2; do
  for prec in hybrid fp16; do
    for v in 1 0; do
      ms=$(SARSSL_ATTN_POS_LONG=$v SARSSL_BENCH_NO_TELEMETRY=1 python bench.py $B --precision $prec 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "round $round  $prec  SARSSL_ATTN_POS_LONG=$v  $ms ms"
    done
  done
done
