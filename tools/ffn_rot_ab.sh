#!/bin/bash
# Same-box A/B of the chunk rotation of the fused feed-forward launches (SARSSL_FFN_ROT; ms per captured step, 60 steps each, interleaved):
#   bash tools/ffn_rot_ab.sh
B="--steps 60 --warmup 5 --no-cpu-baseline --no-product-loop --no-other-mode"
for round in 1 2 3; do
  for p in hybrid fp16; do
    for r in 0 1; do
      ms=$(SARSSL_FFN_ROT=$r SARSSL_BENCH_NO_TELEMETRY=1 python bench.py --precision $p $B 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "round $round  $p SARSSL_FFN_ROT=$r  $ms ms"
    done
  done
done
