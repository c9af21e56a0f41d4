#!/bin/bash
# Same-box A/B of which LayerNorm-fed Linear layers contract the activation as a pair (SARSSL_HYBRID_ALO; ms per captured step, 60 steps
# each, interleaved twice), with the fp16 mode next to it:  bash tools/hybrid_alo_ab.sh
B="--steps 60 --warmup 5 --no-cpu-baseline --no-product-loop --no-other-mode"
for round in 1 2; do
  for cfg in "pw1,dec1" "ffn1,qkv,pw1,dec1" "dec1" "pw1" "none" "fp16"; do
    if [ "$cfg" = fp16 ]; then e="SARSSL_X=0"; p=fp16; else e="SARSSL_HYBRID_ALO=$cfg"; p=hybrid; fi
    ms=$(env $e SARSSL_BENCH_NO_TELEMETRY=1 python bench.py --precision $p $B 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $round  $p ALO=$cfg  $ms ms"
  done
done
