#!/bin/bash
# Same-box A/B of the hybrid mode's switches (ms per captured step, 60 steps each, interleaved twice):  bash tools/hybrid_knob_ab.sh
B="--precision hybrid --steps 60 --warmup 5 --no-cpu-baseline --no-product-loop --no-other-mode"
for round in 1 2; do
  for cfg in "" "SARSSL_HYBRID_DLN32=0" "SARSSL_HYBRID_CTX=0" "SARSSL_HYBRID_STEM4=0" "SARSSL_HYBRID_FFN2_FWD=0" "SARSSL_HYBRID_FFN2_BWD=0"; do
    ms=$(env $cfg SARSSL_BENCH_NO_TELEMETRY=1 python bench.py $B 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $round  ${cfg:-default}  $ms ms"
  done
done
