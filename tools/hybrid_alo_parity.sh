#!/bin/bash
# Per setting of SARSSL_HYBRID_ALO: the hybrid parity tests (measured values) and two 60-step bench runs:  bash tools/hybrid_alo_parity.sh
for cfg in "qkv,pw1,dec1" "ffn1,pw1,dec1" "pw1,dec1"; do
  echo "=== ALO=$cfg"
  rm -f gpurun_out/parity_measured.jsonl
  SARSSL_HYBRID_ALO=$cfg python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -q -m gpu -k hybrid 2>&1 | grep "passed\|failed\|FAILED"
  python - <<'P'
import json
for l in open("gpurun_out/parity_measured.jsonl"):
    r = json.loads(l)
    if "hybrid" in r["name"] and any(k in r["name"] for k in ("per_bin", "gradnorm", "curve")):
        print("  %-110s %.3e (tol %.1e)" % (r["name"][:110], r["measured"], r["tol"]))
P
  for i in 1 2; do
  SARSSL_HYBRID_ALO=$cfg SARSSL_BENCH_NO_TELEMETRY=1 python bench.py --precision hybrid --steps 60 --warmup 5 --no-cpu-baseline --no-product-loop --no-other-mode 2>/dev/null | python -c "import json,sys; print('  ms', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  done
done
