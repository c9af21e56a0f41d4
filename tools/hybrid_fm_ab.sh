#!/bin/bash
# Tile height of the pair products (sarssl_gemm_split, csrc/gemm.hip launch_seg): the library's rule against both forced heights, per shape,
# inside single-stream training steps.    bash tools/hybrid_fm_ab.sh
for fm in 0 1 2; do
  echo "== SARSSL_SPLIT_FM=$fm (0: the library's rule)"
  SARSSL_SPLIT_FM=$fm python tools/step_gemm_table.py --precision hybrid 2>&1 | grep -E "gemm_split\["
done
