for k in 1024 4096 0; do
  echo "== SARSSL_SPLIT_FM1_KMAX=$k"
  SARSSL_SPLIT_FM1_KMAX=$k python tools/step_gemm_table.py --precision hybrid 2>&1 | grep -E "gemm_split\[(8192,1024,3072|16384,256,1024|16384,256,256|16384,512,512|16384,512,2048)"
done
