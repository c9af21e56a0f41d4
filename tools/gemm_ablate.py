import os, subprocess, sys
for dbg in (0, 1, 2, 4, 3, 6, 7):
    env = dict(os.environ, SARSSL_GEMM_DBG=str(dbg))
    r = subprocess.run([sys.executable, "tools/bench_kernels.py"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ffn1 NT d=256", "ffn1 NT d=512", "ffn2 NT d=512", "decoder1"))]
    print("dbg=%d (stage-once=%d nostore=%d nomfma=%d)" % (dbg, dbg & 1, (dbg >> 1) & 1, (dbg >> 2) & 1))
    for l in lines: print("   ", l)
