import os, subprocess, sys
for dbg in (0, 1, 4, 5):
    env = dict(os.environ, SARSSL_CONV_DBG=str(dbg))
    r = subprocess.run([sys.executable, "tools/bench_kernels.py"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if "conv3x3_wgrad" in l]
    print("dbg=%d (stage-once=%d nomfma=%d): %s" % (dbg, dbg & 1, (dbg >> 2) & 1, line[0] if line else r.stderr[-300:]))
