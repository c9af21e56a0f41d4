"""NT GEMM sweep at the step's shapes (M = 16384): the direct-to-LDS pipelined kernel (tools/gemm_nt/gemm_nt.hip, a round-3 experiment
that is NOT in the product library) in each of its tile configurations against the product's register-staged kernel (csrc/gemm.hip) and the
library (torch.matmul -> hipBLASLt), plain and with the step's epilogues; every configuration is first checked against an f64 product of
the same bf16 operands.  Each variant runs in a child process (the configuration switches are read once per process).

    python tools/gemm_nt/gemm_nt_sweep.py --build-only     # (here: compiles tmp_ab/libgemmnt.so, which travels to the GPU box)
    python tools/gemm_nt/gemm_nt_sweep.py                  # all shapes
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PROBE = os.path.join(ROOT, "tmp_ab", "libgemmnt.so")
SHAPES = [("ffn1 d512", 2048, 512), ("ffn2 d512", 512, 2048), ("qkv d512", 1536, 512), ("pw1 d512", 1024, 512), ("out d512", 512, 512),
          ("dec1", 3072, 768), ("dec1 dX", 768, 3072), ("dec2", 1024, 3072), ("dec2 dX", 3072, 1024), ("patch d512", 512, 1024),
          ("ffn1 d256", 1024, 256), ("ffn2 d256", 256, 1024), ("proj d256", 256, 256), ("qkv d256", 768, 256), ("pw1 d256", 512, 256),
          ("patch d256", 256, 1024), ("square 8k", 8192, 8192)]


def child():
    os.environ["SARSSL_HIP_LIB"] = PROBE
    sys.path.insert(0, ROOT)
    import sarssl_boot  # noqa
    import torch
    from sar_ssl_amd import hip
    dev = torch.device("cuda:0")
    mode = os.environ.get("SWEEP_MODE", "plain")
    res = {}

    def t(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    for name, N, K in SHAPES:
        M = 8192 if name.startswith("square") else 16384
        g = torch.Generator(device="cuda").manual_seed(N * 7 + K)
        A = torch.randn((M, K), generator=g, device=dev).bfloat16()
        W = (torch.randn((N, K), generator=g, device=dev) * 0.05).bfloat16()
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        if os.environ.get("SWEEP_LIB") == "1":
            res[name] = t(lambda: torch.matmul(A, W.t(), out=out))
            continue
        kw = {}
        if mode == "epi":                                   # the FFN-1 style epilogue: bias + swish + pre-activation side output + dropout
            kw = dict(bias=torch.randn(N, generator=g, device=dev), act=2, preact=torch.empty_like(out), p_drop=0.1, seed=5)
        elif mode == "resid":                               # the FFN-2 / projection style epilogue: bias + dropout + scaled residual
            kw = dict(bias=torch.randn(N, generator=g, device=dev), p_drop=0.1, seed=5, out_scale=0.5,
                      resid=torch.randn((M, N), generator=g, device=dev).bfloat16(), ldr=N, res_scale=1.0)
        fn = lambda: hip.gemm(A, W, M=M, N=N, K=K, lda=K, ldb=K, out=out, **kw)
        fn()
        if mode == "plain":                                 # correctness on sampled rows (full f64 product of 64 rows)
            rows = torch.arange(0, M, M // 64, device=dev)
            ref = A[rows].double() @ W.double().t()
            err = ((out[rows].double() - ref).abs().max() / ref.abs().max()).item()
            assert err < 1e-2, (name, err)
        res[name] = t(fn)
    print("SWEEP " + json.dumps(res), flush=True)


def run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=env)
    for line in r.stdout.splitlines():
        if line.startswith("SWEEP "):
            return json.loads(line[6:])
    sys.stderr.write("variant %r failed:\n%s\n%s\n" % (env_extra, r.stdout[-1500:], r.stderr[-2500:]))
    return {}


def main():
    if "--build-only" in sys.argv or not os.path.exists(PROBE):
        sys.path.insert(0, HERE)
        import build_probe
        build_probe.build(PROBE)
        if "--build-only" in sys.argv:
            return
    variants = [("lib", dict(SWEEP_LIB="1")), ("old", dict(SARSSL_GEMM_NT="0")), ("auto", {})]
    variants += [("cfg%d" % c, dict(SARSSL_GEMM_NT_CFG=str(c))) for c in (0, 2, 3, 6)]
    modes = os.environ.get("SWEEP_MODES", "plain,epi,resid").split(",")
    for mode in modes:
        cols = {}
        for vname, env in variants:
            if mode != "plain" and vname == "lib":
                continue
            cols[vname] = run(dict(env, SWEEP_MODE=mode))
        names = list(cols)
        print("== mode %s: us per launch (TFLOP/s)   [cfg0 256x256 8w BK32 4st | cfg2 256x128 8w BK64 3st | cfg3 128x128 4w BK32 4st | cfg6 256x256 4w (one wave per SIMD) BK32 4st]" % mode)
        print("%-12s %5s %5s  " % ("shape", "N", "K") + "  ".join("%14s" % n for n in names))
        for name, N, K in SHAPES:
            M = 8192 if name.startswith("square") else 16384
            fl = 2.0 * M * N * K
            cells = []
            for n in names:
                us = cols[n].get(name)
                cells.append("%7.1f (%4.0f)" % (us, fl / us / 1e6) if us else "      -       ")
            print("%-12s %5d %5d  " % (name, N, K) + "  ".join("%14s" % c for c in cells), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
