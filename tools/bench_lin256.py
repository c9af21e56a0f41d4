#!/usr/bin/env python
"""Tile-resident Linear launches (csrc/lin256.hip) against the sarssl_gemm (+ LayerNorm) launches they replace, at the spat encoder's
shapes (M = 16384, d = 256): event-timed behind a GPU runway, rotating buffer sets."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa: E402,F401
from sar_ssl_amd import hip  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hip.gpu_runway(4.0)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    M, NS = 16384, 6
    dtp, gdt = torch.float16, torch.bfloat16
    g = torch.Generator().manual_seed(1)
    gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    for N, K, what in ((768, 256, "q/k/v"), (512, 256, "pointwise conv 1"), (256, 256, "out proj / pointwise conv 2")):
        x = [torch.randn((M, K), generator=g).to(dtp).to(dev) for _ in range(NS)]
        W = (torch.randn((N, K), generator=g) * K ** -0.5).to(dtp).to(dev)
        bias = torch.zeros(N, device=dev)
        wp = torch.empty(N * K, dtype=dtp, device=dev)
        Wg = W.to(gdt)
        wtp = torch.empty(N * K, dtype=gdt, device=dev)
        hip.ffn_pack([(W, wp), (Wg.t(), wtp)])
        res = [torch.randn((M, N), generator=g).to(dtp).to(dev) for _ in range(NS)] if N == 256 else None
        dy = [(torch.randn((M, N), generator=g) * 1e-3).to(gdt).to(dev) for _ in range(NS)]
        rs = [(torch.randn((M, K), generator=g) * 1e-3).to(gdt).to(dev) for _ in range(NS)]
        _, st = hip.layernorm_fwd(x[0], gamma, beta)
        dg, db = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
        if N == 256:
            t_f = timed(lambda i: hip.lin256_fwd(x[i % NS], wp, bias, N, K, resid=res[i % NS], p_drop=0.1, seed=3))
            t_g = timed(lambda i: hip.gemm(x[i % NS], W, M=M, N=N, K=K, lda=K, ldb=K, bias=bias, p_drop=0.1, seed=3, resid=res[i % NS], ldr=N, res_scale=1.0))
            t_b = timed(lambda i: hip.lin256_bwd(dy[i % NS], wtp, K, N))
            t_gb = timed(lambda i: hip.gemm(dy[i % NS], Wg, a_kc=True, b_kc=False, M=M, N=K, K=N, lda=N, ldb=K))
            print("%-28s N=%d K=%d  fwd(+drop,resid) fused %.1f us | gemm %.1f us   dX fused %.1f us | gemm %.1f us" % (what, N, K, t_f, t_g, t_b, t_gb), flush=True)
        else:
            t_f = timed(lambda i: hip.lin256_fwd(None, wp, bias, N, K, ln_in=(x[i % NS], gamma, beta, 1e-5)))

            def two(i):
                ln, _ = hip.layernorm_fwd(x[i % NS], gamma, beta)
                hip.gemm(ln, W, M=M, N=N, K=K, lda=K, ldb=K, bias=bias)
            t_g = timed(two)
            t_b = timed(lambda i: hip.lin256_bwd(dy[i % NS], wtp, K, N, ln_bwd=(x[i % NS], gamma, st, rs[i % NS], dg, db, (0.1, 5, 1.0))))

            def two_b(i):
                dln = hip.gemm(dy[i % NS], Wg, a_kc=True, b_kc=False, M=M, N=K, K=N, lda=N, ldb=K)
                hip.layernorm_bwd(dln, x[i % NS], gamma, st, resid=rs[i % NS], dgamma=dg, dbeta=db, drop=(0.1, 5, 1.0))
            t_gb = timed(two_b)
            print("%-28s N=%d K=%d  LN+fwd fused %.1f us | LN + gemm %.1f us   dX+LN' fused %.1f us | gemm + LN' (+reduce) %.1f us" % (what, N, K, t_f, t_g, t_b, t_gb), flush=True)


if __name__ == "__main__":
    main()
