#!/usr/bin/env python
"""Fused feed-forward launch (csrc/ffn2.hip) against the two GEMM launches it replaces, forward and backward, at the step's shapes
(M = 16384; d = 256: spat encoder, d = 512: spec encoder).  Event-timed loops of 20 on rotating buffer sets (cold operands)."""
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
    hip.gpu_runway(4.0)               # the host needs ~15 us per Python call: keep the queue fed so the events bracket GPU time only
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    M = int(os.environ.get("M", "16384"))
    NSET = 6
    for dtp in (torch.float16, torch.bfloat16):
        for d in (256, 512):
            H = 4 * d
            g = torch.Generator().manual_seed(1)
            ln = [torch.randn((M, d), generator=g).to(dtp).to(dev) for _ in range(NSET)]
            x = [torch.randn((M, d), generator=g).to(dtp).to(dev) for _ in range(NSET)]
            W1 = (torch.randn((H, d), generator=g) * d ** -0.5).to(dtp).to(dev)
            W2 = (torch.randn((d, H), generator=g) * H ** -0.5).to(dtp).to(dev)
            b1, b2 = torch.zeros(H, device=dev), torch.zeros(d, device=dev)
            gdt = torch.bfloat16
            w1p, w2p = torch.empty(H * d, dtype=dtp, device=dev), torch.empty(H * d, dtype=dtp, device=dev)
            w2tp, w1tp = torch.empty(H * d, dtype=gdt, device=dev), torch.empty(H * d, dtype=gdt, device=dev)
            W2g, W1g = W2.to(gdt), W1.to(gdt)
            t_pack = timed(lambda i: hip.ffn_pack([(W1, w1p), (W2, w2p), (W2g.t(), w2tp), (W1g.t(), w1tp)]))
            hpre = [torch.empty((M, H), dtype=dtp, device=dev) for _ in range(NSET)]
            dz2 = [(torch.randn((M, d), generator=g) * 1e-3).to(gdt).to(dev) for _ in range(NSET)]
            for p in (0.0, 0.1):
                def fused(i):
                    hip.ffn2_fwd(ln[i % NSET], w1p, w2p, b1, b2, x[i % NSET], d, p1=p, s1=11, p2=p, s2=12, out_scale=0.5)

                def pair(i):
                    a = hip.gemm(ln[i % NSET], W1, M=M, N=H, K=d, lda=d, ldb=d, bias=b1, act=2, preact=hpre[i % NSET], p_drop=p, seed=11)
                    hip.gemm(a, W2, M=M, N=d, K=H, lda=H, ldb=H, bias=b2, p_drop=p, seed=12, out_scale=0.5, resid=x[i % NSET], ldr=d, res_scale=1.0)

                def fused_b(i):
                    hip.ffn2_bwd(dz2[i % NSET], w2tp, w1tp, hpre[i % NSET], d, p1=p, s1=11)

                def pair_b(i):
                    dh = hip.gemm(dz2[i % NSET], W2g, a_kc=True, b_kc=False, M=M, N=H, K=d, lda=d, ldb=H, aux=hpre[i % NSET], aux_act=2, p_drop=p, seed=11)
                    hip.gemm(dh, W1g, a_kc=True, b_kc=False, M=M, N=d, K=H, lda=H, ldb=d)
                pair(0)
                tf, tp, tfb, tpb = timed(fused), timed(pair), timed(fused_b), timed(pair_b)
                flop = 4.0 * M * d * H
                print("%s d=%d p=%.1f  fwd fused %.1f us (%.0f TF/s) | two launches %.1f us   bwd fused %.1f us (%.0f TF/s) | two launches %.1f us   pack(4) %.1f us"
                      % (str(dtp)[6:], d, p, tf, flop / tf * 1e-6, tp, tfb, flop / tfb * 1e-6, tpb, t_pack), flush=True)


if __name__ == "__main__":
    main()
