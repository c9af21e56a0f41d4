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


def hybrid(dev, M, NSET=6):
    """csrc/ffn2h.hip (hybrid mode, f32 stream, weight pairs) against LayerNorm -> pair + two sarssl_gemm_split launches, d = 256."""
    d, H = 256, 1024
    g = torch.Generator().manual_seed(1)
    x = [torch.randn((M, d), generator=g).to(dev) for _ in range(NSET)]
    y = [torch.empty((M, d), device=dev) for _ in range(NSET)]
    W1 = (torch.randn((H, d), generator=g) * d ** -0.5).to(dev)
    W2 = (torch.randn((d, H), generator=g) * H ** -0.5).to(dev)
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    b1, b2 = torch.zeros(H, device=dev), torch.zeros(d, device=dev)
    w1, w2 = hip.split_pair(W1), hip.split_pair(W2)
    pk = [torch.empty(H * d, dtype=torch.float16, device=dev) for _ in range(4)]
    hip.ffn_pack([(w1.hi, pk[0]), (w1.lo, pk[1]), (w2.hi, pk[2]), (w2.lo, pk[3])])
    for ap in (True, False):
        for p in (0.0, 0.1):
            def fused(i):
                hip.ffn2h_fwd(x[i % NSET], gam, bet, 1e-5, pk[0], pk[1], pk[2], pk[3], b1, b2, d, p1=p, s1=11, p2=p, s2=12, out_scale=0.5,
                              out=y[i % NSET], act_pair=ap)

            def unfused(i):
                ln, _ = hip.layernorm_fwd_pair(x[i % NSET], gam, bet, 1e-5, want_lo=ap)
                pre = torch.empty((M, H), dtype=torch.float16, device=dev)
                a = hip.gemm_split(ln, w1.hi, w1.lo, M=M, N=H, K=d, out_dtype=torch.float16, bias=b1, act=2, preact=pre, p_drop=p, seed=11)
                hip.gemm_split(a, w2.hi, w2.lo, M=M, N=d, K=H, out=y[i % NSET], bias=b2, p_drop=p, seed=12, out_scale=0.5, resid=x[i % NSET], ldr=d,
                               res_scale=1.0)
            tf, tu = timed(fused), timed(unfused)
            issued = 2.0 * M * d * H * ((3 if ap else 2) + 2)
            print("hybrid d=256 act_pair=%d p=%.1f  fwd fused %.1f us (%.0f TF/s issued) | LayerNorm + two launches %.1f us   [SARSSL_FFN_ROT=%s]"
                  % (ap, p, tf, issued / tf * 1e-6, tu, os.environ.get("SARSSL_FFN_ROT", "0")), flush=True)


def hybrid_stamps(dev, M):
    """Phase boundaries (s_memtime, shader clock cycles) of workgroup 0 of one ffn2h launch, per wave."""
    import ctypes
    d, H = 256, 1024
    g = torch.Generator().manual_seed(1)
    x = torch.randn((M, d), generator=g).to(dev)
    W1 = (torch.randn((H, d), generator=g) * d ** -0.5).to(dev)
    W2 = (torch.randn((d, H), generator=g) * H ** -0.5).to(dev)
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    b1, b2 = torch.zeros(H, device=dev), torch.zeros(d, device=dev)
    w1, w2 = hip.split_pair(W1), hip.split_pair(W2)
    pk = [torch.empty(H * d, dtype=torch.float16, device=dev) for _ in range(4)]
    hip.ffn_pack([(w1.hi, pk[0]), (w1.lo, pk[1]), (w2.hi, pk[2]), (w2.lo, pk[3])])
    buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
    names = ["start", "LN done", "barrier"] + sum([["P1.%d" % c, "E1.%d" % c, "bar.%d" % c, "stores.%d" % c, "P2.%d" % c, "bar'.%d" % c] for c in range(4)], []) + ["sY", "bar", "end"]
    for ap in (True, False):
        for _ in range(3):
            hip.ffn2h_fwd(x, gam, bet, 1e-5, pk[0], pk[1], pk[2], pk[3], b1, b2, d, p1=0.1, s1=11, p2=0.1, s2=12, out_scale=0.5, act_pair=ap)
        hip._lib.call("sarssl_ffn_stamp_buffer", ctypes.c_void_p(buf.data_ptr()))
        hip.ffn2h_fwd(x, gam, bet, 1e-5, pk[0], pk[1], pk[2], pk[3], b1, b2, d, p1=0.1, s1=11, p2=0.1, s2=12, out_scale=0.5, act_pair=ap)
        hip._lib.call("sarssl_ffn_stamp_buffer", ctypes.c_void_p(0))
        torch.cuda.synchronize()
        t = buf.view(8, 64).cpu()
        print("act_pair=%d: cycles since the workgroup's first stamp, waves 0 / 3 / 7 (delta of wave 0)" % ap)
        t0 = int(t[:, 0].min())
        prev = 0
        for i, n in enumerate(names):
            a = [int(t[wv, i]) - t0 for wv in (0, 3, 7)]
            print("  %-10s %8d %8d %8d   (+%d)" % (n, a[0], a[1], a[2], a[0] - prev))
            prev = a[0]


def main():
    dev = torch.device("cuda:0")
    M = int(os.environ.get("M", "16384"))
    NSET = 6
    if "--hybrid" in sys.argv:
        return hybrid(dev, M)
    if "--hybrid-stamps" in sys.argv:
        return hybrid_stamps(dev, M)
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
