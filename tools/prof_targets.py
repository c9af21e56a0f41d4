"""Runs every hot kernel family of the pretraining step at BASELINE config-2 shapes (B = 64), each group preceded by a marker
launch (tools/prof_marker.hip, grid = tag) - the target of tools/prof_counters.py's rocprofv3 --pmc passes.
PROF_TIME=1: no markers, each group timed with events (median / min over >= 15 calls; PROF_ONLY=30,31 selects groups)."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip

MARK = os.path.join(HERE, "libprofmarker.so")
TIME = os.environ.get("PROF_TIME", "0") != "0"
# The marker library is built BEFORE the profiler starts this program (tools/prof_counters.py build_marker(), or
# `python tools/prof_counters.py --build-marker`): under rocprofv3 the profiler's preloaded library has already initialised the GPU
# when this module runs, and hipcc's clang / lld children would be exec'd from such a process - forbidden on this pool.
if not TIME:
    if not os.path.exists(MARK) or os.path.getmtime(MARK) < os.path.getmtime(os.path.join(HERE, "prof_marker.hip")):
        raise SystemExit("tools/libprofmarker.so is missing or older than prof_marker.hip: run `python tools/prof_counters.py "
                         "--build-marker` first (it is never compiled from inside the profiled process)")
    _mk = ctypes.CDLL(MARK)
dev = torch.device("cuda:0")
NREP = int(os.environ.get("PROF_NREP", "3"))
# dtype of the FORWARD tensors (activations, forward weights, tensors saved for backward): fp16 = the timed mode since round 4
# (fp16 forward / bf16 backward), PROF_PRECISION=bf16 for the all-bf16 mode; gradients are bf16 in both
AD = torch.bfloat16 if os.environ.get("PROF_PRECISION", "fp16") == "bf16" else torch.float16
GD = torch.bfloat16
ONLY = set(int(t) for t in os.environ.get("PROF_ONLY", "").split(",") if t)

TAGS = {}          # tag -> (label, algorithmic flop per launch, algorithmic bytes per launch)


def group(tag, label, fn, flop=0.0, nbytes=0.0):
    TAGS[tag] = (label, flop, nbytes)
    if ONLY and tag not in ONLY:
        return
    fn()                                             # warm (allocations)
    torch.cuda.synchronize()
    if TIME:                                         # PROF_TIME=1: event-timed median instead of the marker protocol
        ts = []
        for _ in range(max(NREP, 15)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        print("%3d %-62s %8.1f us  min %8.1f%s%s" % (tag, label, us, ts[0], "  %6.0f TFLOP/s" % (flop / us / 1e6) if flop else "",
                                                    "  %5.2f TB/s" % (nbytes / us / 1e6) if nbytes else ""), flush=True)
        return
    _mk.prof_marker(ctypes.c_int(tag), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(NREP):
        fn()
    _mk.prof_marker(ctypes.c_int(0x3fff), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))      # closes the group
    torch.cuda.synchronize()


def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(shape, device=dev) * scale).to(dtype)


def gemm_group(tag, label, M, N, K, a_kc=True, b_kc=True, out_dtype=None, split=0, adt=None, bdt=None, **kw):
    """adt / bdt: operand dtypes (default: forward tensors)"""
    adt, bdt = adt or AD, bdt or AD
    out_dtype = out_dtype or adt
    A = rnd(*((M, K) if a_kc else (K, M)), dtype=adt)
    B = rnd(*((N, K) if b_kc else (K, N)), dtype=bdt)
    out = torch.zeros((M, N), dtype=out_dtype, device=dev)
    osz = 4 if out_dtype == torch.float32 else 2
    group(tag, label, lambda: hip.gemm(A, B, a_kc=a_kc, b_kc=b_kc, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], out=out,
                                       split_k=split, **kw), 2.0 * M * N * K, 2.0 * (M * K + N * K) + osz * M * N)


def main():
    Mr = 64 * 256
    bias = torch.zeros(4096, device=dev)
    d = 512
    pre = torch.empty((Mr, 4 * d), dtype=AD, device=dev)
    gemm_group(1, "gemm ffn1 NT d=512 (+bias,swish,preact,dropout)", Mr, 4 * d, d, bias=bias[:4 * d], act=2, preact=pre, p_drop=0.1, seed=7)
    res = rnd(Mr, d, dtype=AD)
    gemm_group(2, "gemm ffn2 NT d=512 (+bias,dropout,resid)", Mr, d, 4 * d, bias=bias[:d], p_drop=0.1, seed=9, resid=res, ldr=d, out_scale=0.5)
    gemm_group(3, "gemm qkv NT d=512 (N=1536)", Mr, 3 * d, d, bias=bias[:3 * d])
    aux = rnd(Mr, 4 * d, dtype=AD)                      # saved pre-activation (forward dtype) next to bf16 gradients
    gemm_group(4, "gemm ffn2 dX NN d=512 (+swish' aux)", Mr, 4 * d, d, b_kc=False, aux=aux, aux_act=2, adt=GD, bdt=GD)
    gemm_group(5, "gemm ffn1 dX NN d=512", Mr, d, 4 * d, b_kc=False, adt=GD, bdt=GD)
    d = 512
    gemm_group(6, "gemm ffn dW TN d=512 (split-K; dY bf16 x saved X)", 4 * d, d, Mr, a_kc=False, b_kc=False, out_dtype=torch.float32, split=8, adt=GD)
    gemm_group(7, "gemm decoder1 NT (768->3072, relu)", Mr, 3072, 768, bias=bias[:3072], act=1)
    gemm_group(8, "gemm decoder2 NT (3072->1024)", Mr, 1024, 3072, bias=bias[:1024])
    gemm_group(9, "gemm decoder2 dW TN (split-K 8, slice-major)", 1024, 3072, Mr, a_kc=False, b_kc=False, out_dtype=torch.float32, split=8, adt=GD)
    gemm_group(10, "gemm patch NT (1024->512)", Mr, 512, 1024)
    gemm_group(11, "gemm ffn1 NT d=256", Mr, 1024, 256, bias=bias[:1024], act=2)
    gemm_group(12, "gemm ffn2 NT d=256", Mr, 256, 1024, bias=bias[:256])
    # ---- fused feed-forward launches (csrc/ffn2.hip): forward and backward, d = 256 (the default) and d = 512
    for tag, dd in ((13, 256), (15, 512)):
        Hh = 4 * dd
        W1 = rnd(Hh, dd, dtype=AD, scale=dd ** -0.5); W2 = rnd(dd, Hh, dtype=AD, scale=Hh ** -0.5)
        packs = [torch.empty(Hh * dd, dtype=AD, device=dev), torch.empty(Hh * dd, dtype=AD, device=dev),
                 torch.empty(Hh * dd, dtype=GD, device=dev), torch.empty(Hh * dd, dtype=GD, device=dev)]
        hip.ffn_pack([(W1, packs[0]), (W2, packs[1]), (W2.to(GD).t(), packs[2]), (W1.to(GD).t(), packs[3])])
        lnx, resx = rnd(Mr, dd, dtype=AD), rnd(Mr, dd, dtype=AD)
        hp = rnd(Mr, Hh, dtype=AD)
        dz = rnd(Mr, dd, dtype=GD, scale=1e-3)
        flop = 4.0 * Mr * dd * Hh
        group(tag, "ffn2 fused forward d=%d (LN out -> y; pre-activation + hidden saved; dropout 0.1)" % dd,
              lambda: hip.ffn2_fwd(lnx, packs[0], packs[1], bias[:Hh], bias[:dd], resx, dd, p1=0.1, s1=7, p2=0.1, s2=9, out_scale=0.5),
              flop, 2.0 * (3 * Mr * dd + 2 * Mr * Hh))
        group(tag + 1, "ffn2 fused backward d=%d (dz2 -> dh saved, dln)" % dd,
              lambda: hip.ffn2_bwd(dz, packs[2], packs[3], hp, dd, p1=0.1, s1=7), flop, 2.0 * (2 * Mr * dd + 2 * Mr * Hh))
    # ---- stem
    B = 64
    x = rnd(B, 256, 256, 64, dtype=AD)              # a forward activation
    y = rnd(B, 256, 256, 64, dtype=AD)              # a saved pre-BatchNorm activation
    g64 = rnd(B, 256, 256, 64, dtype=GD)            # a gradient
    w = rnd(9, 64, 64, scale=0.05, dtype=AD)
    wg = rnd(9, 64, 64, scale=0.05, dtype=GD)
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    aff = torch.stack([sc, sh, sh, sc]).contiguous()
    cfl = 2.0 * B * 65536 * 64 * 576
    tb = x.numel() * 2.0
    group(20, "conv3x3 fwd (BN+ReLU prologue, stats epilogue)", lambda: hip.conv3x3_fwd(x, w, sc, sh, want_stats=True), cfl, 2 * tb)
    group(21, "conv3x3 dgrad (identity prologue)", lambda: hip.conv3x3_fwd(g64, wg), cfl, 2 * tb)
    group(22, "conv3x3 dgrad + BN-backward sums", lambda: hip.conv3x3_dgrad_bnred(g64, wg, y, aff), cfl, 3 * tb)
    group(23, "conv3x3 wgrad", lambda: hip.conv3x3_wgrad(g64, y, sc, sh), cfl, 2 * tb)
    a0 = rnd(B, 256, 256, 4, dtype=AD)
    W1 = torch.randn((64, 4), device=dev)
    W4 = torch.randn((4, 64), device=dev)
    group(30, "stem_c1_fwd (4->64 + BN sums)", lambda: hip.stem_c1_fwd(a0, W1, want_stats=True), 0, tb + tb / 16)
    group(36, "stem_c1_fwd (4->64, no sums)", lambda: hip.stem_c1_fwd(a0, W1), 0, tb + tb / 16)
    group(31, "stem_c4_fwd (BN+ReLU, 64->4)", lambda: hip.stem_c4_fwd(x, W4, sc, sh), 0, tb + tb / 16)
    dy4 = rnd(B, 256, 256, 4, dtype=GD)
    group(32, "stem_c4_bwd two-phase (sums + apply)", lambda: hip.stem_c4_bwd_two_phase(x, dy4, W4, aff, True), 0, 3 * tb + 2 * tb / 16)
    gW, gg, gb = torch.zeros((64, 4), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    group(33, "stem_c1_bwd (one pass)", lambda: hip.stem_c1_bwd(g64, y, a0, aff, True, gW, gg, gb), 0, 2 * tb + tb / 16)
    group(37, "stem_c1_bwd_a0 (one pass, y1 recomputed from the input)", lambda: hip.stem_c1_bwd_a0(g64, a0, W1, aff, True, gW, gg, gb), 0, tb + tb / 16)
    wt = rnd(9, 64, 64, scale=0.05, dtype=AD)
    group(24, "conv3x3 fwd from the 4-channel input (C1IN, + stats)", lambda: hip.conv3x3_fwd_c1(a0, W1, sc, sh, wt, want_stats=True), cfl, tb + tb / 16)
    gacc = torch.zeros((64, 64, 3, 3), device=dev)
    group(25, "conv3x3 wgrad from the 4-channel input (C1IN)", lambda: hip.conv3x3_wgrad_c1(g64, a0, W1, sc, sh, gacc), cfl, tb + tb / 16)
    _, mom = hip.stem_c1_stats(a0, W1, keep_moments=True)
    group(26, "conv3x3 dgrad consumed in its epilogue (C1RED: mask + [a0|1] contraction on MFMA, nothing stored)",
          lambda: hip.conv3x3_dgrad_c1red(g64, wg, a0, W1, aff, mom, True, gW, gg, gb), cfl, tb + tb / 16)
    red = torch.zeros(128, dtype=torch.float64, device=dev)
    group(34, "cl_bn_bwd_apply C=64 (in place)", lambda: hip.cl_bn_bwd_apply(g64, y, 64, aff, 1, False, True, red, out=g64), 0, 3 * tb)
    group(35, "cl_bn_bwd_reduce C=64", lambda: hip.cl_bn_bwd_reduce(g64, y, 64, aff, 1), 0, 2 * tb)
    # ---- attention glue (d = 512: 4 heads, T = 256)
    T, H = 256, 4
    content = torch.randn((B, H, T, T), device=dev)
    pos = torch.randn((B, H, T, T), device=dev)
    group(40, "softmax_relshift_fwd (B,4,256,256) [unfused core, fp32 mode only]", lambda: hip.softmax_relshift_fwd(content, pos, 0.044, AD, 0.1, 5), 0,
          content.numel() * (8.0 + 4.0))
    del content, pos
    # ---- fused attention (csrc/attention.hip)
    for tag, dh in ((41, 128), (44, 64)):
        d = H * dh
        qkv = rnd(B * T, 3 * d, dtype=AD)
        qu, dctx = rnd(B * T, d, dtype=AD), rnd(B * T, d, dtype=GD)
        aflop = 2.0 * 2 * B * H * T * T * dh                        # QK^T + PV
        abytes = 2.0 * (4 * B * T * d + B * H * T * T)
        k_, v_ = qkv[:, d:2 * d], qkv[:, 2 * d:]
        # the product path for T <= 256: positional score formed in the forward kernel, its gradients' products in the dQ kernel
        qv, posp = rnd(B * T, d, dtype=AD), rnd(T, d, dtype=AD)
        pflop = 2.0 * B * H * T * T * dh                            # (q + v) P^T
        group(tag, "relpos_attn_fwd_pos dh=%d (B=64,H=4,T=256; score in-kernel, slab saved)" % dh,
              lambda: hip.relpos_attn_fwd_pos(qu, qv, k_, v_, posp, B, H, T, dh, 0.044, 0.1, 5), aflop + pflop,
              2.0 * (5 * B * T * d + B * H * T * T) + 4.0 * B * T * d)
        ctx, aux, bias = hip.relpos_attn_fwd_pos(qu, qv, k_, v_, posp, B, H, T, dh, 0.044, 0.1, 5)
        dqkv = torch.empty(qkv.shape, dtype=GD, device=dev)
        dqv = torch.empty((B * T, d), dtype=GD, device=dev)
        group(tag + 1, "relpos_attn_bwd_pos dh=%d (dQ + positional gradients (+ D) + dK/dV kernels)" % dh,
              lambda: hip.relpos_attn_bwd_pos(qu, qv, k_, v_, posp, bias, aux, dctx, dqkv[:, :d], dqv, dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                                              B, H, T, dh, 0.044, 0.1, 5),
              3.5 * aflop + 2 * pflop, 2.0 * (10 * B * T * d + 2 * B * H * T * T + 2 * T * d * B))
    # ---- convolution-module tiles (csrc/dwconv.hip)
    d = 512
    hh, dcc = rnd(B * T, 2 * d, dtype=AD), rnd(B * T, d, dtype=GD)
    wdw = torch.randn((d, 31), device=dev)
    gdw = torch.zeros((d, 31), device=dev)
    eb = 2.0 * B * T * d
    group(50, "dwglu_fwd d=512 (GLU + depthwise conv + BN sums)", lambda: hip.dwglu_fwd(hh, wdw, B, T, want_stats=True), 0, 3 * eb)
    group(51, "dwglu_bwd d=512 (data gradient + GLU backward)", lambda: hip.dwglu_bwd(dcc, hh, wdw, B, T), 0, 5 * eb)
    group(52, "dwglu_wgrad d=512", lambda: hip.dwglu_wgrad(dcc, hh, gdw, B, T), 0, 3 * eb)
    # ---- LayerNorm (csrc/elementwise.hip)
    for tag, d in ((70, 256), (72, 512)):
        xx, dyy, rr = rnd(Mr, d, dtype=AD), rnd(Mr, d, dtype=GD), rnd(Mr, d, dtype=GD)
        gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        gg, gb = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        _, st = hip.layernorm_fwd(xx, gam, bet)
        group(tag, "layernorm_fwd d=%d (M=16384)" % d, lambda: hip.layernorm_fwd(xx, gam, bet), 0, 2.0 * 2 * Mr * d)
        group(tag + 1, "layernorm_bwd d=%d (+resid, dgamma/dbeta)" % d, lambda: hip.layernorm_bwd(dyy, xx, gam, st, resid=rr, dgamma=gg, dbeta=gb),
              0, 2.0 * 4 * Mr * d)
        group(tag + 4, "layernorm_bwd d=%d (+resid, dgamma/dbeta, dropout output)" % d,
              lambda: hip.layernorm_bwd(dyy, xx, gam, st, resid=rr, dgamma=gg, dbeta=gb, drop=(0.1, 3, 0.5)), 0, 2.0 * 5 * Mr * d)
    # ---- hybrid numeric mode (round 6): fp16-pair products, LayerNorm -> pair, LayerNorm backward on the f32 stream, fused feed-forward
    if AD == torch.float16:
        f32 = torch.float32
        for tag, (M_, N_, K_, nseg, odt, lbl) in ((80, (Mr, 1024, 256, 3, AD, "ffn1 d=256 (3 products; +bias,swish,preact,dropout -> fp16)")),
                                                  (81, (Mr, 256, 1024, 2, f32, "ffn2 d=256 (2 products; +bias,dropout,f32 resid -> f32)")),
                                                  (82, (Mr, 2048, 512, 3, AD, "ffn1 d=512 (3 products -> fp16)")),
                                                  (83, (Mr, 512, 2048, 2, f32, "ffn2 d=512 (2 products, f32 resid -> f32)")),
                                                  (84, (Mr // 2, 3072, 768, 3, AD, "decoder layer 1 on the masked rows (3 products, relu -> fp16)")),
                                                  (85, (Mr // 2, 1024, 3072, 2, f32, "decoder layer 2 on the masked rows (2 products -> f32)"))):
            xa = hip.split_pair(torch.randn((M_, K_), device=dev))
            wp = hip.split_pair(torch.randn((N_, K_), device=dev) * K_ ** -0.5)
            A_ = xa if nseg == 3 else xa.hi
            o_ = torch.empty((M_, N_), dtype=odt, device=dev)
            kw = dict(bias=bias[:N_])
            if odt == f32:
                kw.update(resid=torch.randn((M_, N_), device=dev), ldr=N_, p_drop=0.1, seed=9, out_scale=0.5)
            elif tag in (80, 82):
                kw.update(act=2, preact=torch.empty((M_, N_), dtype=AD, device=dev), p_drop=0.1, seed=7)
            else:
                kw.update(act=1)
            osz = 4 if odt == f32 else 2
            group(tag, "gemm_split " + lbl, lambda: hip.gemm_split(A_, wp.hi, wp.lo, M=M_, N=N_, K=K_, out=o_, **kw),
                  2.0 * M_ * N_ * K_ * nseg, 2.0 * ((nseg - 1) * M_ * K_ + 2 * N_ * K_) + osz * M_ * N_ * (2 if (odt == f32 or tag in (80, 82)) else 1))
        dd, Hh = 256, 1024
        W1 = hip.split_pair(torch.randn((Hh, dd), device=dev) * dd ** -0.5); W2 = hip.split_pair(torch.randn((dd, Hh), device=dev) * Hh ** -0.5)
        pk = [torch.empty(Hh * dd, dtype=AD, device=dev) for _ in range(4)]
        hip.ffn_pack([(W1.hi, pk[0]), (W1.lo, pk[1]), (W2.hi, pk[2]), (W2.lo, pk[3])])
        xs = torch.randn((Mr, dd), device=dev)
        gam, bet = torch.ones(dd, device=dev), torch.zeros(dd, device=dev)
        group(86, "ffn2h fused forward d=256 on the f32 stream (LayerNorm + 3 / 2 products; pre-activation + hidden + LN hi saved)",
              lambda: hip.ffn2h_fwd(xs, gam, bet, 1e-5, pk[0], pk[1], pk[2], pk[3], bias[:Hh], bias[:dd], dd, p1=0.1, s1=7, p2=0.1, s2=9, out_scale=0.5),
              2.0 * Mr * dd * Hh * 5, 4.0 * 2 * Mr * dd + 2.0 * (Mr * dd + 2 * Mr * Hh))
        for tag, d in ((90, 256), (92, 512)):
            xx = torch.randn((Mr, d), device=dev)
            gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
            gg, gb = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
            _, st = hip.layernorm_fwd_pair(xx, gam, bet)
            dyb, rr = rnd(Mr, d, dtype=GD), torch.randn((Mr, d), device=dev)
            group(tag, "layernorm_fwd_pair d=%d (f32 rows -> fp16 hi + lo)" % d, lambda: hip.layernorm_fwd_pair(xx, gam, bet), 0, (4.0 + 4.0) * Mr * d)
            group(tag + 1, "layernorm_bwd_stream d=%d (bf16 branch gradient, f32 x / resid / dx, bf16 dropped copy)" % d,
                  lambda: hip.layernorm_bwd_stream(dyb, xx, gam, st, resid=rr, dgamma=gg, dbeta=gb, drop=(0.1, 3, 0.5)), 0, (2.0 + 4 + 4 + 4 + 2) * Mr * d)
    # ---- fp8 GEMM (csrc/gemm_fp8.hip)
    for tag, (M_, N_, K_) in ((60, (Mr, 2048, 512)), (61, (Mr, 1024, 3072))):
        A = rnd(M_, K_, dtype=GD); Bm = rnd(N_, K_, scale=0.05, dtype=GD)
        Aq, sa = hip.fp8_quantize(A); Bq, sb = hip.fp8_quantize(Bm)
        out = torch.empty((M_, N_), dtype=torch.bfloat16, device=dev)
        group(tag, "fp8 gemm %dx%dx%d (e4m3, block-scaled MFMA)" % (M_, N_, K_), lambda: hip.gemm_fp8(Aq, sa, Bq, sb, M=M_, N=N_, K=K_, out=out),
              2.0 * M_ * N_ * K_, 1.0 * (M_ * K_ + N_ * K_) + 2.0 * M_ * N_)
        group(tag + 2, "fp8 activation quantise %dx%d (amax + convert)" % (M_, K_), lambda: hip.fp8_quantize(A), 0, 2.0 * 2 * M_ * K_ + M_ * K_)
    import json
    print("PROF_TAGS " + json.dumps(TAGS))


if __name__ == "__main__":
    main()
