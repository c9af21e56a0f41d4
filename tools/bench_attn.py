"""Isolated timings of the fused relative-position attention kernels (csrc/attention.hip) at the shapes of the timed step:
B = 64 segments, H = 4, T = 256, d_head = 128 (spectral encoder, 1 layer) and 64 (spatial encoder, 3 layers).
Forward and backward are timed as whole C-ABI calls (backward = dsum + bwd_q + bwd_kv launches); per-kernel splits come from
`rocprofv3 --kernel-trace --stats -- python3 tools/bench_attn.py`.

    python tools/bench_attn.py [--iters 50] [--drop 0.1]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sar_ssl_amd import hip  # noqa: E402


def run(B, H, T, dh, p_drop, iters):
    dev = torch.device("cuda:0")
    d = H * dh
    g = torch.Generator(device=dev).manual_seed(3)
    qkv = (torch.randn((B * T, 3 * d), device=dev, generator=g) * 0.5).to(torch.bfloat16)
    qu, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    bias = (torch.randn((B, H, T, T), device=dev, generator=g) * 0.5).to(torch.bfloat16)
    dctx = (torch.randn((B * T, d), device=dev, generator=g) * 0.1).to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    scale = 1.0 / math.sqrt(d)
    ctx, aux = hip.relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, p_drop, 11)

    def fwd():
        hip.relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, p_drop, 11)

    qv = (torch.randn((B * T, d), device=dev, generator=g) * 0.5).to(torch.bfloat16)
    pos = (torch.randn((T, d), device=dev, generator=g) * 0.5).to(torch.bfloat16)
    quc = qu.contiguous()                       # (q + u and q + v share a row stride, as the engine's bias2 produces them)

    def gemm_fwd():                             # what a layer issues with SARSSL_ATTN_POS=0: shifted positional-score GEMM, then the kernel
        hip.gemm(qv, pos, M=T, N=T, K=dh, lda=d, ldb=d, nbatch=B * H, batch_inner=H, sA=(T * d, dh), sB=(0, dh), out=bias, ldc=T,
                 sC=(H * T * T, T * T), c_row_shift=True)
        hip.relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, p_drop, 11)

    def fwd_pos():                              # the default: positional score formed in the kernel, score tile written for the backward
        hip.relpos_attn_fwd_pos(quc, qv, k, v, pos, B, H, T, dh, scale, p_drop, 11)

    def fwd_pos_inf():                          # inference: no score tile written
        hip.relpos_attn_fwd_pos(quc, qv, k, v, pos, B, H, T, dh, scale, p_drop, 11, need_bwd=False)

    ctx_p, aux_p, bias_p = hip.relpos_attn_fwd_pos(quc, qv, k, v, pos, B, H, T, dh, scale, p_drop, 11)
    dqv = torch.empty((B * T, d), dtype=torch.bfloat16, device=dev)

    def bwd_unfused_all():                      # SARSSL_ATTN_POS=0: kernels + un-shift pass + the two batched products of the positional gradients
        dbias = hip.relpos_attn_bwd(qu, k, v, bias, aux, dctx, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, T, dh, scale, p_drop, 11)
        dps = hip.relshift_bwd(dbias)
        hip.gemm(dps, pos, a_kc=True, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=B * H, batch_inner=H,
                 sA=(H * T * T, T * T), sB=(0, dh), out=dqv, ldc=d, sC=(T * d, dh))
        dposb = torch.empty((B, T, d), dtype=torch.bfloat16, device=dev)
        hip.gemm(dps, qv, a_kc=False, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=B * H, batch_inner=H,
                 sA=(H * T * T, T * T), sB=(T * d, dh), out=dposb, ldc=d, sC=(T * d, dh))
        hip.colsum_store(dposb.view(B, T * d))

    def bwd_pos():                              # the default: positional gradients formed in the dQ kernel
        dpart = hip.relpos_attn_bwd_pos(quc, qv, k, v, pos, bias_p, aux_p, dctx, dqkv[:, :d], dqv, dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                                        B, H, T, dh, scale, p_drop, 11)
        hip.colsum_store(dpart.view(dpart.shape[0], T * d))

    def bwd():
        hip.relpos_attn_bwd(qu, k, v, bias, aux, dctx, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, T, dh, scale, p_drop, 11)

    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd), ("gemm_fwd", gemm_fwd), ("fwd_pos", fwd_pos), ("fwd_pos_inf", fwd_pos_inf),
                     ("bwd_unfused_all", bwd_unfused_all), ("bwd_pos", bwd_pos)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters * 1e3
    fl_f = 2 * 2 * T * T * dh * B * H          # S and PV
    fl_b = 5 * 2 * T * T * dh * B * H * 1.0 + 2 * 2 * T * T * dh * B * H   # dQ, dK, dV, + recomputed S and dP twice
    print(f"B={B} H={H} T={T} dh={dh} drop={p_drop}:  fwd {out['fwd']:8.1f} us ({fl_f / out['fwd'] / 1e6:6.1f} TFLOP/s)   "
          f"bwd {out['bwd']:8.1f} us ({fl_b / out['bwd'] / 1e6:6.1f} TFLOP/s)   pos-score GEMM + fwd {out['gemm_fwd']:7.1f} us   "
          f"fwd with in-kernel pos score {out['fwd_pos']:7.1f} us  (no score tile written: {out['fwd_pos_inf']:7.1f} us)\n"
          f"        backward incl. positional gradients: kernels + un-shift + 2 products + batch sum {out['bwd_unfused_all']:7.1f} us   "
          f"in the dQ kernel {out['bwd_pos']:7.1f} us", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    for dh in (128, 64):
        run(a.batch, 4, 256, dh, a.drop, a.iters)
        run(a.batch, 4, 256, dh, 0.0, a.iters)
    # the unshift of d(bias) that follows every backward (elementwise.hip: relshift_bwd)
    db = torch.randn((a.batch, 4, 256, 256), device="cuda").to(torch.bfloat16)
    for _ in range(3):
        hip.relshift_bwd(db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        hip.relshift_bwd(db)
    e1.record()
    torch.cuda.synchronize()
    print("relshift_bwd (B,4,256,256) bf16: %.1f us" % (e0.elapsed_time(e1) / a.iters * 1e3))
