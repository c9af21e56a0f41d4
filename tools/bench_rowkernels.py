"""Isolated timings of the LayerNorm / depthwise-conv-module tiles at the shapes of the timed step (B = 64, T = 256: M = 16384 rows,
d = 256 for the three spatial blocks, 512 for the spectral one), fp16 forward tensors / bf16 gradients, next to each launch's HBM
floor (algorithmic bytes at 5 TB/s).     python tools/bench_rowkernels.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip
dev = torch.device("cuda:0")
B, T = 64, 256
M = B * T
F16, BF = torch.float16, torch.bfloat16


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    hip.gpu_runway(3.0)                 # the GPU is kept busy while the host enqueues the loop: launches of a few us are host-bound otherwise
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def row(name, us, nbytes):
    print("%-58s %7.1f us   floor %5.1f us (%5.1f MB)   x%.1f" % (name, us, nbytes / 5e6, nbytes / 1e6, us / (nbytes / 5e6)), flush=True)


for d in (256, 512):
    x = torch.randn((M, d), device=dev).to(F16)
    dy = torch.randn((M, d), device=dev).to(BF)
    res = torch.randn((M, d), device=dev).to(BF)
    g, bt = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    y, st = hip.layernorm_fwd(x, g, bt)
    row("layernorm_fwd d=%d" % d, timeit(lambda: hip.layernorm_fwd(x, g, bt)), 2 * 2 * M * d)
    row("layernorm_bwd d=%d (+resid, dgamma/dbeta)" % d, timeit(lambda: hip.layernorm_bwd(dy, x, g, st, resid=res, dgamma=dg, dbeta=db)), 4 * 2 * M * d)
    row("layernorm_bwd d=%d (+resid, params, dropped copy)" % d,
        timeit(lambda: hip.layernorm_bwd(dy, x, g, st, resid=res, dgamma=dg, dbeta=db, drop=(0.1, 5, 1.0))), 5 * 2 * M * d)
    h = torch.randn((M, 2 * d), device=dev).to(F16)
    w = torch.randn((d, 31), device=dev)
    dc = torch.randn((M, d), device=dev).to(BF)
    dw = torch.zeros((d, 31), device=dev)
    row("dwglu_fwd d=%d (+BatchNorm sums)" % d, timeit(lambda: hip.dwglu_fwd(h, w, B, T, want_stats=True)), 2 * M * 3 * d)
    row("dwglu_bwd d=%d" % d, timeit(lambda: hip.dwglu_bwd(dc, h, w, B, T)), 2 * M * 5 * d)
    row("dwglu_wgrad d=%d" % d, timeit(lambda: hip.dwglu_wgrad(dc, h, dw, B, T)), 2 * M * 3 * d)
    c = torch.randn((M, d), device=dev).to(F16)
    aff = torch.randn((4, d), device=dev)
    row("cl_affine_act d=%d (BatchNorm + Swish)" % d, timeit(lambda: hip.cl_affine_act(c, d, aff, 2)), 2 * 2 * M * d)
    hip.sums_arena_reset(dev)
    red = hip.cl_bn_bwd_reduce(dc, c, d, aff, 2)

    def bn_reduce():
        hip.sums_arena_reset(dev)                               # (the arena hands out one slice per call; a launch of its own, ~2 us)
        hip.cl_bn_bwd_reduce(dc, c, d, aff, 2)
    row("cl_bn_bwd_reduce d=%d (Swish; + arena reset + memset launches)" % d, timeit(bn_reduce), 2 * 2 * M * d)
    outb = torch.empty_like(dc)
    row("cl_bn_bwd_apply d=%d (Swish)" % d, timeit(lambda: hip.cl_bn_bwd_apply(dc, c, d, aff, 2, False, True, red, out=outb)), 3 * 2 * M * d)
    q = torch.randn((M, d), device=dev).to(F16)
    u, v = torch.randn(d, device=dev), torch.randn(d, device=dev)
    row("bias2 d=%d (q+u, q+v)" % d, timeit(lambda: hip.bias2(q, u, v)), 3 * 2 * M * d)
