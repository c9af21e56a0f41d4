"""Hybrid numeric mode (round 6): fp16 CNN stem, f32 residual stream in the Conformer blocks / decoder, f32 tensors and weights entering
the forward products as fp16 pairs (csrc/hybrid.hip, sarssl_gemm_split).  Kernel tests against f64 references computed with torch on
the same inputs; the model-level gates (F3 / F13 / F14 / F15) live next to the other modes' in test_gpu_model.py / test_gpu_graph.py.
"""
import numpy as np
import pytest
import torch

from conftest import check

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max()).item()


def test_split_pair_reconstructs_22_bits():
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(1 << 16, generator=g) * torch.exp(torch.randn(1 << 16, generator=g))).to(dev)
    p = hip.split_pair(x)
    assert p.hi.dtype == torch.float16 and p.lo.dtype == torch.float16
    assert torch.equal(p.hi, x.half())
    assert torch.equal(p.lo, (x - x.half().float()).half())
    # 22 significant bits while the lo part is a normal fp16 number (|x| >= 0.25), fp16's subnormal step (2^-24) below
    err = ((p.float() - x).abs() / torch.maximum(x.abs() * 2.0 ** -21, torch.full_like(x, 2.0 ** -24))).max().item()
    check("hybrid.split_pair.err_over_bound", err, 1.0)
    lo = hip.split_pair(x, want_hi=False)
    assert torch.equal(lo, p.lo)


@pytest.mark.parametrize("shape", [(512, 1024, 256), (256, 256, 1024), (80, 96, 32), (130, 64, 40)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16])
def test_gemm_split_vs_f64(shape, out_dtype):
    """hi hi + lo hi + hi lo on the fp16 matrix cores: ~2^-21 per product against 2^-11 of a single fp16 pass.  Weight scale 0.03 puts
    the weights' lo parts into fp16's subnormal range - they must not be flushed."""
    from sar_ssl_amd import hip
    dev = _dev()
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev).to(out_dtype)
    ref = x.double() @ w.double().t()
    xp, wp = hip.split_pair(x), hip.split_pair(w)
    # three segments, plain product
    y = hip.gemm_split(xp, wp.hi, wp.lo, M=M, N=N, K=K, out_dtype=torch.float32)
    e3 = _rel(y, ref)
    # two segments (fp16 activation against the weight pair)
    y2 = hip.gemm_split(xp.hi, wp.hi, wp.lo, M=M, N=N, K=K, out_dtype=torch.float32)
    e2 = _rel(y2, xp.hi.double() @ w.double().t())
    # one segment = the plain fp16 product
    y1 = hip.gemm_split(xp.hi, wp.hi, None, M=M, N=N, K=K, out_dtype=torch.float32)
    e1 = _rel(y1, ref)
    check("hybrid.gemm_split.%dx%dx%d.3seg" % shape, e3, 3e-6)
    check("hybrid.gemm_split.%dx%dx%d.2seg" % shape, e2, 3e-6)
    assert e1 > 20 * e3, (e1, e3)            # (the single pass is the 2^-11 class: the split is what buys the accuracy)
    # epilogue: bias + swish + saved pre-activation; bias + scaled residual
    pre = torch.empty((M, N), dtype=out_dtype, device=dev)
    ya = hip.gemm_split(xp, wp.hi, wp.lo, M=M, N=N, K=K, out_dtype=out_dtype, bias=b, act=2, preact=pre)
    z = ref + b.double()
    tol = 2e-3 if out_dtype == torch.float16 else 1e-5
    check("hybrid.gemm_split.%dx%dx%d.preact.%s" % (shape + (str(out_dtype)[6:],)), _rel(pre, z), tol)
    check("hybrid.gemm_split.%dx%dx%d.swish.%s" % (shape + (str(out_dtype)[6:],)), _rel(ya, z * torch.sigmoid(z)), tol)
    yr = hip.gemm_split(xp, wp.hi, wp.lo, M=M, N=N, K=K, out_dtype=out_dtype, bias=b, out_scale=0.5, resid=r, ldr=N, res_scale=1.0)
    check("hybrid.gemm_split.%dx%dx%d.resid.%s" % (shape + (str(out_dtype)[6:],)), _rel(yr, 0.5 * z + r.double()), tol)
    # dropout: the mask is the same function of (seed, element index) as sarssl_gemm's
    yd = hip.gemm_split(xp, wp.hi, wp.lo, M=M, N=N, K=K, out_dtype=torch.float32, p_drop=0.25, seed=77)
    keep = (yd != 0)
    frac = keep.float().mean().item()
    assert 0.70 < frac < 0.80, frac
    assert _rel(yd[keep], (ref / 0.75)[keep]) < 1e-5
    x16 = xp.hi
    yd16 = hip.gemm(x16, wp.hi, M=M, N=N, K=K, lda=K, ldb=K, out_dtype=torch.float32, p_drop=0.25, seed=77)
    assert torch.equal(yd16 != 0, keep)


@pytest.mark.parametrize("shape", [(16384, 256), (8192, 512), (300, 32), (77, 768)])
def test_layernorm_pair_and_stream_backward(shape):
    from sar_ssl_amd import hip
    dev = _dev()
    M, d = shape
    g = torch.Generator(device="cpu").manual_seed(M + d)
    x = (torch.randn(M, d, generator=g) * 2 + 0.3).to(dev)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(dev)
    beta = (0.1 * torch.randn(d, generator=g)).to(dev)
    ref = torch.nn.functional.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-5)
    p, stats = hip.layernorm_fwd_pair(x, gamma, beta, 1e-5)
    check("hybrid.ln_pair.%dx%d" % shape, _rel(p.float(), ref), 2e-6)
    y32, st32 = hip.layernorm_fwd(x, gamma, beta, 1e-5)                 # the f32 kernel: same arithmetic, the pair is its split
    assert torch.equal(p.hi, y32.half()) and torch.equal(stats, st32)
    assert torch.equal(p.lo, (y32 - y32.half().float()).half())
    p2, _, y2 = hip.layernorm_fwd_pair(x, gamma, beta, 1e-5, want32=True)
    assert torch.equal(y2, y32) and torch.equal(p2.hi, p.hi)
    p3, st3 = hip.layernorm_fwd_pair(x, gamma, beta, 1e-5, want_lo=False)            # hi half only
    assert p3.lo is None and torch.equal(p3.hi, p.hi) and torch.equal(st3, stats)
    # two LayerNorms in a row
    gb = (1 + 0.1 * torch.randn(d, generator=g)).to(dev)
    bb = (0.1 * torch.randn(d, generator=g)).to(dev)
    ya, sa, z, sb = hip.layernorm_fwd2_pair(x, gamma, beta, 1e-5, gb, bb, 1e-5)
    assert torch.equal(ya, y32) and torch.equal(sa, st32)
    zp, zs = hip.layernorm_fwd_pair(ya, gb, bb, 1e-5)
    assert torch.equal(z.hi, zp.hi) and torch.equal(z.lo, zp.lo) and torch.equal(sb, zs)
    ya3, _, z3, sb3 = hip.layernorm_fwd2_pair(x, gamma, beta, 1e-5, gb, bb, 1e-5, want_lo=False)
    assert z3.lo is None and torch.equal(ya3, ya) and torch.equal(z3.hi, z.hi) and torch.equal(sb3, sb)
    # backward on the stream: f32 and bf16 branch gradients, with and without the dropped bf16 copy
    xr = x.double().requires_grad_(True)
    dyf = torch.randn(M, d, generator=g).to(dev)
    res = torch.randn(M, d, generator=g).to(dev)
    for dy in (dyf, dyf.bfloat16()):
        xr.grad = None
        gr = gamma.double().requires_grad_(True)
        br = beta.double().requires_grad_(True)
        torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5).backward(dy.double())
        dg = torch.zeros(d, device=dev)
        db = torch.zeros(d, device=dev)
        dx, dx16 = hip.layernorm_bwd_stream(dy, x, gamma, stats, resid=res, dgamma=dg, dbeta=db, drop=(0.1, 1234, 0.5))
        tag = "hybrid.ln_bwd_stream.%dx%d.%s" % (shape + (str(dy.dtype)[6:],))
        check(tag + ".dx", _rel(dx, xr.grad + res.double()), 1e-5)
        check(tag + ".dgamma", _rel(dg, gr.grad), 2e-5)
        check(tag + ".dbeta", _rel(db, br.grad), 2e-5)
        assert dx16.dtype == torch.bfloat16
        # the dropped copy = act_bwd of the rounded gradient with the same (p, seed, gscale)
        want = hip.act_bwd(dx.contiguous(), None, 0, p_drop=0.1, seed=1234, gscale=0.5).bfloat16()
        assert torch.equal(dx16, want)
        dx_b, c16 = hip.layernorm_bwd_stream(dy, x, gamma, stats, resid=res, copy16=True)
        assert torch.equal(dx_b, dx) and torch.equal(c16, dx.bfloat16())


@pytest.mark.parametrize("act_pair", [False, True])
@pytest.mark.parametrize("train", [False, True])
def test_fused_feed_forward_on_the_f32_stream_equals_the_unfused_sequence(train, act_pair, monkeypatch):
    """csrc/ffn2h.hip (LayerNorm + both Linear layers in one launch, d = 256) against the hybrid mode's unfused sequence (LayerNorm -> pair,
    two sarssl_gemm_split launches): the tensors saved for the backward pass bit for bit (same MFMA products in the same order, same
    dropout masks), the f32 result to an ulp (one fused multiply-add in the residual), and the result against an f64 reference.
    act_pair: LN(x) enters the first product as a pair (three products) or as its fp16 rounding (two; engine._H_ALO)."""
    from sar_ssl_amd import engine, runtime
    from sar_ssl_amd.common.conformer.feed_forward import FeedForwardModule
    dev = _dev()
    runtime.set_precision("hybrid")
    try:
        torch.manual_seed(5)
        ff = FeedForwardModule(encoder_dim=256, expansion_factor=4, dropout_p=0.1).to(dev).train(train)
        with torch.no_grad():
            ff.sequential[0].weight.add_(0.1 * torch.randn(256, device=dev)); ff.sequential[0].bias.add_(0.1 * torch.randn(256, device=dev))
        x = (torch.randn(2048, 256, device=dev) * 1.5 + 0.2)
        res = {}
        monkeypatch.setattr(engine, "_H_ALO", {"ffn1"} if act_pair else set())
        tag = "hybrid.ffn2h.%s.%s" % ("pair" if act_pair else "hi", "train" if train else "eval")
        for fused in (True, False):
            monkeypatch.setattr(engine, "_H_FFN2_FWD", fused)
            runtime.RT.manual_seed(99)
            saved = []
            y = engine.ffn_fwd(x.clone(), ff, 0.5, train, saved)
            res[fused] = (y, saved[0])
        (yf, sf), (yu, su) = res[True], res[False]
        for i, name in ((1, "ln_hi"), (2, "stats")):
            assert torch.equal(sf[i], su[i]), name
        # pre-activation / hidden tensor: the same products accumulated in another order (k-step-major here, segment-major there) - equal
        # up to the f32 summation order, i.e. an fp16 ulp on a fraction of the entries
        for i, name in ((3, "preact"), (4, "hidden")):
            a, b = sf[i].float(), su[i].float()
            check("%s.%s_vs_unfused" % (tag, name), ((a - b).abs().max() / b.abs().max()).item(), 1e-3)
            assert (a != b).float().mean().item() < 0.02, name
            if train and name == "hidden":
                assert torch.equal(a == 0, b == 0)            # the same dropout mask
        assert sf[5:] == su[5:]
        check(tag + ".y_vs_unfused", _rel(yf, yu), 2e-4)
        if not train:
            seq = ff.sequential
            xd = x.double()
            ln = torch.nn.functional.layer_norm(xd, (256,), seq[0].weight.double(), seq[0].bias.double(), seq[0].eps)
            h = torch.nn.functional.silu(ln @ seq[1].linear.weight.double().t() + seq[1].linear.bias.double())
            ref = xd + 0.5 * (h @ seq[4].linear.weight.double().t() + seq[4].linear.bias.double())
            # the hidden activation is an fp16 tensor (2^-12 per element): the module's contribution carries that, the stream itself is f32
            check(tag + ".y_vs_f64", _rel(yf, ref), 2e-4)
    finally:
        runtime.set_precision("bf16")


@pytest.mark.parametrize("T,dh,H", [(256, 64, 4), (256, 128, 4), (128, 64, 4)])
def test_attention_forward_writes_the_context_pair(T, dh, H):
    """sarssl_relpos_attn_fwd_pos_pair: the same launch as sarssl_relpos_attn_fwd_pos, plus the lo half of its unrounded f32 context -
    bit for bit what sarssl_split_pair makes of ctx32 (the separate pass it replaces in the hybrid mode's attention module)."""
    from sar_ssl_amd import hip
    dev = _dev()
    B, d = 3, H * dh
    g = torch.Generator().manual_seed(11)
    q, k, v = [(torch.randn(B * T, d, generator=g) * 0.7).half().to(dev) for _ in range(3)]
    pos = (torch.randn(T, d, generator=g) * 0.7).half().to(dev)
    ub, vb = (torch.randn(d, generator=g) * 0.1).to(dev), (torch.randn(d, generator=g) * 0.1).to(dev)
    scale = d ** -0.5
    for p_drop in (0.0, 0.1):
        ctx, (c32, lse), bias = hip.relpos_attn_fwd_pos(q, q, k, v, pos, B, H, T, dh, scale, p_drop, 77, biases=(ub, vb), want_ctx32=True)
        pr, (c32p, lsep), biasp = hip.relpos_attn_fwd_pos(q, q, k, v, pos, B, H, T, dh, scale, p_drop, 77, biases=(ub, vb), pair=True)
        assert isinstance(pr, hip.Pair) and torch.equal(pr.hi, ctx) and torch.equal(c32p, c32) and torch.equal(lsep, lse) and torch.equal(biasp, bias)
        ref = hip.split_pair(c32)
        assert torch.equal(ref.hi, ctx) and torch.equal(pr.lo, ref.lo)
        # inference: no f32 copy is kept next to the pair
        pi, (c32i, _), bi = hip.relpos_attn_fwd_pos(q, q, k, v, pos, B, H, T, dh, scale, 0.0, 0, need_bwd=False, biases=(ub, vb), pair=True)
        assert c32i is None and bi is None
        if p_drop == 0.0:
            assert torch.equal(pi.hi, ctx) and torch.equal(pi.lo, pr.lo)


def test_scatter_with_the_dropped_bf16_copy_equals_the_three_passes():
    """sarssl_scatter_rows_drop16 (block tails on the gathered rows, hybrid mode) = sarssl_scatter_rows + sarssl_cast + sarssl_act_bwd, bit for bit."""
    from sar_ssl_amd import hip
    dev = _dev()
    B, T, nm, d = 5, 64, 23, 256
    g = torch.Generator().manual_seed(3)
    idx = torch.stack([torch.randperm(T, generator=g)[:nm].sort().values for _ in range(B)]).to(torch.int32).to(dev)
    src = torch.randn(B * nm, d, generator=g).to(dev)
    for p, gs in ((0.1, 0.5), (0.25, 1.0)):
        full = hip.scatter_rows(src, idx, B, T)
        want16 = hip.act_bwd(hip.cast(full, torch.bfloat16), None, 0, p_drop=p, seed=4242, gscale=gs)
        d32, d16 = hip.scatter_rows_drop16(src, idx, B, T, p, 4242, gs)
        assert torch.equal(d32, full) and d16.dtype == torch.bfloat16 and torch.equal(d16, want16)
        assert (d16 == 0).float().mean().item() > 1.0 - nm / T - 1e-6          # (rows nobody scattered to stay zero)
