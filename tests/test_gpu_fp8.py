"""GPU: the OCP-fp8 (e4m3fn) GEMM path of BASELINE.json config 5 (csrc/gemm_fp8.hip).  The reference has no fp8 arithmetic
(fp32 / fp16-AMP, code/learner.py:46-50), so the path is pinned (a) kernel by kernel against torch's own float8_e4m3fn conversion
and an f64 product of the quantised operands, (b) end to end against this build's bf16 path - itself pinned to the reference - on
the 4-microphone 10-second segment of fixture F10, with the tolerances stated here."""
import json
import os

import numpy as np
import pytest
import torch

import recipes
from conftest import GOLD, check

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(300, 264), (4096, 512), (64, 2048)])
def test_fp8_quantize_matches_torch_e4m3fn(rows, cols, dtype):
    from sar_ssl_amd import hip
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn((rows, cols), generator=g) * torch.rand((rows, 1), generator=g) * 3).to(dtype).cuda()
    q, inv = hip.fp8_quantize(x)
    amax = x.float().abs().max()
    assert abs(inv.item() / (amax.item() / 448.0) - 1) < 1e-6
    want = (x.float() * (448.0 / amax)).to(torch.float8_e4m3fn)                     # torch: round-to-nearest-even, OCP encoding
    assert torch.equal(q.view(torch.float8_e4m3fn).float(), want.float())
    qt, inv_t = hip.fp8_quantize(x, transpose=True)
    assert tuple(qt.shape) == (cols, rows) and torch.equal(qt.t().contiguous(), q) and inv_t.item() == inv.item()
    z, inv_z = hip.fp8_quantize(torch.zeros((8, 16), dtype=dtype, device="cuda"))   # all-zero tensor: no NaNs
    assert int(z.max()) == 0 and inv_z.item() == 1.0


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (1000, 264, 528), (16384, 512, 2048)])
def test_fp8_gemm_vs_exact_product_of_the_quantised_operands(M, N, K):
    """With unit block scales the block-scaled MFMA is a plain fp8 x fp8 -> f32 dot product: the result equals the f64 product of the
    dequantised operands up to f32 accumulation order; and it stays within fp8 rounding of the bf16 GEMM on the original operands."""
    from sar_ssl_amd import hip
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn((N,), generator=g).cuda()
    xq, sx = hip.fp8_quantize(x)
    wq, sw = hip.fp8_quantize(w)
    y = hip.gemm_fp8(xq, sx, wq, sw, M=M, N=N, K=K, bias=bias, out_dtype=torch.float32)
    xd = xq.view(torch.float8_e4m3fn).double() * sx.double()
    wd = wq.view(torch.float8_e4m3fn).double() * sw.double()
    check("fp8.gemm_vs_dequantised_f64[%dx%dx%d]" % (M, N, K), _relerr(y, xd @ wd.t() + bias.double()), 2e-5)
    yb = hip.gemm(x, w, M=M, N=N, K=K, lda=K, ldb=K, bias=bias, out_dtype=torch.float32)
    check("fp8.gemm_vs_bf16_gemm[%dx%dx%d]" % (M, N, K), _relerr(y, yb), 5e-2)            # e4m3 operands: 2^-4 relative rounding per element
    # fused epilogue: Swish + pre-activation + residual + scale, bf16 output
    R = torch.randn((M, N), generator=g).to(torch.bfloat16).cuda()
    pre = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    y2 = hip.gemm_fp8(xq, sx, wq, sw, M=M, N=N, K=K, bias=bias, act=2, preact=pre, out_scale=0.5, resid=R, ldr=N, res_scale=1.0)
    h = xd @ wd.t() + bias.double()
    assert _relerr(pre, h) < 1e-2 and _relerr(y2, R.double() + 0.5 * h * torch.sigmoid(h)) < 1e-2


def _set_dropout(m, p):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = p


def _config5(prec):
    from sar_ssl_amd import hip, model, runtime
    z = np.load(os.path.join(GOLD, "f10_multich.npz"), allow_pickle=False)
    runtime.set_precision(prec)
    try:
        man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
        net = model.SARSSL(sig_shape=(256, 624, 2, 2), pretrain=True, device="cuda:0")
        net.load_state_dict(recipes.recipe_state_dict(man, 0))
        _set_dropout(net, 0.0)
        net.cuda().train()
        x = hip.stft_frontend(recipes.recipe_signal(1, 160000, 4, seed=21).cuda())
        net.set_masks(z["c5.mask_idx"], z["c5.mask_ch"])
        loss, diff, vis = net(x)
        loss.backward()
        grads = {k: p.grad.double().norm().item() for k, p in net.named_parameters()}
        return float(loss), vis["pred"].float().clone(), grads, z
    finally:
        runtime.set_precision("bf16")


def test_config5_fp8_path_tracks_the_bf16_path_and_the_reference():
    """BASELINE.json config 5: 4 microphones x 10 s (3 pairs, T = 624) through forward + backward with the fp8 GEMM path, against
    the bf16 path on identical inputs / weights / masks, and against the reference's own loss (fixture F10).  Stated tolerances:
    loss 5e-3 of the bf16 / reference loss; individual outputs 2e-1 of the output range (with the recipe weights the prediction is
    ~10x smaller than the data it regresses, so e4m3 noise of the block outputs is large relative to it while the loss moves by
    1e-3); per-parameter gradient norms 25 %."""
    l8, p8, g8, z = _config5("fp8")
    l16, p16, g16, _ = _config5("bf16")
    check("fp8.config5.loss_vs_bf16", abs(l8 / l16 - 1), 5e-3)
    check("fp8.config5.loss_vs_reference", abs(l8 / float(z["c5.loss"]) - 1), 5e-3)
    check("fp8.config5.pred_vs_bf16", ((p8 - p16).abs().max() / p16.abs().max()).item(), 2e-1)
    top = max(g16.values())
    worst = max((abs(g8[k] - g16[k]) / g16[k], k) for k in g16 if g16[k] > 1e-6 * top)
    check("fp8.config5.gradnorm_vs_bf16[worst=%s]" % worst[1], worst[0], 0.25)
    assert l8 != l16                                                    # the fp8 kernels really ran


def test_fp8_mode_trains():
    """A few optimiser steps in fp8 mode (dropout on, fused Adam): finite, and step by step within 2 % of the bf16 run on the same
    data, masks and dropout seeds."""
    from sar_ssl_amd import hip, model, runtime, synth
    import random
    losses = {}
    for prec in ("bf16", "fp8"):
        runtime.set_precision(prec)
        try:
            torch.manual_seed(3)
            net = model.SARSSL(sig_shape=(256, 64, 2, 2), pretrain=True, device="cuda:0").cuda().train()
            man = {k: list(v.shape) for k, v in net.state_dict().items()}
            net.load_state_dict(recipes.recipe_state_dict(man, 4))
            flat = runtime.FlatParams(net)
            opt = runtime.FusedAdam(flat, lr=1e-3)
            opt.zero_grad()
            sig = torch.from_numpy(synth.make_batch(0, 8, nsample=512 + 256 * 63)).cuda()
            runtime.RT.manual_seed(11)
            random.seed(5)
            cur = []
            for _ in range(6):
                loss, _, _ = net(hip.stft_frontend(sig))
                loss.backward()
                opt.step()
                opt.zero_grad()
                cur.append(float(loss))
            losses[prec] = cur
        finally:
            runtime.set_precision("bf16")
    assert all(np.isfinite(losses["fp8"]))
    check("fp8.train6.loss_vs_bf16", max(abs(a / b - 1) for a, b in zip(losses["fp8"], losses["bf16"])), 2e-2)
