"""GPU parity tests of individual HIP kernels against plain PyTorch fp32 / the oracle."""
import os

import numpy as np
import pytest
import torch

import recipes
import sarssl_oracle as orc
from conftest import GOLD

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# ---------------------------------------------------------------- GEMM
LAYOUTS = [(True, True), (True, False), (False, True), (False, False)]


def _mk(shape, dtype, dev, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(dtype).to(dev)


@pytest.mark.parametrize("a_kc,b_kc", LAYOUTS)
@pytest.mark.parametrize("mode", ["bf16", "bf16_f32out", "f32_precise", "f32_fast", "fp16", "fp16_f32out", "mix_g_a", "mix_g_a_f32out", "mix_a_g"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 40), (48, 136, 32), (1000, 264, 520)])
def test_gemm_layouts(a_kc, b_kc, mode, M, N, K):
    """fp16*: both operands fp16 (v_mfma_f32_32x32x16_f16: the forward products of the fp16-forward mode); mix_g_a / mix_a_g: a bf16
    gradient times a saved fp16 activation (either side), contracted in bf16 with the fp16 operand re-encoded while staged."""
    from sar_ssl_amd import hip
    dev = _dev()
    if not a_kc and M % 8:
        pytest.skip("transposed A needs M % 8 == 0")
    in_dt = torch.bfloat16 if mode.startswith("bf16") else (torch.float16 if mode.startswith("fp16") else torch.float32)
    out_dt = torch.bfloat16 if mode == "bf16" else (torch.float16 if mode == "fp16" else torch.float32)
    A = _mk((M, K) if a_kc else (K, M), in_dt, dev, 1)
    B = _mk((N, K) if b_kc else (K, N), in_dt, dev, 2)      # asymmetric random operands
    if mode.startswith("mix"):
        a_dt, b_dt = (torch.bfloat16, torch.float16) if mode.startswith("mix_g_a") else (torch.float16, torch.bfloat16)
        A, B = A.to(a_dt), B.to(b_dt)
        out_dt = torch.float32 if mode.endswith("f32out") else torch.bfloat16
    bias = _mk((N,), torch.float32, dev, 3)
    C = hip.gemm(A, B, a_kc=a_kc, b_kc=b_kc, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], out_dtype=out_dt,
                 bias=bias, alpha=0.5, precise=(mode == "f32_precise"))
    assert C.dtype == out_dt
    Af = A.float() if a_kc else A.float().t()
    Bf = B.float() if b_kc else B.float().t()
    if mode.startswith("mix"):          # the fp16 operand enters the contraction rounded to bf16
        Af, Bf = Af.bfloat16().float(), Bf.bfloat16().float()
    ref = 0.5 * (Af.double() @ Bf.double().t()) + bias.double()
    err = _relerr(C, ref)
    tol = {"bf16": 1e-2, "bf16_f32out": 1e-5, "f32_precise": 5e-5, "f32_fast": 1e-2, "fp16": 1.5e-3, "fp16_f32out": 1e-5, "mix_g_a": 1e-2,
           "mix_g_a_f32out": 1e-5, "mix_a_g": 1e-2}[mode]
    if mode == "bf16_f32out":
        tol = 1e-5       # bf16 inputs are exact in the reference too; only f32 accumulation order differs
    assert err < tol, (mode, err)


@pytest.mark.parametrize("a_kc,b_kc", LAYOUTS)
@pytest.mark.parametrize("M,N,K,nbatch", [(4096, 1024, 256, 1), (512, 256, 768, 128), (520, 264, 776, 60)])
def test_gemm_large_tiles_all_epilogues(a_kc, b_kc, M, N, K, nbatch):
    """Grids large enough for the 256 x 128-tile instantiations (plain and ragged-edge): batched product with bias, Swish,
    pre-activation side output, dropout, output scale and residual, in bf16 against an f64 reference of the same bf16 operands;
    and the split-K accumulate path with f32 output."""
    from sar_ssl_amd import hip
    dev = _dev()
    A = _mk((nbatch, M, K) if a_kc else (nbatch, K, M), torch.bfloat16, dev, 11)
    B = _mk((nbatch, N, K) if b_kc else (nbatch, K, N), torch.bfloat16, dev, 12)
    bias = _mk((N,), torch.float32, dev, 13)
    R = _mk((nbatch, M, N), torch.bfloat16, dev, 14)
    pre = torch.empty((nbatch, M, N), dtype=torch.bfloat16, device=dev)
    kw = dict(a_kc=a_kc, b_kc=b_kc, M=M, N=N, K=K, lda=A.shape[2], ldb=B.shape[2], nbatch=nbatch, sA=(A.shape[1] * A.shape[2], 0),
              sB=(B.shape[1] * B.shape[2], 0), sC=(M * N, 0), sR=(M * N, 0))
    Y = hip.gemm(A, B, bias=bias, act=2, preact=pre, p_drop=0.1, seed=77, out_scale=0.5, resid=R, ldr=N, res_scale=1.0, alpha=0.25,
                 out_shape=(nbatch, M, N), **kw)
    Af = A.double() if a_kc else A.double().transpose(1, 2)
    Bf = B.double() if b_kc else B.double().transpose(1, 2)
    h = 0.25 * torch.bmm(Af, Bf.transpose(1, 2)) + bias.double()
    assert _relerr(pre, h) < 1e-2
    Y0 = hip.gemm(A, B, bias=bias, act=2, out_scale=0.5, resid=R, ldr=N, res_scale=1.0, alpha=0.25, out_shape=(nbatch, M, N), **kw)
    want0 = R.double() + 0.5 * h * torch.sigmoid(h)
    assert _relerr(Y0, want0) < 1e-2
    branch = 0.5 * h * torch.sigmoid(h)                                              # what dropout acts on
    sizable = branch.abs() > 0.05 * branch.abs().max()
    dropped = ((Y.double() - R.double()).abs() < 0.2 * branch.abs()) & sizable        # dropped elements are exactly the residual
    frac = dropped.float().sum().item() / sizable.float().sum().item()
    assert 0.07 < frac < 0.13, frac
    want = R.double() + branch / 0.9
    kept = sizable & ~dropped
    assert ((Y.double() - want).abs()[kept].max() / want.abs().max()).item() < 1.5e-2
    if nbatch == 1:                                                    # split-K accumulate (weight-gradient form), f32 output
        G = torch.ones((M, N), dtype=torch.float32, device=dev)
        hip.gemm(A[0], B[0], a_kc=a_kc, b_kc=b_kc, M=M, N=N, K=K, lda=A.shape[2], ldb=B.shape[2], out=G, ldc=N, split_k=3)
        assert _relerr(G, 1.0 + torch.bmm(Af, Bf.transpose(1, 2))[0]) < 1e-5


def test_gemm_epilogues_and_batch():
    from sar_ssl_amd import hip
    dev = _dev()
    nb, H, T, dh = 3, 4, 40, 16
    d = H * dh
    # batched attention-like product: scores[b,h] = q[b,:,h,:] @ k[b,:,h,:]^T
    q = _mk((nb, T, d), torch.float32, dev, 4)
    k = _mk((nb, T, d), torch.float32, dev, 5)
    s = hip.gemm(q, k, M=T, N=T, K=dh, lda=d, ldb=d, nbatch=nb * H, batch_inner=H, sA=(T * d, dh), sB=(T * d, dh),
                 precise=True, out_shape=(nb, H, T, T))
    ref = torch.einsum("bihd,bjhd->bhij", q.view(nb, T, H, dh).double(), k.view(nb, T, H, dh).double())
    assert _relerr(s, ref) < 5e-5
    # epilogue: swish + residual + preact
    M, N, K = 96, 64, 48
    X = _mk((M, K), torch.float32, dev, 6); W = _mk((N, K), torch.float32, dev, 7)
    b = _mk((N,), torch.float32, dev, 8); R = _mk((M, N), torch.float32, dev, 9)
    pre = torch.empty((M, N), dtype=torch.float32, device=dev)
    Y = hip.gemm(X, W, M=M, N=N, K=K, lda=K, ldb=K, bias=b, act=2, resid=R, ldr=N, res_scale=1.0, out_scale=0.5,
                 preact=pre, precise=True)
    h = X.double() @ W.double().t() + b.double()
    assert _relerr(pre, h) < 5e-5
    assert _relerr(Y, R.double() + 0.5 * h * torch.sigmoid(h)) < 5e-5
    Yr = hip.gemm(X, W, M=M, N=N, K=K, lda=K, ldb=K, bias=b, act=1, precise=True)
    assert _relerr(Yr, torch.relu(h)) < 5e-5
    # dropout: same seed -> same mask, keep-rate ~ 0.9, kept values scaled by 1/0.9
    Yd1 = hip.gemm(X, W, M=M, N=N, K=K, lda=K, ldb=K, bias=b, p_drop=0.1, seed=123, precise=True)
    Yd2 = hip.gemm(X, W, M=M, N=N, K=K, lda=K, ldb=K, bias=b, p_drop=0.1, seed=123, precise=True)
    assert torch.equal(Yd1, Yd2)
    keep = (Yd1 != 0).float().mean().item()
    assert 0.85 < keep < 0.95
    kept = Yd1 != 0
    assert _relerr(Yd1[kept], (h.float() / 0.9)[kept]) < 1e-4


# ---------------------------------------------------------------- STFT front-end
def test_stft_frontend_vs_golden_and_oracle():
    from sar_ssl_amd import hip
    dev = _dev()
    z = np.load(os.path.join(GOLD, "f1_frontend.npz"))
    small = recipes.recipe_signal(2, 2048, 2, seed=1)
    out = hip.stft_frontend(small.to(dev))
    assert _relerr(out, torch.from_numpy(z["small_out"])) < 1e-5
    X = hip.stft_raw(small.to(dev))
    assert _relerr(X.real, torch.from_numpy(z["small_stft_re"])) < 1e-5
    assert _relerr(X.imag, torch.from_numpy(z["small_stft_im"])) < 1e-5
    small4 = recipes.recipe_signal(2, 1536, 4, seed=2)
    out4 = hip.stft_frontend(small4.to(dev))
    assert _relerr(out4, torch.from_numpy(z["small4_out"])) < 1e-5
    full = recipes.recipe_signal(2, 65792, 2, seed=3)
    outf = hip.stft_frontend(full.to(dev))
    assert list(outf.shape) == [2, 2, 256, 256, 2]
    ref = orc.data_preprocess(full)
    assert _relerr(outf, ref) < 1e-5
    idx = torch.from_numpy(z["full_idx"])
    got = outf.cpu().reshape(-1)[idx]
    want = torch.from_numpy(z["full_vals"])
    # per-bin magnitudes within 1e-3 relative (north_star) - we are ~1e-6
    big = want.abs() > 1e-3
    assert ((got[big] - want[big]).abs() / want[big].abs()).max().item() < 1e-3
    # int16 PCM input path == float path on the same quantised samples
    pcm = torch.clamp(torch.round(full * 32768.0), -32768, 32767).to(torch.int16)
    outq = hip.stft_frontend(pcm.to(dev))
    refq = orc.data_preprocess(pcm.float() / 32768.0)
    assert _relerr(outq, refq) < 1e-5
    # odd channel count (3 mics -> 2 pairs), ragged frame count (nt not a multiple of 16)
    s3 = recipes.recipe_signal(1, 512 + 256 * 20, 3, seed=4)
    assert _relerr(hip.stft_frontend(s3.to(dev)), orc.data_preprocess(s3)) < 1e-5


# ---------------------------------------------------------------- conv stem kernels
@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("nch,ch_mode,pcm", [(2, "M", True), (3, "M", False), (4, "MM", True)])
def test_frontend_with_masks_in_the_same_pass_equals_the_two_launch_sequence(nch, ch_mode, pcm, dtp):
    """`sarssl_stft_frontend_pairs_masked` (what the captured step calls): data_preprocess output and the two encoders' masked inputs
    from ONE pass over the spectrum - bit for bit what `stft_frontend` followed by `mask_inputs` (pinned on F1 / the oracle above and
    below) produce; 2 / 3 / 4 microphones, both pairing modes, int16 PCM and f32 signals, ragged frame count."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(nch * 7 + len(ch_mode))
    B, nt = 3, 21
    sig = torch.randn((B, 512 + 256 * (nt - 1) + 37, nch), generator=g) * 0.1
    sig = (sig * 32767).clamp(-32768, 32767).to(torch.int16) if pcm else sig
    npair = nch - 1 if ch_mode == "M" else nch * (nch - 1) // 2
    mp = (torch.rand((B * npair, nt), generator=g) < 0.5).to(torch.uint8).to(dev)
    ch = torch.randint(0, 2, (B * npair,), generator=g).to(torch.int32).to(dev)
    x = hip.stft_frontend(sig.to(dev), ch_mode=ch_mode)
    spec, spat = hip.mask_inputs(x, mp, ch, 0, dtp)
    x2, spec2, spat2 = hip.stft_frontend(sig.to(dev), ch_mode=ch_mode, masks=(mp, ch), dtype=dtp)
    assert x2.shape == x.shape and spec2.shape == spec.shape == (B * npair, 256, nt, 4)
    assert torch.equal(x2, x) and torch.equal(spec2, spec) and torch.equal(spat2, spat)
    assert float(spec2.float().abs().max()) > 0 and not torch.equal(spec2, spat2)


def _cl(x_nchw):          # (B,C,F,T) -> channels-last (B,F,T,C)
    return x_nchw.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("mode", ["bf16", "fp16", "f32_precise"])
@pytest.mark.parametrize("B,F,T", [(2, 16, 8), (1, 24, 136), (3, 8, 64)])
@pytest.mark.parametrize("prologue", [False, True])
def test_conv3x3_fwd(mode, B, F, T, prologue):
    from sar_ssl_amd import hip
    dev = _dev()
    dtp = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(mode, torch.float32)
    g = torch.Generator().manual_seed(B * 100 + F + T)
    x = torch.randn((B, 64, F, T), generator=g)
    W = torch.randn((64, 64, 3, 3), generator=g) * 0.05
    sc = torch.rand(64, generator=g) + 0.5
    sh = torch.randn(64, generator=g) * 0.3
    xq = x.to(dtp).float(); Wq = W.to(dtp).float()
    z = torch.relu(xq * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if prologue else xq
    if mode in ("bf16", "fp16") and prologue:
        z = z.to(dtp).float()            # the kernel rounds the prologue output to its 16-bit operand type before the MFMA
    ref = torch.nn.functional.conv2d(z.double(), Wq.double(), padding=1)
    w_tap = Wq.permute(2, 3, 0, 1).reshape(9, 64, 64).contiguous().to(dtp).to(dev)
    out = hip.conv3x3_fwd(_cl(xq).to(dtp).to(dev), w_tap, sc.to(dev) if prologue else None, sh.to(dev) if prologue else None,
                          precise=(mode == "f32_precise"))
    err = _relerr(out.float().cpu(), _cl(ref))
    assert out.dtype == dtp
    assert err < {"bf16": 1e-2, "fp16": 1.5e-3}.get(mode, 5e-5), err
    if mode in ("bf16", "fp16"):            # fused BatchNorm statistics of the stored output
        out2, sums = hip.conv3x3_fwd(_cl(xq).to(dtp).to(dev), w_tap, sc.to(dev) if prologue else None, sh.to(dev) if prologue else None,
                                     want_stats=True)
        assert torch.equal(out2, out)
        o64 = out.float().reshape(-1, 64).double()
        assert _relerr(sums[:64], o64.sum(0)) < 1e-5 and _relerr(sums[64:], (o64 ** 2).sum(0)) < 1e-5


@pytest.mark.parametrize("mode", ["bf16", "mix16", "f32_precise"])
@pytest.mark.parametrize("B,F,T", [(2, 16, 8), (1, 24, 136)])
def test_conv3x3_wgrad_and_dgrad(mode, B, F, T):
    """mix16: the saved forward activation is fp16, the gradient bf16 (the backward pass of the fp16-forward mode)."""
    from sar_ssl_amd import hip
    dev = _dev()
    dtp = torch.bfloat16 if mode in ("bf16", "mix16") else torch.float32
    adt = torch.float16 if mode == "mix16" else dtp                # dtype of the saved forward tensor
    g = torch.Generator().manual_seed(7 + F)
    yprev = torch.randn((B, 64, F, T), generator=g).to(adt).float()
    dy = torch.randn((B, 64, F, T), generator=g).to(dtp).float()
    W = (torch.randn((64, 64, 3, 3), generator=g) * 0.05).to(dtp).float()
    sc = torch.rand(64, generator=g) + 0.5
    sh = torch.randn(64, generator=g) * 0.3
    z = torch.relu(yprev * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if mode in ("bf16", "mix16"):
        z = z.to(dtp).float()
    z64 = z.double().requires_grad_(True)
    W64 = W.double().requires_grad_(True)
    out = torch.nn.functional.conv2d(z64, W64, padding=1)
    out.backward(dy.double())
    dW = hip.conv3x3_wgrad(_cl(dy).to(dtp).to(dev), _cl(yprev).to(adt).to(dev), sc.to(dev), sh.to(dev), precise=(mode == "f32_precise"))
    ref_dW = W64.grad.permute(2, 3, 0, 1).reshape(9, 64, 64)
    assert _relerr(dW, ref_dW) < (5e-5 if mode == "f32_precise" else 1e-2)
    if mode != "f32_precise":        # the variant that adds straight into the (64,64,3,3) parameter-gradient buffer
        acc = torch.zeros((64, 64, 3, 3), device=dev)
        assert hip.conv3x3_wgrad(_cl(dy).to(dtp).to(dev), _cl(yprev).to(adt).to(dev), sc.to(dev), sh.to(dev), acc_into=acc) is None
        assert _relerr(acc, W64.grad) < 1e-2
    # data gradient = forward kernel with flipped / transposed weights
    w_dgrad = W.flip(2, 3).permute(2, 3, 1, 0).reshape(9, 64, 64).contiguous().to(dtp).to(dev)
    dz = hip.conv3x3_fwd(_cl(dy).to(dtp).to(dev), w_dgrad, precise=(mode == "f32_precise"))
    assert _relerr(dz.float().cpu(), _cl(z64.grad)) < (5e-5 if mode == "f32_precise" else 1e-2)


def test_conv3x3_full_batch_shape_against_the_precise_f32_kernels():
    """BASELINE config-2 shape (B = 64, 256 x 256 pixels): the persistent 32-round tile loops with the XCD-aware tile order of the
    bf16 kernels (forward with BatchNorm prologue + fused statistics, data gradient with fused BatchNorm-backward sums, weight
    gradient) against the three-pass f32 kernels on the same bf16-representable data - every pixel of four sampled images for the
    forward / data-gradient launches, the full weight gradient over all 64 images."""
    from sar_ssl_amd import hip
    from conftest import check
    dev = _dev()
    B, F, T = 64, 256, 256
    g = torch.Generator(device="cuda").manual_seed(4242)
    x16 = torch.randn((B, F, T, 64), generator=g, device=dev).to(torch.bfloat16)
    dy16 = torch.randn((B, F, T, 64), generator=g, device=dev).to(torch.bfloat16)
    w16 = (torch.randn((9, 64, 64), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    sc = torch.rand(64, generator=g, device=dev) + 0.5
    sh = torch.randn(64, generator=g, device=dev) * 0.3
    aff = torch.stack([sc, sh, torch.zeros_like(sc), torch.ones_like(sc)]).contiguous()
    pick = [0, 21, 42, 63]
    x32s, dy32s, w32 = x16[pick].float().contiguous(), dy16[pick].float().contiguous(), w16.float()
    # forward: BN + ReLU prologue, fused output statistics
    out, sums = hip.conv3x3_fwd(x16, w16, sc, sh, want_stats=True)
    zs = torch.relu(x32s * sc + sh).to(torch.bfloat16).float()          # the bf16 kernel rounds the prologue output before the MFMA
    ref = hip.conv3x3_fwd(zs, w32, precise=True)
    check("conv_full.fwd", _relerr(out[pick].float(), ref.cpu()), 1e-2)
    o64 = out.float().reshape(-1, 64).double()
    check("conv_full.fwd_stats", max(_relerr(sums[:64], o64.sum(0).cpu()), _relerr(sums[64:], (o64 ** 2).sum(0).cpu())), 1e-5)
    # data gradient, plain and with the fused BatchNorm-backward sums
    dz = hip.conv3x3_fwd(dy16, w16)
    refd = hip.conv3x3_fwd(dy32s, w32, precise=True)
    check("conv_full.dgrad", _relerr(dz[pick].float(), refd.cpu()), 1e-2)
    dz2, red = hip.conv3x3_dgrad_bnred(dy16, w16, x16, aff)
    assert red is not None and torch.equal(dz2, dz)
    red_ref = hip.cl_bn_bwd_reduce(dz, x16, 64, aff, 1)
    check("conv_full.dgrad_bnred_sums", _relerr(red, red_ref.cpu()), 1e-5)
    # weight gradient over all 64 images (identity prologue: no rounding difference between the two paths)
    dW = hip.conv3x3_wgrad(dy16, x16)
    dWref = hip.conv3x3_wgrad(dy16.float(), x16.float(), precise=True)
    check("conv_full.wgrad", _relerr(dW, dWref.cpu()), 1e-4)
    dWp = hip.conv3x3_wgrad(dy16, x16, sc, sh)
    zfull = torch.relu(x16.float() * sc + sh).to(torch.bfloat16).float()
    dWpref = hip.conv3x3_wgrad(dy16.float(), zfull, precise=True)
    check("conv_full.wgrad_prologue", _relerr(dWp, dWpref.cpu()), 1e-4)


def test_first_layer_conv_variants_at_the_full_batch_shape():
    """BASELINE config-2 shape (B = 64, 256 x 256): the three kernels that only exist in the timed bf16 configuration - the first 3x3
    convolution formed from the 4-channel input (C1IN forward, C1IN weight gradient) and the data gradient consumed in its epilogue
    (C1RED) - through the full persistent tile loops (32+ rounds, XCD-aware tile order, the DEFAULT 224-CU grids of the gradient
    launches) against the precise three-pass f32 kernels / the stored path: every pixel of four sampled images for the forward, the
    full weight gradient and the full first-layer parameter gradients over all 64 images."""
    from sar_ssl_amd import hip
    from conftest import check
    dev = _dev()
    B, F, T = 64, 256, 256
    g = torch.Generator(device="cuda").manual_seed(777)
    a0 = torch.randn((B, F, T, 4), generator=g, device=dev).to(torch.bfloat16)
    W1 = torch.randn((64, 4), generator=g, device=dev) * 0.5
    scale = torch.rand(64, generator=g, device=dev) + 0.5
    scale[7] = -scale[7]
    shift = torch.randn(64, generator=g, device=dev) * 0.3
    w16 = (torch.randn((9, 64, 64), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    dy16 = torch.randn((B, F, T, 64), generator=g, device=dev).to(torch.bfloat16)
    npix = B * F * T
    pick = [0, 21, 42, 63]
    hip.sums_arena_reset(dev)
    # the operand the kernels form while staging, restated in f32 (rounded to bf16 like the staged operand)
    z = torch.relu((a0.float().view(-1, 4) @ W1.t()) * scale + shift).to(torch.bfloat16).view(B, F, T, 64)
    # ---- forward from the 4-channel input
    out, stats = hip.conv3x3_fwd_c1(a0, W1, scale, shift, w16, want_stats=True)
    ref = hip.conv3x3_fwd(z[pick].float().contiguous(), w16.float(), precise=True)
    check("c1_full.fwd", _relerr(out[pick].float(), ref), 1e-2)                      # bf16 output rounding (4e-3) + rare 1-ulp operand flips
    o64 = out.float().reshape(-1, 64).double()
    check("c1_full.fwd_stats", max(_relerr(stats[:64], o64.sum(0)), _relerr(stats[64:], (o64 ** 2).sum(0))), 1e-5)
    del o64
    out_stored = hip.conv3x3_fwd(z, w16)                                             # stored-path kernel on the same operand
    check("c1_full.fwd_vs_stored_kernel", _relerr(out.float(), out_stored.float()), 1e-2)
    del out_stored, out
    # ---- weight gradient with the operand formed from a0 (default gradient grid)
    gW = torch.zeros((64, 64, 3, 3), device=dev)
    assert hip.conv3x3_wgrad_c1(dy16, a0, W1, scale, shift, gW)
    dWref = hip.conv3x3_wgrad(dy16.float(), z.float(), precise=True)                # [9][co][ci]
    check("c1_full.wgrad", _relerr(gW.permute(2, 3, 0, 1).reshape(9, 64, 64), dWref), 1e-3)
    del dWref
    # ---- data gradient consumed in the epilogue: first-layer parameter gradients
    sums, mom = hip.stem_c1_stats(a0, W1, keep_moments=True)
    mean = (sums[:64] / npix).float()
    var = (sums[64:] / npix - (sums[:64] / npix) ** 2).clamp_min(0).float()
    aff = torch.stack([scale, shift, mean, 1.0 / torch.sqrt(var + 1e-5)]).contiguous()
    # reference: the UNROUNDED data gradient (three-pass f32 kernel over all 64 images) reduced in f64.  Both device paths round the
    # masked gradient to bf16 before they reduce it (the stored path when it stores dz1, the fused path when it hands the masked tile
    # to the matrix cores), and on random data the reduced sums are random-walk sums - of the size of their own rounding noise times
    # sqrt(N) - so a relative deviation of the order of the bf16 epsilon is what BOTH must show; the gate is that the fused path is
    # in the same class as the stored one (measured: 2.1e-3 vs 1.2e-3 on dW1) and within 5e-3.
    dz1_32 = hip.conv3x3_fwd(dy16.float(), w16.float(), precise=True).view(-1, 64)
    y = a0.float().view(-1, 4) @ W1.t()
    mask = (y * scale + shift) > 0
    xh = ((y - aff[2]) * aff[3]).double()
    gg = torch.where(mask, dz1_32, torch.zeros_like(dz1_32)).double()
    del dz1_32, y, mask
    s1, s2 = gg.sum(0), (gg * xh).sum(0)
    a0d = a0.double().view(-1, 4)
    G = gg.t() @ a0d                                                                  # sum_p g a0
    X = xh.t() @ a0d                                                                  # sum_p xhat a0
    Sa = a0d.sum(0)
    del gg, xh
    dz1 = hip.conv3x3_fwd(dy16, w16)                                                 # stored path: dz1 (bf16), then the one-pass backward
    for train in (True, False):
        ref = scale.double()[:, None] * ((G - s1[:, None] * Sa[None, :] / npix - s2[:, None] * X / npix) if train else G)
        dW, dga, dbe = torch.zeros((64, 4, 1, 1), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        assert hip.conv3x3_dgrad_c1red(dy16, w16, a0, W1, aff, mom, train, dW, dga, dbe)
        dW2, dga2, dbe2 = torch.zeros((64, 4, 1, 1), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        hip.stem_c1_bwd_a0(dz1, a0, W1, aff, train, dW2, dga2, dbe2)
        for name, got, got2, want in (("dW1", dW.view(64, 4), dW2.view(64, 4), ref), ("dgamma", dga, dga2, s2), ("dbeta", dbe, dbe2, s1)):
            e_fused, e_stored = _relerr(got, want), _relerr(got2, want)
            check("c1_full.c1red.%s[train=%d]" % (name, train), e_fused, 5e-3)
            check("c1_full.c1red.%s[train=%d].stored_path" % (name, train), e_stored, 5e-3)
            assert e_fused <= 3.0 * e_stored + 1e-3, (name, e_fused, e_stored)      # (same rounding class; the max over 64-256 random-walk sums varies ~2x)
    del dz1


def test_statistics_epilogues_and_bias_gradients_are_run_to_run_reproducible():
    """Round 3: the BatchNorm statistics / backward-sum epilogues of the convolution kernels and the bias-gradient column sums fold
    their partial sums in a fixed order (round 2: f32 atomics in arrival order).  Same launch twice -> identical bits."""
    from sar_ssl_amd import hip
    dev = _dev()
    B, F, T = 16, 256, 256
    g = torch.Generator(device="cuda").manual_seed(99)
    x16 = torch.randn((B, F, T, 64), generator=g, device=dev).to(torch.bfloat16)
    dy16 = torch.randn((B, F, T, 64), generator=g, device=dev).to(torch.bfloat16)
    w16 = (torch.randn((9, 64, 64), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    sc = torch.rand(64, generator=g, device=dev) + 0.5
    sh = torch.randn(64, generator=g, device=dev) * 0.3
    aff = torch.stack([sc, sh, torch.zeros_like(sc), torch.ones_like(sc)]).contiguous()
    a0 = torch.randn((B, F, T, 4), generator=g, device=dev).to(torch.bfloat16)
    W1 = torch.randn((64, 4), generator=g, device=dev) * 0.5

    def once():
        hip.sums_arena_reset(dev)
        _, s1 = hip.conv3x3_fwd(x16, w16, sc, sh, want_stats=True)
        _, r1 = hip.conv3x3_dgrad_bnred(dy16, w16, x16, aff)
        _, s2 = hip.conv3x3_fwd_c1(a0, W1, sc, sh, w16, want_stats=True)
        sums, mom = hip.stem_c1_stats(a0, W1, keep_moments=True)
        dW, dga, dbe = torch.zeros((64, 4, 1, 1), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        hip.conv3x3_dgrad_c1red(dy16, w16, a0, W1, aff, mom, True, dW, dga, dbe)
        bias = torch.zeros(64, device=dev)
        hip.colsum(dy16.view(-1, 64), bias)
        return [t.clone() for t in (s1, r1, s2, dW, dga, dbe, bias)]

    first = once()
    for _ in range(3):
        again = once()
        for a, b in zip(first, again):
            assert torch.equal(a, b)
    ref = dy16.view(-1, 64).double().sum(0)
    assert _relerr(first[-1], ref) < 1e-5


def test_gradient_convolutions_do_not_depend_on_the_workgroup_count():
    """sarssl_conv_cus_override (model.py: every CU for the stem that runs last, 7/8 for the other) only changes which persistent
    workgroup takes which tile: the stored data gradient is bit-identical, the per-workgroup partial sums of the weight gradient and of
    the BatchNorm-backward epilogue agree to f32 summation order."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    B, F, T = 6, 64, 96                                       # 1152 tiles of 8 x 32: several rounds on any grid
    dy = _cl(torch.randn((B, 64, F, T), generator=g)).to(torch.bfloat16).to(dev)
    y = _cl(torch.randn((B, 64, F, T), generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn((9, 64, 64), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    sc, sh = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    aff = torch.stack([sc, sh, torch.zeros(64, device=dev), torch.ones(64, device=dev)]).contiguous()
    out = {}
    try:
        for ncus in (0, 1 << 16, 40):                         # the default rule, every CU, an odd small grid
            hip.conv_cus_override(ncus)
            dz, red = hip.conv3x3_dgrad_bnred(dy, w, y, aff)
            gw = torch.zeros((64, 64, 3, 3), device=dev)
            hip.conv3x3_wgrad(dy, y, sc, sh, acc_into=gw)
            out[ncus] = (dz, red.clone(), gw)
    finally:
        hip.conv_cus_override(0)
    for ncus in (1 << 16, 40):
        assert torch.equal(out[ncus][0], out[0][0])
        assert _relerr(out[ncus][1], out[0][1]) < 1e-5
        assert _relerr(out[ncus][2], out[0][2]) < 1e-5


def test_two_contexts_on_one_device_do_not_interfere():
    """SURVEY.md 8(b) / round-3 verdict item 8: everything a caller configures lives in a sarssl_ctx (include/sarssl_hip.h) and kernels
    run under the context current on the calling thread - no process-global setters.  Two contexts on this device: the gradient-launch
    workgroup count and the attached step state (dropout salt) set on one are invisible under the other."""
    from sar_ssl_amd import hip, _lib
    dev = _dev()
    lib = _lib.lib()
    c1, c2 = _lib.ctx(dev.index), _lib.new_ctx(dev.index)
    get = lambda c: int(lib.sarssl_ctx_get_conv_cus(_lib.c_void_p(c)))
    st = hip.step_state_new(dev, 12345, 1e-3)
    A, B = _mk((256, 128), torch.bfloat16, dev, 1), _mk((128, 128), torch.bfloat16, dev, 2)
    drop = lambda: hip.gemm(A, B, M=256, N=128, K=128, lda=128, ldb=128, p_drop=0.5, seed=77).float().clone()
    try:
        assert c1 != c2 and int(lib.sarssl_ctx_device(_lib.c_void_p(c2))) == dev.index
        hip.conv_cus_override(64)
        y_plain = drop()
        with _lib.use_ctx(c2):
            assert get(c2) == 0
            hip.conv_cus_override(32)
            hip.step_state_attach(st)                   # only launches under c2 add the state's salt to their dropout seeds
            y_salted = drop()
        assert get(c1) == 64 and get(c2) == 32
        assert torch.equal(drop(), y_plain) and not torch.equal(y_salted, y_plain)
        with _lib.use_ctx(c2):
            assert torch.equal(drop(), y_salted)
            hip.step_state_attach(None)
            assert torch.equal(drop(), y_plain)
    finally:
        hip.conv_cus_override(0)
        _lib.destroy_ctx(c2)


@pytest.mark.parametrize("B,F,T", [(2, 16, 8), (1, 24, 136), (3, 8, 64)])
def test_conv3x3_dgrad_with_fused_bn_backward_sums(B, F, T):
    """The data-gradient launch that also accumulates the BatchNorm-backward sums of the layer in front must store exactly the
    same dz as the plain launch, and its sums must equal the stand-alone cl_bn_bwd_reduce pass over (dz, y)."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(11 * B + T)
    dy = _cl(torch.randn((B, 64, F, T), generator=g)).to(torch.bfloat16).to(dev)
    y = _cl(torch.randn((B, 64, F, T), generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn((9, 64, 64), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    aff = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1,
                       torch.rand(64, generator=g) + 0.5]).contiguous().to(dev)
    aff[0, 9] = -aff[0, 9]
    dz_ref = hip.conv3x3_fwd(dy, w)
    red_ref = hip.cl_bn_bwd_reduce(dz_ref, y, 64, aff, 1)
    dz, red = hip.conv3x3_dgrad_bnred(dy, w, y, aff)
    assert torch.equal(dz, dz_ref)
    assert _relerr(red[:64], red_ref[:64]) < 1e-5 and _relerr(red[64:], red_ref[64:]) < 2e-5


@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("shape", [(2, 24, 40), (3, 32, 48)])      # ragged tiles (general kernel) / full 16 x 16 tiles (bf16: stem_c4_bwd_sums16_kernel)
def test_stem_c4_backward_two_phase_matches_one_pass_plus_apply(dtp, train, shape):
    """sums-only pass + direct dy3 pass == (g3 + sums) pass followed by the in-place BatchNorm-backward normalisation."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    B, F, T = shape
    y3 = torch.randn((B, F, T, 64), generator=g).to(dtp).to(dev)
    dy4 = torch.randn((B, T, F, 4), generator=g).to(dtp).to(dev)
    W4 = (torch.randn((4, 64), generator=g) * 0.2).to(dev)
    aff = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1,
                       torch.rand(64, generator=g) + 0.5]).contiguous().to(dev)
    g3, red_ref = hip.stem_c4_bwd(y3, dy4, W4, aff)
    dy3_ref = hip.cl_bn_bwd_apply(g3, y3, 64, aff, 1, True, train, red_ref[256:])
    dy3, red = hip.stem_c4_bwd_two_phase(y3, dy4, W4, aff, train)
    for a, b in ((0, 256), (256, 320), (320, 384)):           # dW4 | sum g | sum g*xhat (the 16-bit kernel forms the last from raw moments)
        assert _relerr(red[a:b], red_ref[a:b]) < 2e-6
    assert _relerr(dy3.float(), dy3_ref.float()) < (1e-5 if dtp == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("train", [True, False])
def test_stem_c1_wgrad_with_fused_bn_backward(dtp, train):
    """dW1 from (dz1, y1) with the BatchNorm-backward normalisation in registers == normalise (cl_bn_bwd_apply) then stem_c1_wgrad."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(33)
    B, F, T = 2, 16, 24
    dz1 = torch.randn((B, F, T, 64), generator=g).to(dtp).to(dev)
    y1 = torch.randn((B, F, T, 64), generator=g).to(dtp).to(dev)
    a0 = torch.randn((B, F, T, 4), generator=g).to(dtp).to(dev)
    aff = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1,
                       torch.rand(64, generator=g) + 0.5]).contiguous().to(dev)
    red = hip.cl_bn_bwd_reduce(dz1, y1, 64, aff, 1)
    dy1 = hip.cl_bn_bwd_apply(dz1, y1, 64, aff, 1, False, train, red)
    ref = torch.zeros((64, 4, 1, 1), device=dev)
    hip.stem_c1_wgrad(dy1, a0, ref)
    got = torch.zeros((64, 4, 1, 1), device=dev)
    hip.stem_c1_wgrad_bn(dz1, y1, a0, aff, red, train, got)
    assert _relerr(got, ref) < (1e-5 if dtp == torch.float32 else 1e-2)      # (bf16: the reference path rounds dy1 to bf16 in between)


@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("train", [True, False])
def test_stem_first_layer_backward_in_one_pass(dtp, train):
    """stem_c1_bwd (BatchNorm sums, dgamma/dbeta and dW1 from one pass, combined algebraically) == reduce + normalise + weight gradient."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(35)
    B, F, T = 2, 16, 24
    dz1 = torch.randn((B, F, T, 64), generator=g).to(dtp).to(dev)
    y1 = (torch.randn((B, F, T, 64), generator=g) * 0.7 + 0.2).to(dtp).to(dev)
    a0 = torch.randn((B, F, T, 4), generator=g).to(dtp).to(dev)
    aff = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1 + 0.2,
                       torch.rand(64, generator=g) + 0.5]).contiguous().to(dev)
    aff[0, 5] = -aff[0, 5]                                        # a negative BatchNorm scale flips the ReLU threshold test
    red = hip.cl_bn_bwd_reduce(dz1, y1, 64, aff, 1)
    dy1 = hip.cl_bn_bwd_apply(dz1.float(), y1.float(), 64, aff, 1, False, train, red)      # f32 reference normalisation
    ref = torch.zeros((64, 4, 1, 1), device=dev)
    hip.stem_c1_wgrad(dy1, a0.float(), ref)
    dW, dga, dbe = torch.zeros((64, 4, 1, 1), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    hip.stem_c1_bwd(dz1, y1, a0, aff, train, dW, dga, dbe)
    assert _relerr(dW, ref) < 2e-5
    assert _relerr(dbe, red[:64].float()) < 1e-5 and _relerr(dga, red[64:].float()) < 1e-5
    hip.stem_c1_bwd(dz1, y1, a0, aff, train, dW, dga, dbe)          # accumulates
    assert _relerr(dW, 2 * ref) < 2e-5


@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
def test_stem_pointwise_and_bn(dtp):
    from sar_ssl_amd import hip
    dev = _dev()
    tol = 1e-5 if dtp == torch.float32 else 2e-2
    g = torch.Generator().manual_seed(3)
    B, F, T = 2, 16, 8
    # mask_inputs vs the oracle's formulas
    x = torch.randn((B, 2, F, T, 2), generator=g)
    idx, ch = orc.gen_masks(B, T, T // 2, 2)
    mp, mc = orc.dense_masks(idx, ch, T, 2)
    v = x.permute(0, 3, 2, 4, 1)
    mp5, mc5 = mp.view(B, T, 1, 1, 1), mc.view(B, 1, 1, 1, 2)
    spec_ref = (v * (1 - mp5) * mc5 + v * mp5 * (1 - mc5)).permute(0, 2, 1, 3, 4).reshape(B, F, T, 4)
    spat_ref = (v * mp5).permute(0, 2, 1, 3, 4).reshape(B, F, T, 4)
    spec, spat = hip.mask_inputs(x.to(dev), mp.to(torch.uint8).to(dev), ch.to(torch.int32).to(dev), 0, dtp)
    assert _relerr(spec.float(), spec_ref) < tol and _relerr(spat.float(), spat_ref) < tol
    s2, t2 = hip.mask_inputs(x.to(dev), mp.to(torch.uint8).to(dev), ch.to(torch.int32).to(dev), 1, dtp)
    assert _relerr(s2.float(), v.permute(0, 2, 1, 3, 4).reshape(B, F, T, 4)) < tol and torch.equal(s2, t2)
    # c1 fwd / wgrad
    a0 = torch.randn((B, F, T, 4), generator=g).to(dtp)
    W1 = torch.randn((64, 4), generator=g)
    y1, s1k = hip.stem_c1_fwd(a0.to(dev), W1.to(dev), want_stats=True)
    assert _relerr(s1k[:64], y1.float().reshape(-1, 64).double().sum(0)) < 1e-5 and _relerr(s1k[64:], (y1.float().reshape(-1, 64).double() ** 2).sum(0)) < 1e-5
    assert _relerr(y1.float(), a0.float() @ W1.t()) < tol
    dy1 = torch.randn((B, F, T, 64), generator=g).to(dtp)
    gW1 = torch.zeros((64, 4), device=dev)
    hip.stem_c1_wgrad(dy1.to(dev), a0.to(dev), gW1)
    assert _relerr(gW1, dy1.float().reshape(-1, 64).t() @ a0.float().reshape(-1, 4)) < 1e-4
    # batch-norm statistics / finalize / running stats
    y = (torch.randn((B, F, T, 64), generator=g) * 2 + 0.7).to(dtp)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    rm, rv = torch.zeros(64), torch.ones(64)
    rm_d, rv_d, nbt = rm.to(dev), rv.to(dev), torch.zeros((), dtype=torch.int64, device=dev)
    aff = hip.bn_train_affine(y.to(dev), 64, gamma.to(dev), beta.to(dev), rm_d, rv_d, nbt)
    yf = y.float().reshape(-1, 64).double()
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    assert _relerr(aff[2], mean) < 1e-5 and _relerr(aff[3], (var + 1e-5).rsqrt()) < 1e-5
    assert _relerr(rm_d, 0.1 * mean) < 1e-5 and _relerr(rv_d, 0.9 + 0.1 * yf.var(0, unbiased=True)) < 1e-5
    assert int(nbt) == 1
    z = hip.cl_affine_act(y.to(dev), 64, aff, 1)
    zref = torch.relu((yf - mean) / (var + 1e-5).sqrt() * gamma.double() + beta.double())
    assert _relerr(z.float().reshape(-1, 64), zref) < tol
    # c4 fwd (BN+ReLU prologue, (B,T,F,4) output) and its fused backward
    W4 = torch.randn((4, 64), generator=g) * 0.2
    y4 = hip.stem_c4_fwd(y.to(dev), W4.to(dev), aff[0], aff[1])
    zq = zref.float()
    y4ref = (zq @ W4.t()).reshape(B, F, T, 4).permute(0, 2, 1, 3)
    assert _relerr(y4.float(), y4ref) < tol
    dy4 = torch.randn((B, T, F, 4), generator=g).to(dtp)
    g3, red = hip.stem_c4_bwd(y.to(dev), dy4.to(dev), W4.to(dev), aff)
    d4 = dy4.float().permute(0, 2, 1, 3).reshape(-1, 4).double()
    g3ref = (d4 @ W4.double()) * (zref > 0)
    assert _relerr(g3.float().reshape(-1, 64), g3ref) < tol
    assert _relerr(red[:256].reshape(4, 64), d4.t() @ zref) < 1e-4
    xhat = (yf - mean) / (var + 1e-5).sqrt()
    assert _relerr(red[256:320], g3ref.sum(0)) < 1e-4 and _relerr(red[320:], (g3ref * xhat).sum(0)) < 1e-4
    # generic BN backward (relu) against autograd
    yy = yf.clone().requires_grad_(True)
    zz = torch.relu(torch.nn.functional.batch_norm(yy, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5))
    dz = torch.randn((B * F * T, 64), generator=g).to(dtp)
    zz.backward(dz.double())
    red2 = hip.cl_bn_bwd_reduce(dz.to(dev), y.to(dev), 64, aff, 1)
    dyk = hip.cl_bn_bwd_apply(dz.to(dev), y.to(dev), 64, aff, 1, False, True, red2)
    assert _relerr(dyk.float().reshape(-1, 64), yy.grad) < tol
    # C = 4 path (folded 64-wide view)
    y4t = torch.randn((B, T, F, 4), generator=g).to(dtp)
    sums, N = hip.cl_stats(y4t.to(dev), 4)
    assert N == B * T * F
    assert _relerr(sums[:4], y4t.float().reshape(-1, 4).double().sum(0)) < 1e-5


# ---------------------------------------------------------------- row / elementwise kernels
@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d", [32, 256, 512])
def test_layernorm(dtp, d):
    from sar_ssl_amd import hip
    dev = _dev()
    tol = 2e-5 if dtp == torch.float32 else 2e-2
    g = torch.Generator().manual_seed(d)
    M = 70
    x = (torch.randn((M, d), generator=g) * 1.5 + 0.3).to(dtp)
    gamma, beta = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    dy = torch.randn((M, d), generator=g).to(dtp)
    res = torch.randn((M, d), generator=g).to(dtp)
    x64 = x.double().requires_grad_(True); g64 = gamma.double().requires_grad_(True); b64 = beta.double().requires_grad_(True)
    yref = torch.nn.functional.layer_norm(x64, (d,), g64, b64, 1e-5)
    yref.backward(dy.double())
    y, stats = hip.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev))
    assert _relerr(y.float(), yref.detach()) < tol
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dx = hip.layernorm_bwd(dy.to(dev), x.to(dev), gamma.to(dev), stats, resid=res.to(dev), dgamma=dg, dbeta=db)
    assert _relerr(dx.float(), x64.grad + res.double()) < tol
    assert _relerr(dg, g64.grad) < 1e-4 and _relerr(db, b64.grad) < 1e-4
    # strided output (writes into a wider concat buffer)
    wide = torch.zeros((M, d + 64), dtype=dtp, device=dev)
    hip.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev), out=wide[:, 64:], save=False)
    assert _relerr(wide[:, 64:].float(), yref.detach()) < tol and float(wide[:, :64].abs().max()) == 0.0


@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("d", [256, 512])
def test_layernorm_several_rows_per_wave_is_bit_identical(dtp, d):
    """Large row counts take layernorm_fwd_rows_kernel (four / two rows per wave, a row's reductions inside 16 / 32 lanes): the same
    summation order as the one-row-per-wave kernel that shorter inputs take - outputs and saved statistics equal bit for bit, also for a
    row count that is no multiple of the rows per wave, strided input and output; and against the f64 LayerNorm."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(d + 1)
    M = 8191
    xw = (torch.randn((M, d + 64), generator=g) * 1.7 + 0.4).to(dtp).to(dev)
    x = xw[:, 32:32 + d]                                          # row stride d + 64
    gamma, beta = (torch.rand(d, generator=g) + 0.5).to(dev), torch.randn(d, generator=g).to(dev)
    wide = torch.zeros((M, d + 8), dtype=dtp, device=dev)
    y, st = hip.layernorm_fwd(x, gamma, beta, out=wide[:, 8:])
    parts = [hip.layernorm_fwd(x[a:b], gamma, beta) for a, b in ((0, 3000), (3000, 6000), (6000, M))]       # < 4096 rows: one row per wave
    assert torch.equal(y, torch.cat([p[0] for p in parts])) and torch.equal(st, torch.cat([p[1] for p in parts], dim=1))
    assert float(wide[:, :8].abs().max()) == 0.0
    ref = torch.nn.functional.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-5)
    assert _relerr(y.float(), ref) < (2e-5 if dtp == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
def test_glu_dwconv_misc(dtp):
    from sar_ssl_amd import hip
    dev = _dev()
    tol = 2e-5 if dtp == torch.float32 else 2e-2
    g = torch.Generator().manual_seed(11)
    B, T, d = 2, 40, 64
    h = torch.randn((B * T, 2 * d), generator=g).to(dtp)
    dg = torch.randn((B * T, d), generator=g).to(dtp)
    h64 = h.double().requires_grad_(True)
    gref = h64[:, :d] * torch.sigmoid(h64[:, d:])
    gref.backward(dg.double())
    assert _relerr(hip.glu_fwd(h.to(dev)).float(), gref.detach()) < tol
    assert _relerr(hip.glu_bwd(dg.to(dev), h.to(dev)).float(), h64.grad) < tol
    # depthwise conv k=31 (T both longer and shorter than the kernel)
    for TT in (40, 16):
        x = torch.randn((B, TT, d), generator=g).to(dtp)
        w = torch.randn((d, 31), generator=g) * 0.2
        dy = torch.randn((B, TT, d), generator=g).to(dtp)
        x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
        yref = torch.nn.functional.conv1d(x64.transpose(1, 2), w64.unsqueeze(1), None, padding=15, groups=d).transpose(1, 2)
        yref.backward(dy.double())
        assert _relerr(hip.dwconv(x.to(dev), w.to(dev)).float(), yref.detach()) < tol
        assert _relerr(hip.dwconv(dy.to(dev), w.to(dev), flip=True).float(), x64.grad) < tol
        dw = torch.zeros((d, 31), device=dev)
        hip.dwconv_wgrad(dy.to(dev), x.to(dev), dw)
        assert _relerr(dw, w64.grad) < 1e-4
    # bias2 / axpby / colsum / act_bwd / cast
    q = torch.randn((50, d), generator=g).to(dtp)
    u, v = torch.randn(d, generator=g), torch.randn(d, generator=g)
    qu, qv = hip.bias2(q.to(dev), u.to(dev), v.to(dev))
    assert _relerr(qu.float(), q.float() + u) < tol and _relerr(qv.float(), q.float() + v) < tol
    assert _relerr(hip.axpby(q.to(dev), qu, 0.5, 2.0).float(), 0.5 * q.float() + 2 * qu.float().cpu()) < tol
    cs = torch.zeros(d, device=dev)
    hip.colsum(q.to(dev), cs)
    assert _relerr(cs, q.float().sum(0)) < 1e-4
    hpre = torch.randn((50, d), generator=g).to(dtp)
    h64 = hpre.double().requires_grad_(True)
    (h64 * torch.sigmoid(h64)).backward(q.double())
    assert _relerr(hip.act_bwd(q.to(dev), hpre.to(dev), 2).float(), h64.grad) < tol
    assert _relerr(hip.act_bwd(q.to(dev), hpre.to(dev), 1, gscale=0.5).float(), 0.5 * q.float() * (hpre.float() > 0)) < tol
    assert torch.equal(hip.cast(u.to(dev), torch.bfloat16).cpu(), u.to(torch.bfloat16))


@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T", [16, 100, 256])
def test_softmax_relshift(dtp, T):
    from sar_ssl_amd import hip
    dev = _dev()
    tol = 2e-5 if dtp == torch.float32 else 1e-2
    g = torch.Generator().manual_seed(T)
    nmat = 6
    content = torch.randn((nmat, T, T), generator=g) * 3
    pos = torch.randn((nmat, T, T), generator=g) * 3
    c64 = content.double().requires_grad_(True); p64 = pos.double().requires_grad_(True)
    score = (c64 + orc.relative_shift(p64)) * 0.17
    pref = torch.softmax(score, -1)
    dpd = torch.randn((nmat, T, T), generator=g)
    pref.backward(dpd.double())
    p, pd = hip.softmax_relshift_fwd(content.to(dev), pos.to(dev), 0.17, dtp)
    assert pd is p and _relerr(p.float(), pref.detach()) < tol
    ds = hip.softmax_bwd(dpd.to(dev), p, 0.17)
    assert _relerr(ds.float(), c64.grad) < tol
    dpos = hip.relshift_bwd(ds)
    assert _relerr(dpos.float(), p64.grad) < tol
    # dropout: mask consistent between forward and backward
    p2, pd2 = hip.softmax_relshift_fwd(content.to(dev), pos.to(dev), 0.17, torch.float32, p_drop=0.1, seed=99)
    keep = (pd2 != 0)
    assert 0.85 < keep.float().mean().item() < 0.95
    assert _relerr(pd2[keep], (p2 / 0.9)[keep]) < 1e-5
    ds2 = hip.softmax_bwd(dpd.to(dev), p2, 0.17, p_drop=0.1, seed=99)
    dp = dpd.double() * keep.cpu().double() / 0.9
    pr = p2.double().cpu()
    assert _relerr(ds2, 0.17 * pr * (dp - (dp * pr).sum(-1, keepdim=True))) < 1e-4


@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,d", [(1000, 256), (333, 512), (64, 32), (50, 1024)])
def test_two_layernorms_in_one_launch(M, d, dtp):
    """`sarssl_layernorm_fwd2` (a block's closing LayerNorm + the next block's first): both outputs and both statistics bit for bit
    what two `sarssl_layernorm_fwd` launches produce."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(M + d)
    x = (torch.randn((M, d), generator=g) * 1.7 - 0.4).to(dtp).to(dev)
    ga, ba, gb, bb = [(torch.randn(d, generator=g) * 0.3 + (1.0 if i % 2 == 0 else 0.0)).to(dev) for i in range(4)]
    y1, s1 = hip.layernorm_fwd(x, ga, ba, 1e-5)
    z1, t1 = hip.layernorm_fwd(y1, gb, bb, 1e-5)
    y2, s2, z2, t2 = hip.layernorm_fwd2(x, ga, ba, 1e-5, gb, bb, 1e-5)
    assert torch.equal(y2, y1) and torch.equal(s2, s1) and torch.equal(z2, z1) and torch.equal(t2, t1)


@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,C", [(1000, 256), (4096, 512), (77, 64)])
def test_batchnorm_finalize_affine_activation_in_one_launch(rows, C, dtp):
    """`sarssl_cl_bn_train_act` (the convolution module's BatchNorm + Swish, conformer/convolution.py:141-142): affine rows, running
    statistics, batch counter and the activated output - bit for bit what `sarssl_bn_finalize` + `sarssl_cl_affine_act` produce."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn((rows, C), generator=g) * 1.5 + 0.3).to(dtp).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    sums, N = hip.cl_stats(x, C)
    st = lambda: (torch.linspace(-1, 1, C).to(dev), torch.linspace(0.5, 2, C).to(dev), torch.tensor(7, device=dev))
    rm1, rv1, n1 = st()
    aff1 = hip.bn_train_affine(None, C, gamma, beta, rm1, rv1, n1, sums=sums, N=N)
    z1 = hip.cl_affine_act(x, C, aff1, 2)
    rm2, rv2, n2 = st()
    aff2, z2 = hip.cl_bn_train_act(x, C, gamma, beta, rm2, rv2, n2, sums, 2)
    assert torch.equal(aff2, aff1) and torch.equal(z2, z1) and torch.equal(rm2, rm1) and torch.equal(rv2, rv1) and int(n2) == int(n1) == 8
    assert float(z2.float().abs().max()) > 0


@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
def test_masked_mse_and_adam(dtp):
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, F, T = 3, 16, 8
    x = torch.randn((B, 2, F, T, 2), generator=g)
    pred = torch.randn((B, T, F * 4), generator=g).to(dtp)
    idx, ch = orc.gen_masks(B, T, T // 2, 2)
    mp, mc = orc.dense_masks(idx, ch, T, 2)
    v = x.permute(0, 3, 2, 4, 1)
    mc5 = mc.view(B, 1, 1, 1, 2)
    p64 = pred.double().reshape(B, T, F, 2, 2).requires_grad_(True)
    gi = idx.view(B, -1, 1, 1).expand(-1, -1, F, 2)
    pm = (p64 * (1 - mc5)).sum(-1).gather(1, gi)
    tm = (v.double() * (1 - mc5)).sum(-1).gather(1, gi)
    om = (v.double() * mc5).sum(-1).gather(1, gi)
    loss, diff = ((pm - tm) ** 2).mean(), ((tm - om) ** 2).mean()
    loss.backward()
    out = hip.masked_mse_fwd(pred.to(dev), x.to(dev), idx.to(torch.int32).to(dev), ch.to(torch.int32).to(dev))
    assert abs(out[0].item() / loss.item() - 1) < 1e-5 and abs(out[1].item() / diff.item() - 1) < 1e-5
    dpred = hip.masked_mse_bwd(pred.to(dev), x.to(dev), mp.to(torch.uint8).to(dev), ch.to(torch.int32).to(dev), T // 2)
    assert _relerr(dpred.float().reshape(B, T, F, 2, 2), p64.grad) < (1e-5 if dtp == torch.float32 else 1e-2)
    # loss and its gradient from ONE pass (the captured step): the two launches' results bit for bit, incl. the all-zero rows of
    # unmasked frames, ragged frame groups (T = 12: groups of 8) and fp16 predictions next to a bf16 gradient
    for dt2, T2 in ((dtp, T), (dtp, 12), (torch.float16, 12)):
        x2 = torch.randn((B, 2, F, T2, 2), generator=g).to(dev)
        pr2 = torch.randn((B, T2, F * 4), generator=g).to(dt2).to(dev)
        idx2, ch2 = orc.gen_masks(B, T2, T2 // 2, 2)
        mp2, _ = orc.dense_masks(idx2, ch2, T2, 2)
        i2, c2 = idx2.to(torch.int32).to(dev), ch2.to(torch.int32).to(dev)
        o_sep = hip.masked_mse_fwd(pr2, x2, i2, c2)
        d_sep = hip.masked_mse_bwd(pr2, x2, mp2.to(torch.uint8).to(dev), c2, T2 // 2)
        o_one, d_one = hip.masked_mse_fwd(pr2, x2, i2, c2, with_grad=True)
        assert torch.equal(o_one, o_sep) and d_one.dtype == d_sep.dtype and torch.equal(d_one, d_sep)
        assert float(d_one.float().abs().sum()) > 0
    if dtp == torch.float32:
        n = 1000
        p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g)
        pr = {"w": p.clone()}; st = {}
        pd_, m, vv = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
        for step in (1, 2, 3):
            orc.adam_step(pr, {"w": gr}, st, 1e-3)
            hip.adam_step(pd_, gr.to(dev), m, vv, p16, 1e-3, step)
        assert _relerr(pd_, pr["w"]) < 1e-6
        assert torch.equal(p16.cpu(), pd_.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("mode", ["bf16", "f32_precise"])
@pytest.mark.parametrize("split", [1, 3, 8])
def test_gemm_split_k_accumulate(mode, split):
    """dW += dY^T X with the contraction split over several workgroups and f32 atomic accumulation."""
    from sar_ssl_amd import hip
    dev = _dev()
    dtp = torch.bfloat16 if mode == "bf16" else torch.float32
    M, N, K = 1000, 136, 72                      # contraction length M
    dy = _mk((M, N), dtp, dev, 21); x = _mk((M, K), dtp, dev, 22)
    g0 = _mk((N, K), torch.float32, dev, 23)
    g = g0.clone()
    hip.gemm(dy, x, a_kc=False, b_kc=False, M=N, N=K, K=M, lda=N, ldb=K, out=g, ldc=K, precise=(mode == "f32_precise"), split_k=split)
    ref = g0.double() + dy.double().t() @ x.double()
    assert _relerr(g, ref) < (1e-5 if mode == "bf16" else 5e-5)


# ---------------------------------------------------------------- fused relative-position attention (csrc/attention.hip)
def _ref_shift(R):
    """RelativeMultiHeadAttention._relative_shift (code/common/conformer/attention.py:105-113) on (B,H,T,T)."""
    B, H, T, _ = R.shape
    padded = torch.cat([R.new_zeros(B, H, T, 1), R], dim=-1).view(B, H, T + 1, T)
    return padded[:, :, 1:].reshape(B, H, T, T)


def _attn_mask(B, H, T, dh, p_drop, seed, dev):
    """Keep-mask of the fused kernel's dropout (a pure function of seed and element index): uniform probabilities (q = 0, no bias)
    against one-hot value blocks expose the dropped probability matrix dh columns at a time."""
    from sar_ssl_amd import hip
    d = H * dh
    z = torch.zeros((B * T, d), dtype=torch.bfloat16, device=dev)
    zb = torch.zeros((B, H, T, T), dtype=torch.bfloat16, device=dev)
    mask = torch.zeros((B, H, T, T), dtype=torch.bool, device=dev)
    for t0 in range(0, T, dh):
        V = torch.zeros((B, T, H, dh), dtype=torch.bfloat16, device=dev)
        n = min(dh, T - t0)
        V[:, t0 + torch.arange(n), :, torch.arange(n)] = 1.0
        ctx, _ = hip.relpos_attn_fwd(z, z, V.view(B * T, d), zb, B, H, T, dh, 1.0, p_drop, seed, need_bwd=False)
        mask[:, :, :, t0:t0 + n] = (ctx.view(B, T, H, dh).permute(0, 2, 1, 3)[..., :n] != 0)
    return mask


@pytest.mark.parametrize("adt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("B,H,T,dh", [(2, 4, 256, 128), (3, 4, 40, 64), (1, 2, 624, 64), (2, 4, 64, 32), (1, 4, 136, 128),
                                      (2, 2, 8, 32), (1, 2, 128, 64), (2, 1, 248, 64), (1, 3, 200, 32), (2, 4, 256, 64),
                                      (1, 2, 624, 128), (2, 2, 264, 64), (1, 1, 1000, 64), (1, 2, 512, 128)])
def test_fused_relpos_attention_fwd_bwd(B, H, T, dh, p_drop, adt):
    """Positional-score GEMM written in the relative-shift layout + fused attention forward / backward against an f64 torch
    restatement of attention.py:87-113 on the same 16-bit operands (dropout mask read back from the kernel's own hash).  adt = dtype
    of the forward tensors: bf16, or fp16 with bf16 gradients (the fp16-forward mode)."""
    from sar_ssl_amd import hip
    from conftest import check
    dev = _dev()
    d = H * dh
    scale = 1.0 / (d ** 0.5)
    g = torch.Generator().manual_seed(B * 1000 + T + dh)
    mk = lambda *shape, s=1.0, dt_=adt: (torch.randn(shape, generator=g) * s).to(dt_).to(dev)
    qkv = mk(B * T, 3 * d)
    qu, k, v = mk(B * T, d), qkv[:, d:2 * d], qkv[:, 2 * d:]                 # k / v as column slices (row stride 3d), as in the engine
    qv, pos = mk(B * T, d), mk(T, d, s=2.0)
    dctx = mk(B * T, d, dt_=torch.bfloat16)
    ftol = 1e-2 if adt == torch.bfloat16 else 2e-3                           # forward quantities: fp16 operands / outputs are 8x finer
    # positional score in the shifted layout straight from the GEMM epilogue
    bias = torch.full((B, H, T, T), float("nan"), dtype=adt, device=dev)
    hip.gemm(qv, pos, M=T, N=T, K=dh, lda=d, ldb=d, nbatch=B * H, batch_inner=H, sA=(T * d, dh), sB=(0, dh), out=bias, ldc=T,
             sC=(H * T * T, T * T), c_row_shift=True)
    raw = torch.einsum("bihc,mhc->bhim", qv.view(B, T, H, dh).double(), pos.view(T, H, dh).double())
    want_bias = _ref_shift(raw)
    ii = torch.arange(T - 1, device=dev)
    got_bias = bias.double().clone()
    assert torch.isnan(got_bias[:, :, ii, ii + 1]).all()                     # the padding zeros are never written ...
    got_bias[:, :, ii, ii + 1] = 0.0                                         # ... and ignored by the attention kernels
    check("attn.bias_shift_gemm", _relerr(got_bias, want_bias), ftol)
    seed = 991
    ctx, aux = hip.relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, p_drop, seed)
    lse = aux[1]
    ctx2, _ = hip.relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, p_drop, seed)
    assert torch.equal(ctx, ctx2)
    if hip.relpos_attn_pos_supported(T, dh, adt):
        # the forward that forms the shifted positional score inside the kernel (T <= 256): the same MFMA products in the same
        # order as the GEMM, so the score tile, the output and the log-sum-exp are the two-launch path's bit for bit
        ctx_p, aux_p, bias_p = hip.relpos_attn_fwd_pos(qu, qv, k, v, pos, B, H, T, dh, scale, p_drop, seed)
        bp = bias_p.double()
        bp[:, :, ii, ii + 1] = 0.0
        check("attn.pos_in_kernel.bias", _relerr(bp, want_bias), ftol)
        assert torch.equal(bp, got_bias), (bp - got_bias).abs().max().item()
        assert torch.equal(ctx_p, ctx) and torch.equal(aux_p[0], aux[0]) and torch.equal(aux_p[1], aux[1])
        ctx_i, _, none_bias = hip.relpos_attn_fwd_pos(qu, qv, k, v, pos, B, H, T, dh, scale, p_drop, seed, need_bwd=False)
        assert none_bias is None and torch.equal(ctx_i, ctx)
    else:
        assert T > 256
    if hip.relpos_attn_pos_long_supported(T, dh, adt):
        # round 6, T > 256: the forward forms the shifted score per 256-key block of its slab - the GEMM's MFMA products in the GEMM's order:
        # score tensor, output and log-sum-exp bit for bit those of the two-launch path
        ctx_l, aux_l, bias_l = hip.relpos_attn_fwd_pos_long(qu, qv, k, v, pos, B, H, T, dh, scale, p_drop, seed)
        bl = bias_l.double()
        bl[:, :, ii, ii + 1] = 0.0
        assert torch.equal(bl, got_bias), (bl - got_bias).abs().max().item()
        assert torch.equal(ctx_l, ctx) and torch.equal(aux_l[0], aux[0]) and torch.equal(aux_l[1], aux[1])
        ctx_li, _, none_bias = hip.relpos_attn_fwd_pos_long(qu, qv, k, v, pos, B, H, T, dh, scale, p_drop, seed, need_bwd=False)
        assert none_bias is None and torch.equal(ctx_li, ctx)
    keep = torch.ones((B, H, T, T), dtype=torch.float64, device=dev)
    if p_drop > 0:
        mask = _attn_mask(B, H, T, dh, p_drop, seed, dev)
        frac = mask.double().mean().item()
        assert abs(frac - (1 - p_drop)) < 0.01, frac
        keep = mask.double() / (1 - p_drop)
    # f64 reference on the same bf16 operands (bias = what the kernel reads)
    q64 = qu.view(B, T, H, dh).double().permute(0, 2, 1, 3).requires_grad_(True)
    k64 = k.reshape(B, T, H, dh).double().permute(0, 2, 1, 3).requires_grad_(True)
    v64 = v.reshape(B, T, H, dh).double().permute(0, 2, 1, 3).requires_grad_(True)
    b64 = got_bias.clone().requires_grad_(True)
    A = torch.softmax((q64 @ k64.transpose(-1, -2) + b64) * scale, dim=-1)
    O = (A * keep) @ v64
    assert ctx.dtype == adt
    check("attn.fwd.ctx[p=%g]" % p_drop, _relerr(ctx.view(B, T, H, dh).permute(0, 2, 1, 3), O.detach()), ftol)
    lse_ref = torch.logsumexp((q64 @ k64.transpose(-1, -2) + b64) * scale, dim=-1) / np.log(2.0)
    check("attn.fwd.lse[p=%g]" % p_drop, (lse.double() - lse_ref.detach()).abs().max().item(), 1e-3)
    O.backward(dctx.view(B, T, H, dh).double().permute(0, 2, 1, 3))
    dqkv = torch.full((B * T, 3 * d), float("nan"), dtype=torch.bfloat16, device=dev)
    dbias = hip.relpos_attn_bwd(qu, k, v, bias, aux, dctx, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, T, dh, scale,
                                p_drop, seed)
    un = lambda t: t.reshape(B, T, H, dh).permute(0, 2, 1, 3)
    check("attn.bwd.dq[p=%g]" % p_drop, _relerr(un(dqkv[:, :d]), q64.grad), 2e-2)
    check("attn.bwd.dk[p=%g]" % p_drop, _relerr(un(dqkv[:, d:2 * d]), k64.grad), 2e-2)
    check("attn.bwd.dv[p=%g]" % p_drop, _relerr(un(dqkv[:, 2 * d:]), v64.grad), 2e-2)
    gb, wb = dbias.double().clone(), b64.grad.clone()
    gb[:, :, ii, ii + 1] = 0.0
    wb[:, :, ii, ii + 1] = 0.0
    check("attn.bwd.dbias[p=%g]" % p_drop, _relerr(gb, wb), 2e-2)
    # the unshift the engine applies afterwards equals autograd through the reference's pad-and-reshape
    dps = hip.relshift_bwd(dbias)
    raw_leaf = raw.clone().requires_grad_(True)
    (_ref_shift(raw_leaf) * wb).sum().backward()
    check("attn.bwd.unshift[p=%g]" % p_drop, _relerr(dps, raw_leaf.grad), 2e-2)
    if hip.relpos_attn_pos_supported(T, dh, adt):
        # backward with the positional-score gradients formed in the dQ kernel (no d(bias) tensor): dqu / dk / dv are the two-kernel
        # path's bit for bit; dqv and the positional-projection gradient against f64 products with the reference's un-shifted gradient
        quc = qu.contiguous()
        dqkv2 = torch.full((B * T, 3 * d), float("nan"), dtype=torch.bfloat16, device=dev)
        dqv = torch.full((B * T, d), float("nan"), dtype=torch.bfloat16, device=dev)
        dpart = hip.relpos_attn_bwd_pos(quc, qv, k, v, pos, bias_p, aux_p, dctx, dqkv2[:, :d], dqv, dqkv2[:, d:2 * d], dqkv2[:, 2 * d:],
                                        B, H, T, dh, scale, p_drop, seed)
        assert torch.equal(dqkv2, dqkv)
        g64 = raw_leaf.grad                                                      # (B,H,T,T): d loss / d R[b,h,r,m]
        want_dqv = torch.einsum("bhrm,mhc->brhc", g64, pos.view(T, H, dh).double()).reshape(B * T, d)
        want_dpos = torch.einsum("bhrm,brhc->mhc", g64, qv.view(B, T, H, dh).double()).reshape(T, d)
        assert not torch.isnan(dqv.float()).any() and not torch.isnan(dpart.float()).any()
        check("attn.bwd_pos.dqv[p=%g]" % p_drop, _relerr(dqv, want_dqv), 2e-2)
        assert dpart.shape == (B * ((T + 127) // 128), T, d)
        check("attn.bwd_pos.dpos[p=%g]" % p_drop, _relerr(dpart.double().sum(0), want_dpos), 2e-2)
        # the same kernels fed the PLAIN query projection and the u / v biases (they add them while loading their rows): bit for bit
        # what they produce from the biased copies a separate pass (bias2) stores
        q0 = mk(B * T, 3 * d)[:, :d]                                             # (a column slice: row stride 3d, as in the engine)
        ub, vb = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
        qu2, qv2 = hip.bias2(q0, ub, vb)
        ca, auxa, biasa = hip.relpos_attn_fwd_pos(qu2, qv2, k, v, pos, B, H, T, dh, scale, p_drop, seed)
        cb, auxb, biasb = hip.relpos_attn_fwd_pos(q0, q0, k, v, pos, B, H, T, dh, scale, p_drop, seed, biases=(ub, vb))
        assert torch.equal(ca, cb) and torch.equal(auxa[0], auxb[0]) and torch.equal(auxa[1], auxb[1]) and torch.equal(biasa, biasb)
        outs = []
        for qa, qb, bi in ((qu2, qv2, None), (q0, q0, (ub, vb))):
            o = [torch.full((B * T, d), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(5)]
            part = hip.relpos_attn_bwd_pos(qa, qb, k, v, pos, biasa, auxa, dctx, o[0], o[1], o[2], o[3], B, H, T, dh, scale, p_drop, seed,
                                           biases=bi, dq_sum=o[4])
            outs.append(o + [part])
        assert all(torch.equal(x1, x2) for x1, x2 in zip(*outs)) and not torch.isnan(outs[1][1].float()).any()
        # dq = dqu + dqv from the dK / dV kernel (after the boundary row is final): the separate add launch's result bit for bit
        assert torch.equal(outs[0][4], hip.axpby2d(outs[0][0], outs[0][1], 1.0, 1.0))
        # per row: the tile-boundary row (128) is assembled from two workgroups' halves
        rows = (dqv.double() - want_dqv).view(B, T, d).norm(dim=-1) / want_dqv.view(B, T, d).norm(dim=-1).clamp_min(1e-30)
        check("attn.bwd_pos.dqv_worst_row[p=%g]" % p_drop, rows.max().item(), 5e-2)


# ---------------------------------------------------------------- depthwise-conv tiles of the convolution module (csrc/dwconv.hip)
@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,d", [(2, 40, 32), (3, 256, 512), (1, 624, 256), (2, 70, 72)])
def test_dwglu_fused_kernels(B, T, d, dtp):
    """GLU + depthwise conv (k = 31, pad 15) + BatchNorm sums / data gradient + GLU backward / weight gradient against torch
    (convolution.py:139-143) in f64 on the same stored operands."""
    from sar_ssl_amd import hip
    from conftest import check
    dev = _dev()
    g = torch.Generator().manual_seed(B * 100 + T + d)
    h = torch.randn((B * T, 2 * d), generator=g).to(dtp).to(dev)
    w = (torch.randn((d, 31), generator=g) * 0.2).to(dev)
    dc = torch.randn((B * T, d), generator=g).to(dtp).to(dev)
    tol = 1e-5 if dtp == torch.float32 else 1e-2
    c, sums = hip.dwglu_fwd(h, w, B, T, want_stats=True)
    h64 = h.double().view(B, T, 2 * d).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    glu = h64[..., :d] * torch.sigmoid(h64[..., d:])
    if dtp == torch.bfloat16:
        glu = glu + (glu.detach().to(dtp).double() - glu.detach())          # forward value as stored (bf16), gradient of the exact op
    ref = torch.nn.functional.conv1d(glu.transpose(1, 2), w64.view(d, 1, 31), padding=15, groups=d).transpose(1, 2)
    check("dwglu.fwd[%d,%d,%d,%s]" % (B, T, d, dtp), _relerr(c.view(B, T, d), ref.detach()), tol)
    cs = c.double().view(-1, d)
    check("dwglu.fwd_sums", max(_relerr(sums[:d], cs.sum(0)), _relerr(sums[d:], (cs ** 2).sum(0))), 1e-5)
    ref.backward(dc.double().view(B, T, d))
    dh = hip.dwglu_bwd(dc, h, w, B, T)
    check("dwglu.bwd[%d,%d,%d,%s]" % (B, T, d, dtp), _relerr(dh.view(B, T, 2 * d), h64.grad), tol)
    dw = torch.ones((d, 31), dtype=torch.float32, device=dev)
    hip.dwglu_wgrad(dc, h, dw, B, T)
    check("dwglu.wgrad[%d,%d,%d,%s]" % (B, T, d, dtp), _relerr(dw - 1.0, w64.grad), 1e-4 if dtp == torch.float32 else 1e-2)


def test_gemm_group_tn_equals_separate_products():
    """Grouped weight-gradient launch (csrc/gemm.hip gemm_group_tn_kernel) vs the same products issued one by one and vs f64; the bias
    gradients (column sums of dy) that ride on the same launch vs f64 and vs the stand-alone column-sum kernel."""
    from sar_ssl_amd import hip
    torch.manual_seed(3)
    K = 1024
    shapes = [(512, 256), (256, 256), (2048, 512), (128, 128), (768, 256)]
    items, refs, outs, brefs, bouts = [], [], [], [], []
    for q, (M, N) in enumerate(shapes):
        dy = (torch.randn(K, M, device="cuda") * 0.1).to(torch.bfloat16)
        x = torch.randn(K, N, device="cuda").to(torch.bfloat16)
        out = torch.zeros(M, N, device="cuda") + 1.0                    # (+= semantics: starts non-zero)
        bias = (torch.zeros(M, device="cuda") + 2.0) if q != 1 else None    # (one product without a bias gradient)
        items.append((dy, x, out, 4, bias))
        refs.append(1.0 + dy.double().t() @ x.double())
        outs.append(out)
        brefs.append(2.0 + dy.double().sum(0))
        bouts.append(bias)
    with hip.splitk_batched():
        assert hip.gemm_group_tn(items)
    sep = []
    for dy, x, _, split, _b in items:
        o = torch.zeros(dy.shape[1], x.shape[1], device="cuda") + 1.0
        hip.gemm(dy, x, a_kc=False, b_kc=False, M=dy.shape[1], N=x.shape[1], K=K, lda=dy.stride(0), ldb=x.stride(0), out=o, ldc=x.shape[1],
                 split_k=split)
        sep.append(o)
    for o, r, s_ in zip(outs, refs, sep):
        assert float((o.double() - r).abs().max() / r.abs().max()) < 2e-3
        assert torch.equal(o, s_)                                        # same tiles, same split, same fold order
    for (dy, _x, _o, _s, _b), b, r in zip(items, bouts, brefs):
        if b is None:
            continue
        assert float((b.double() - r).abs().max() / r.abs().max()) < 1e-5    # exact bf16 terms, f32 accumulation
        alone = torch.zeros_like(b) + 2.0
        hip.colsum(dy, alone)
        assert float((b - alone).abs().max() / alone.abs().max()) < 1e-5
    # ragged shape: refused, nothing launched
    dy = torch.randn(K, 192, device="cuda").to(torch.bfloat16)
    with hip.splitk_batched():
        assert not hip.gemm_group_tn([(dy, dy, torch.zeros(192, 192, device="cuda"), 2, None), items[0]])


@pytest.mark.parametrize("dtp", [torch.bfloat16, torch.float32])
def test_layernorm_bwd_fused_dropout_output_equals_act_bwd(dtp):
    """hip.layernorm_bwd(drop=(p, seed, gscale)): the second output is what act_bwd would make of the first one - same keep mask
    (same counter hash on the element index), same scaling; in bf16 it is rounded once from the unrounded gradient instead of twice."""
    from sar_ssl_amd import hip
    torch.manual_seed(5)
    M, d = 384, 256
    x = torch.randn(M, d, device="cuda").to(dtp)
    dy = torch.randn(M, d, device="cuda").to(dtp)
    res = torch.randn(M, d, device="cuda").to(dtp)
    gamma = torch.rand(d, device="cuda") + 0.5
    beta = torch.zeros(d, device="cuda")
    _, stats = hip.layernorm_fwd(x, gamma, beta, 1e-5)
    p, seed, g = 0.1, 0x1234567887654321, 0.5
    out, out2 = hip.layernorm_bwd(dy, x, gamma, stats, resid=res, drop=(p, seed, g))
    ref = hip.layernorm_bwd(dy, x, gamma, stats, resid=res)
    assert torch.equal(out, ref)
    want = hip.act_bwd(ref, None, 0, p_drop=p, seed=seed, gscale=g)
    assert torch.equal(out2 == 0, want == 0)                               # identical keep mask
    frac = float((out2 == 0).float().mean())
    assert 0.07 < frac < 0.13
    tol = 1e-6 if dtp == torch.float32 else 1e-2
    assert float((out2.float() - want.float()).abs().max()) <= tol * float(want.float().abs().max())


def test_colsum_store_is_the_batch_sum_in_the_activation_dtype():
    from sar_ssl_amd import hip
    dev = _dev()
    x = _mk((64, 40 * 256), torch.bfloat16, dev, 3)
    got = hip.colsum_store(x)
    ref = x.float().sum(0)
    assert got.dtype == torch.bfloat16 and _relerr(got.float(), ref) < 6e-3
    xf = _mk((7, 132), torch.float32, dev, 4)
    assert _relerr(hip.colsum_store(xf), xf.double().sum(0)) < 1e-6


def test_patch_weight_gradient_from_split_k_partials_equals_zeroed_accumulate():
    """gemm_tn_partials + patch_wgrad_accum(nslice) == split-K product folded into a zeroed buffer, then re-laid-out."""
    from sar_ssl_amd import hip
    dev = _dev()
    M, d, F = 1024, 256, 32
    de = _mk((M, d), torch.bfloat16, dev, 5)
    z4 = _mk((M, F * 4), torch.bfloat16, dev, 6)
    ws, nslice = hip.gemm_tn_partials(de, z4, 8)
    assert nslice == 8 and ws.shape == (8, d, F * 4)
    got = torch.ones((d, 4, F, 1), device=dev)
    hip.patch_wgrad_accum(ws, got, nslice)
    full = de.float().t() @ z4.float()                                  # [d][f*4+c]
    ref = 1.0 + full.view(d, F, 4).permute(0, 2, 1).reshape(d, 4, F, 1)
    assert _relerr(got, ref) < 2e-3


# ---------------------------------------------------------------- first stem layer without its stored output (engine._C1IN)
def _c1_case(B, F, T, seed):
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    a0 = torch.randn((B, F, T, 4), generator=g).bfloat16().to(dev)
    W1 = (torch.randn((64, 4), generator=g) * 0.5).to(dev)
    scale = (torch.rand(64, generator=g) + 0.5).to(dev)
    scale[7] = -scale[7]
    shift = (torch.randn(64, generator=g) * 0.3).to(dev)
    return dev, g, a0, W1, scale, shift


def test_first_layer_batchnorm_sums_from_input_moments():
    from sar_ssl_amd import hip
    dev, g, a0, W1, _, _ = _c1_case(3, 16, 24, 51)
    hip.sums_arena_reset(dev)
    got = hip.stem_c1_stats(a0, W1).clone()
    y = a0.double().view(-1, 4) @ W1.double().t()
    ref = torch.cat([y.sum(0), (y * y).sum(0)])
    assert _relerr(got[:64], ref[:64]) < 1e-6 and _relerr(got[64:], ref[64:]) < 1e-6
    _, s_stored = hip.stem_c1_fwd(a0, W1, want_stats=True)             # sums of the bf16-rounded stored y1: same up to its rounding
    assert _relerr(s_stored[64:], ref[64:]) < 5e-3


@pytest.mark.parametrize("shape", [(2, 16, 40), (1, 24, 72)])
def test_conv3x3_from_the_4_channel_input_matches_the_stored_path(shape):
    """conv3x3_fwd_c1 / conv3x3_wgrad_c1 (operand relu(bn1(W1 a0)) formed while staging) against an f64 restatement and against the
    kernels that read a stored y1."""
    from sar_ssl_amd import hip
    B, F, T = shape
    dev, g, a0, W1, scale, shift = _c1_case(B, F, T, 52)
    w = (torch.randn((9, 64, 64), generator=g) * 0.05).bfloat16().to(dev)
    hip.sums_arena_reset(dev)
    r = hip.conv3x3_fwd_c1(a0, W1, scale, shift, w, want_stats=True)
    assert r is not None
    out, stats = r
    z = torch.relu((a0.double().view(-1, 4) @ W1.double().t()) * scale.double() + shift.double()).view(B, F, T, 64)
    zb = z.float().bfloat16().double()                                          # the operand is rounded to bf16 when staged
    ref = torch.nn.functional.conv2d(zb.permute(0, 3, 1, 2), w.double().view(3, 3, 64, 64).permute(2, 3, 0, 1), padding=1).permute(0, 2, 3, 1)
    assert _relerr(out.float(), ref) < 8e-3                                     # bf16 output rounding
    ob = out.double().view(-1, 64)
    assert _relerr(stats[:64], ob.sum(0)) < 1e-5 and _relerr(stats[64:], (ob * ob).sum(0)) < 1e-5
    y1 = hip.stem_c1_fwd(a0, W1)                                                # stored path: y1 rounded to bf16 in between
    out2 = hip.conv3x3_fwd(y1, w, scale, shift)
    assert _relerr(out.float(), out2.float()) < 3e-2
    # weight gradient
    dy = torch.randn((B, F, T, 64), generator=g).bfloat16().to(dev)
    gW = torch.zeros((64, 64, 3, 3), device=dev)
    assert hip.conv3x3_wgrad_c1(dy, a0, W1, scale, shift, gW)
    zp = torch.nn.functional.pad(zb.permute(0, 3, 1, 2), (1, 1, 1, 1))          # (B,64,F+2,T+2)
    dyd = dy.double().permute(0, 3, 1, 2)
    refW = torch.stack([torch.stack([torch.einsum("bofw,bifw->oi", dyd, zp[:, :, kh:kh + F, kw:kw + T]) for kw in range(3)], -1) for kh in range(3)], -2)
    assert _relerr(gW, refW) < 2e-3
    gW2 = torch.zeros((64, 64, 3, 3), device=dev)
    hip.conv3x3_wgrad(dy, y1, scale, shift, acc_into=gW2)
    assert _relerr(gW, gW2) < 2e-2


@pytest.mark.parametrize("train", [True, False])
def test_first_layer_backward_from_the_input_only(train):
    from sar_ssl_amd import hip
    dev, g, a0, W1, scale, shift = _c1_case(2, 16, 24, 53)
    npix = a0.numel() // 4
    dz1 = torch.randn((npix, 64), generator=g).bfloat16().to(dev)
    mean = (torch.randn(64, generator=g) * 0.2).to(dev); rstd = (torch.rand(64, generator=g) + 0.5).to(dev)
    aff = torch.stack([scale, shift, mean, rstd]).contiguous()
    dW, dga, dbe = torch.zeros((64, 4, 1, 1), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    hip.sums_arena_reset(dev)
    hip.stem_c1_bwd_a0(dz1, a0, W1, aff, train, dW, dga, dbe)
    y = a0.double().view(-1, 4) @ W1.double().t()
    gg = dz1.double() * ((y * scale.double() + shift.double()) > 0)
    xh = (y - mean.double()) * rstd.double()
    s1, s2 = gg.sum(0), (gg * xh).sum(0)
    dy = scale.double() * ((gg - s1 / npix - xh * s2 / npix) if train else gg)
    ref = dy.t() @ a0.double().view(-1, 4)
    assert _relerr(dW.view(64, 4), ref) < 2e-5
    assert _relerr(dbe, s1) < 1e-5 and _relerr(dga, s2) < 1e-5


@pytest.mark.parametrize("shape,train", [((2, 16, 40), True), ((1, 24, 72), False), ((3, 8, 32), True)])
def test_first_conv_data_gradient_consumed_in_its_epilogue(shape, train):
    """conv3x3_dgrad_c1red (data gradient masked and contracted against [a0 | 1] on the matrix cores, nothing stored) + the moment-based
    finalize == data-gradient launch followed by the one-pass first-layer backward, and an f64 restatement."""
    from sar_ssl_amd import hip
    B, F, T = shape
    dev, g, a0, W1, scale, shift = _c1_case(B, F, T, 54)
    npix = B * F * T
    w = (torch.randn((9, 64, 64), generator=g) * 0.05).bfloat16().to(dev)         # [tap][ci][co] data-gradient taps
    dy2 = torch.randn((B, F, T, 64), generator=g).bfloat16().to(dev)
    hip.sums_arena_reset(dev)
    sums, mom = hip.stem_c1_stats(a0, W1, keep_moments=True)
    y = a0.double().view(-1, 4) @ W1.double().t()
    mean = y.mean(0); var = y.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    aff = torch.stack([scale, shift, mean.float(), rstd.float()]).contiguous()
    dW, dga, dbe = torch.zeros((64, 4, 1, 1), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    assert hip.conv3x3_dgrad_c1red(dy2, w, a0, W1, aff, mom, train, dW, dga, dbe)
    # stored path with the same kernels' arithmetic: dz1 (bf16) then the one-pass backward that recomputes y1
    dz1 = hip.conv3x3_fwd(dy2, w)
    dW2, dga2, dbe2 = torch.zeros((64, 4, 1, 1), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    if npix % 64 == 0:
        hip.stem_c1_bwd_a0(dz1, a0, W1, aff, train, dW2, dga2, dbe2)
        assert _relerr(dW, dW2) < 2e-4 and _relerr(dga, dga2) < 2e-4 and _relerr(dbe, dbe2) < 2e-4
    # f64 restatement from the stored bf16 gradient
    gg = dz1.double().view(-1, 64) * ((y * scale.double() + shift.double()) > 0)
    xh = (y - aff[2].double()) * aff[3].double()
    s1, s2 = gg.sum(0), (gg * xh).sum(0)
    dyn = scale.double() * ((gg - s1 / npix - xh * s2 / npix) if train else gg)
    ref = dyn.t() @ a0.double().view(-1, 4)
    assert _relerr(dW.view(64, 4), ref) < 2e-4
    assert _relerr(dbe, s1) < 2e-4 and _relerr(dga, s2) < 2e-4


# frame counts: 40 -> pixel-order loop, 48 -> (8 bins x 16 frames) items, 64 / 96 -> (8 x 32) items (several items per workgroup row);
# 20 bins -> pixel-order loop whatever the frame count
@pytest.mark.parametrize("shape", [(3, 24, 40), (2, 24, 48), (3, 16, 64), (2, 40, 96), (2, 20, 64)])
@pytest.mark.parametrize("dtp", [torch.float32, torch.bfloat16])
def test_stem_c4_forward_statistics_epilogue(dtp, shape):
    """stem_c4_fwd against an f64 restatement (code/model.py:60-62: BatchNorm + ReLU, then Conv2d(64, 4, 1), stored (B,T,F,4)) for every
    loop variant of the kernel, and stem_c4_fwd(want_stats) == stem_c4_fwd followed by the statistics pass over its stored output."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(81)
    B, F, T = shape
    y3 = torch.randn((B, F, T, 64), generator=g).to(dtp).to(dev)
    W4 = (torch.randn((4, 64), generator=g) * 0.2).to(dev)
    sc = (torch.rand(64, generator=g) + 0.5).to(dev); sh = (torch.randn(64, generator=g) * 0.3).to(dev)
    hip.sums_arena_reset(dev)
    ref = hip.stem_c4_fwd(y3, W4, sc, sh)
    want = (torch.relu(y3.double() * sc.double() + sh.double()) @ W4.double().t()).permute(0, 2, 1, 3)     # (B,T,F,4)
    assert ref.shape == (B, T, F, 4)
    assert _relerr(ref.double(), want) < (1e-5 if dtp == torch.float32 else 6e-3)
    sums_ref, _ = hip.cl_stats(ref, 4)
    y4, sums = hip.stem_c4_fwd(y3, W4, sc, sh, want_stats=True)
    assert torch.equal(y4, ref)
    assert _relerr(sums[:4], sums_ref[:4]) < 1e-5 and _relerr(sums[4:], sums_ref[4:]) < 1e-5


# ---------------------------------------------------------------- fused feed-forward module (csrc/ffn2.hip)
@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("M,d", [(128, 256), (320, 512), (1024, 256)])
def test_fused_feed_forward_module_fwd_bwd(M, d, p_drop, dtp):
    """FeedForwardModule under the half-step residual (feed_forward.py:47-54, Conformer.py:60-67) as ONE launch per direction against
    (1) the pair of GEMM launches it replaces - same MFMA products in the same order, same dropout decisions: the saved tensors and
    the result must agree to the last rounding - and (2) an f64 restatement fed the kernel's own dropout mask."""
    from sar_ssl_amd import hip
    dev = _dev()
    gdt = torch.bfloat16
    H = 4 * d
    ln = (_mk((M, d), torch.float32, dev, 1)).to(dtp)
    x = (_mk((M, d), torch.float32, dev, 2)).to(dtp)
    W1 = (_mk((H, d), torch.float32, dev, 3) * d ** -0.5).to(dtp)
    W2 = (_mk((d, H), torch.float32, dev, 4) * H ** -0.5).to(dtp)
    b1 = _mk((H,), torch.float32, dev, 5) * 0.1
    b2 = _mk((d,), torch.float32, dev, 6) * 0.1
    s1, s2, factor = 0x1234567, 0x7654321, 0.5
    w1p, w2p = torch.empty(H * d, dtype=dtp, device=dev), torch.empty(H * d, dtype=dtp, device=dev)
    w2tp, w1tp = torch.empty(H * d, dtype=gdt, device=dev), torch.empty(H * d, dtype=gdt, device=dev)
    hip.ffn_pack([(W1, w1p), (W2, w2p), (W2.to(gdt).t(), w2tp), (W1.to(gdt).t(), w1tp)])
    # the pack itself: block (n / 32, k / 16), lane l = row 32 nb + (l & 31), k = 16 ks + 8 (l >> 5) + 0..7
    pk = w1p.view(H // 32, d // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(H, d)
    assert torch.equal(pk, W1)
    pk = w2tp.view(H // 32, d // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(H, d)
    assert torch.equal(pk, W2.to(gdt).t())
    y, hpre, a = hip.ffn2_fwd(ln, w1p, w2p, b1, b2, x, d, p1=p_drop, s1=s1, p2=p_drop, s2=s2, out_scale=factor)
    # (1) the two-launch sequence of engine.ffn_fwd
    hpre_r = torch.empty((M, H), dtype=dtp, device=dev)
    a_r = hip.gemm(ln, W1, M=M, N=H, K=d, lda=d, ldb=d, bias=b1, act=2, preact=hpre_r, p_drop=p_drop, seed=s1)
    y_r = hip.gemm(a_r, W2, M=M, N=d, K=H, lda=H, ldb=H, bias=b2, p_drop=p_drop, seed=s2, out_scale=factor, resid=x, ldr=d, res_scale=1.0)
    assert torch.equal(hpre, hpre_r)
    assert torch.equal(a, a_r)
    # workgroups 8, 9, ... walk the hidden chunks from another start (csrc/ffn2.hip, chunk rotation): their sum over the hidden dimension
    # has another order than the GEMM's K loop - the 16-bit result may differ by a rounding on a few entries
    rotated = M > 8 * 64

    def same(u, v):
        if not rotated:
            return _relerr(u, v) < 1e-6
        ulp = 2.0 ** (-10 if u.dtype == torch.float16 else -7)
        return _relerr(u, v) <= ulp and (u != v).float().mean().item() < 0.05
    assert same(y, y_r)
    # (2) f64, with the kernel's own masks (read back from its outputs: hidden == 0 where dropped; second mask from the unfused twin at p = 1 - 1)
    h64 = ln.double() @ W1.double().t() + b1.double()
    act = h64 * torch.sigmoid(h64)
    if p_drop > 0:
        # the library's own masks: keep / (1 - p) as a function of (seed, m * N + n), read back through sarssl_act_bwd on ones
        keep1 = hip.act_bwd(torch.ones((M, H), dtype=torch.float32, device=dev), None, 0, p_drop=p_drop, seed=s1).double()
        keep2 = hip.act_bwd(torch.ones((M, d), dtype=torch.float32, device=dev), None, 0, p_drop=p_drop, seed=s2).double()
        assert abs(1.0 - (keep1 != 0).double().mean().item() - p_drop) < 0.02
        act = act * keep1
    else:
        keep2 = torch.ones((M, d), dtype=torch.float64, device=dev)
    tol = 4e-3 if dtp == torch.float16 else 2e-2
    assert _relerr(hpre, h64) < tol
    assert _relerr(a, act) < tol
    y64 = x.double() + factor * ((a.double() @ W2.double().t() + b2.double()) * keep2)
    assert _relerr(y, y64) < tol
    # ---- backward: dh, dln against the two launches of engine.ffn_bwd and f64
    dz2 = (_mk((M, d), torch.float32, dev, 7) * 1e-3).to(gdt)
    dln, dh = hip.ffn2_bwd(dz2, w2tp, w1tp, hpre, d, p1=p_drop, s1=s1)
    W2g, W1g = W2.to(gdt), W1.to(gdt)
    dh_r = hip.gemm(dz2, W2g, a_kc=True, b_kc=False, M=M, N=H, K=d, lda=d, ldb=H, aux=hpre, aux_act=2, p_drop=p_drop, seed=s1)
    dln_r = hip.gemm(dh_r, W1g, a_kc=True, b_kc=False, M=M, N=d, K=H, lda=H, ldb=d)
    assert dh.dtype == gdt and dln.dtype == gdt
    assert torch.equal(dh, dh_r)
    assert same(dln, dln_r)
    hp = hpre.double()
    sg = torch.sigmoid(hp)
    dh64 = (dz2.double() @ W2g.double()) * (sg * (1 + hp * (1 - sg)))
    if p_drop > 0:
        dh64 = dh64 * keep1
    assert _relerr(dh, dh64) < 2e-2
    assert _relerr(dln, dh.double() @ W1g.double()) < 2e-2


@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,d", [(128, 256), (192, 512), (1024, 256)])
def test_fused_feed_forward_module_with_its_layernorm(M, d, dtp):
    """The module's LayerNorm inside the fused launches (feed_forward.py:48): forward - the prologue normalises the tile's rows with the
    arithmetic of sarssl_layernorm_fwd, so ln / statistics and everything downstream are BIT-identical to LayerNorm launch + fused launch;
    backward - the epilogue runs sarssl_layernorm_bwd(_drop) on the second product: dx / dx2 / dgamma / dbeta against the stand-alone
    kernel on the stored dln (same formula; the row sums are folded in another order: f32 round-off apart)."""
    from sar_ssl_amd import hip
    dev = _dev()
    gdt = torch.bfloat16
    H = 4 * d
    x = (_mk((M, d), torch.float32, dev, 11) * 1.5 + 0.3).to(dtp)
    gamma = 1.0 + 0.1 * _mk((d,), torch.float32, dev, 12)
    beta = 0.1 * _mk((d,), torch.float32, dev, 13)
    W1 = (_mk((H, d), torch.float32, dev, 3) * d ** -0.5).to(dtp)
    W2 = (_mk((d, H), torch.float32, dev, 4) * H ** -0.5).to(dtp)
    b1 = _mk((H,), torch.float32, dev, 5) * 0.1
    b2 = _mk((d,), torch.float32, dev, 6) * 0.1
    w1p, w2p = torch.empty(H * d, dtype=dtp, device=dev), torch.empty(H * d, dtype=dtp, device=dev)
    w2tp, w1tp = torch.empty(H * d, dtype=gdt, device=dev), torch.empty(H * d, dtype=gdt, device=dev)
    hip.ffn_pack([(W1, w1p), (W2, w2p), (W2.to(gdt).t(), w2tp), (W1.to(gdt).t(), w1tp)])
    p, s1, s2 = 0.1, 77, 78
    # forward
    ln_r, st_r = hip.layernorm_fwd(x, gamma, beta, 1e-5)
    y_r, hpre_r, a_r = hip.ffn2_fwd(ln_r, w1p, w2p, b1, b2, x, d, p1=p, s1=s1, p2=p, s2=s2, out_scale=0.5)
    y, hpre, a, ln, st = hip.ffn2_fwd(None, w1p, w2p, b1, b2, x, d, p1=p, s1=s1, p2=p, s2=s2, out_scale=0.5, ln_in=(x, gamma, beta, 1e-5))
    assert torch.equal(ln, ln_r)
    if d == 256:
        assert torch.equal(st, st_r)
    else:       # d = 512 (two float4 per lane): the compiler contracts the variance sum differently in the two kernels - an ulp in mean / rstd
        assert _relerr(st, st_r) < 1e-6
    assert torch.equal(hpre, hpre_r) and torch.equal(a, a_r) and torch.equal(y, y_r)
    # backward
    dz2 = (_mk((M, d), torch.float32, dev, 7) * 1e-3).to(gdt)
    dy = (_mk((M, d), torch.float32, dev, 8) * 1e-3).to(gdt)
    for drop in (None, (0.1, 99, 0.5)):
        dln_r, dh_r = hip.ffn2_bwd(dz2, w2tp, w1tp, hpre, d, p1=p, s1=s1)
        dg_r, db_r = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        ref = hip.layernorm_bwd(dln_r, x, gamma, st, resid=dy, dgamma=dg_r, dbeta=db_r, drop=drop)
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        got, dh = hip.ffn2_bwd(dz2, w2tp, w1tp, hpre, d, p1=p, s1=s1, ln_bwd=(x, gamma, st, dy, dg, db, drop))
        assert torch.equal(dh, dh_r)
        dx_r, dx2_r = ref if drop is not None else (ref, None)
        dx, dx2 = got if drop is not None else (got, None)
        assert _relerr(dx, dx_r) < 1e-2                                         # (bf16 outputs: one rounding step of the largest entry)
        assert float((dx.float() - dx_r.float()).abs().mean() / dx_r.float().abs().mean()) < 2e-4
        if drop is not None:
            assert torch.equal(dx2 == 0, dx2_r == 0) and _relerr(dx2, dx2_r) < 1e-2
        assert _relerr(dg, dg_r) < 1e-4 and _relerr(db, db_r) < 1e-4


# ---------------------------------------------------------------- row-tile-resident Linear layers of the d = 256 blocks (csrc/lin256.hip)
@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 768, 256), (192, 512, 256), (256, 256, 256), (128, 256, 768), (192, 256, 512), (1024, 768, 256)])
def test_tile_resident_linear_layers(M, N, K, dtp):
    """nn.Linear / pointwise Conv1d of the d = 256 blocks as one launch per layer (attention.py:82-85, convolution.py:138,143) against the
    sarssl_gemm launches it replaces: forward with bias / dropout / residual (same MFMA products in the same order, same dropout
    decisions: equal to the last rounding), with the LayerNorm in front formed in the prologue (bit-identical to sarssl_layernorm_fwd +
    the plain launch); data gradient plain and with the LayerNorm backward in the epilogue (against sarssl_layernorm_bwd_drop)."""
    from sar_ssl_amd import hip
    dev = _dev()
    gdt = torch.bfloat16
    a = _mk((M, K), torch.float32, dev, 1).to(dtp)
    W = (_mk((N, K), torch.float32, dev, 2) * K ** -0.5).to(dtp)
    bias = _mk((N,), torch.float32, dev, 3) * 0.1
    wp = torch.empty(N * K, dtype=dtp, device=dev)
    hip.ffn_pack([(W, wp)])
    res = _mk((M, N), torch.float32, dev, 4).to(dtp)
    # forward: bias + dropout + residual
    y = hip.lin256_fwd(a, wp, bias, N, K, resid=res, p_drop=0.1, seed=5)
    y_r = hip.gemm(a, W, M=M, N=N, K=K, lda=K, ldb=K, bias=bias, p_drop=0.1, seed=5, resid=res, ldr=N, res_scale=1.0)
    assert _relerr(y, y_r) < 1e-6
    y = hip.lin256_fwd(a, wp, bias, N, K)
    assert torch.equal(y, hip.gemm(a, W, M=M, N=N, K=K, lda=K, ldb=K, bias=bias))
    ref64 = a.double() @ W.double().t() + bias.double()
    assert _relerr(y, ref64) < (4e-3 if dtp == torch.float16 else 2e-2)
    if K == 256:    # LayerNorm prologue
        x = (_mk((M, K), torch.float32, dev, 6) * 1.3 + 0.2).to(dtp)
        gamma, beta = 1.0 + 0.1 * _mk((K,), torch.float32, dev, 7), 0.1 * _mk((K,), torch.float32, dev, 8)
        ln_r, st_r = hip.layernorm_fwd(x, gamma, beta, 1e-5)
        y2, ln, st = hip.lin256_fwd(None, wp, bias, N, K, ln_in=(x, gamma, beta, 1e-5))
        assert torch.equal(ln, ln_r) and torch.equal(st, st_r) and torch.equal(y2, hip.lin256_fwd(ln_r, wp, bias, N, K))
    # data gradient: dx [M, K] = dy [M, N] W   (pack of W^T: [K x N])
    if N <= 768 and (N == 256 or K == 256):
        dy = (_mk((M, N), torch.float32, dev, 9) * 1e-3).to(gdt)
        Wg = W.to(gdt)
        wtp = torch.empty(N * K, dtype=gdt, device=dev)
        hip.ffn_pack([(Wg.t(), wtp)])
        dx = hip.lin256_bwd(dy, wtp, K, N)
        dx_r = hip.gemm(dy, Wg, a_kc=True, b_kc=False, M=M, N=K, K=N, lda=N, ldb=K)
        assert torch.equal(dx, dx_r)
        if K == 256:
            x = (_mk((M, K), torch.float32, dev, 6) * 1.3 + 0.2).to(dtp)
            gamma = 1.0 + 0.1 * _mk((K,), torch.float32, dev, 7)
            _, st = hip.layernorm_fwd(x, gamma, torch.zeros(K, device=dev), 1e-5)
            rs = (_mk((M, K), torch.float32, dev, 10) * 1e-3).to(gdt)
            for drop in (None, (0.1, 31, 0.5)):
                dg_r, db_r = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
                ref = hip.layernorm_bwd(dx_r, x, gamma, st, resid=rs, dgamma=dg_r, dbeta=db_r, drop=drop)
                dg, db = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
                got = hip.lin256_bwd(dy, wtp, K, N, ln_bwd=(x, gamma, st, rs, dg, db, drop))
                g1, r1 = (got[0], ref[0]) if drop is not None else (got, ref)
                assert _relerr(g1, r1) < 1e-2 and float((g1.float() - r1.float()).abs().mean() / r1.float().abs().mean()) < 2e-4
                if drop is not None:
                    assert torch.equal(got[1] == 0, ref[1] == 0) and _relerr(got[1], ref[1]) < 1e-2
                assert _relerr(dg, dg_r) < 1e-4 and _relerr(db, db_r) < 1e-4


# ---------------------------------------------------------------- decoder on the masked frames only: gather / scatter / compact loss
@pytest.mark.parametrize("dtp", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,F", [(3, 16, 16), (2, 40, 256), (2, 624, 256)])
def test_masked_rows_gather_scatter_and_compact_loss(B, T, F, dtp):
    """sarssl_gather_rows / sarssl_scatter_rows and the loss on the compact prediction against the full-frame launches: same loss / diff,
    the same gradient rows at the masked frames (bit for bit: same arithmetic per element), zeros elsewhere."""
    from sar_ssl_amd import hip
    dev = _dev()
    g = np.random.default_rng(5)
    nm = T // 2
    idx = np.stack([np.sort(g.choice(T, nm, replace=False)) for _ in range(B)]).astype(np.int32)
    ch = g.integers(0, 2, size=B).astype(np.int32)
    idx_t, ch_t = torch.from_numpy(idx).to(dev), torch.from_numpy(ch).to(dev)
    mp = np.ones((B, T), dtype=np.uint8)
    np.put_along_axis(mp, idx.astype(np.int64), 0, axis=1)
    mp_t = torch.from_numpy(mp).to(dev)
    d = 768
    e = _mk((B * T, d), torch.float32, dev, 1).to(dtp)
    ec = hip.gather_rows(e, idx_t, B, T)
    want = torch.stack([e.view(B, T, d)[b, torch.from_numpy(idx[b]).long().to(dev)] for b in range(B)]).reshape(B * nm, d)
    assert torch.equal(ec, want)
    back = hip.scatter_rows(ec, idx_t, B, T)
    assert torch.equal(back.view(B, T, d) * (1 - mp_t.view(B, T, 1).to(dtp)), back.view(B, T, d))
    assert torch.equal(back.view(B, T, d)[mp_t == 0], e.view(B, T, d)[mp_t == 0]) and float(back.view(B, T, d)[mp_t == 1].abs().max()) == 0.0
    # loss
    x = _mk((B, 2, F, T, 2), torch.float32, dev, 2)
    pred = _mk((B, T, F * 4), torch.float32, dev, 3).to(dtp)
    pred_c = hip.gather_rows(pred.view(B * T, F * 4), idx_t, B, T)
    gdt = torch.bfloat16 if dtp == torch.float16 else dtp
    out, dpred = hip.masked_mse_fwd(pred, x, idx_t, ch_t, with_grad=True)
    out_c, dpred_c = hip.masked_mse_compact(pred_c, x, idx_t, ch_t, with_grad=True)
    assert _relerr(out_c, out) < 1e-6 and dpred_c.dtype == gdt
    assert torch.equal(dpred_c, hip.gather_rows(dpred.view(B * T, F * 4), idx_t, B, T))
    assert _relerr(hip.masked_mse_compact(pred_c, x, idx_t, ch_t), out) < 1e-6
    gs = torch.tensor([0.37], dtype=torch.float32, device=dev)
    d2 = hip.masked_mse_bwd(pred, x, mp_t, ch_t, nm, 1.0, gs)
    d2c = hip.masked_mse_bwd_compact(pred_c, x, idx_t, ch_t, 1.0, gs)
    assert torch.equal(d2c, hip.gather_rows(d2.view(B * T, F * 4), idx_t, B, T))
