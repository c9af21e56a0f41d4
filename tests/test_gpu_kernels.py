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
@pytest.mark.parametrize("mode", ["bf16", "bf16_f32out", "f32_precise", "f32_fast"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 40), (48, 136, 32), (1000, 264, 520)])
def test_gemm_layouts(a_kc, b_kc, mode, M, N, K):
    from sar_ssl_amd import hip
    dev = _dev()
    if not a_kc and M % 8:
        pytest.skip("transposed A needs M % 8 == 0")
    in_dt = torch.bfloat16 if mode.startswith("bf16") else torch.float32
    out_dt = torch.bfloat16 if mode == "bf16" else torch.float32
    A = _mk((M, K) if a_kc else (K, M), in_dt, dev, 1)
    B = _mk((N, K) if b_kc else (K, N), in_dt, dev, 2)      # asymmetric random operands
    bias = _mk((N,), torch.float32, dev, 3)
    C = hip.gemm(A, B, a_kc=a_kc, b_kc=b_kc, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], out_dtype=out_dt,
                 bias=bias, alpha=0.5, precise=(mode == "f32_precise"))
    Af = A.float() if a_kc else A.float().t()
    Bf = B.float() if b_kc else B.float().t()
    ref = 0.5 * (Af.double() @ Bf.double().t()) + bias.double()
    err = _relerr(C, ref)
    tol = {"bf16": 1e-2, "bf16_f32out": 1e-5, "f32_precise": 5e-5, "f32_fast": 1e-2}[mode]
    if mode == "bf16_f32out":
        tol = 1e-5       # bf16 inputs are exact in the reference too; only f32 accumulation order differs
    assert err < tol, (mode, err)


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
