"""Tensor-level wrappers over the C-ABI kernels (raw ops; autograd lives in ops.py).

PyTorch is used for device memory and streams only: every wrapper passes ``data_ptr()``s and the
current HIP stream to ``libsarssl_hip.so``.
"""
import torch

from . import _lib
from ._lib import c_void_p, c_int, c_long, c_float, c_ulonglong

F32, BF16, I16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.int16: I16}


def dt(t):
    return _DT[t.dtype]


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SarsslHipError("sar_ssl_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")


_ws_cache = {}


def workspace(nbytes, device, tag="default"):
    key = (tag, device)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = w
    return w


def gemm(A, B, *, a_kc=True, b_kc=True, M, N, K, lda, ldb, out=None, out_dtype=None, ldc=None,
         nbatch=1, batch_inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, out_scale=1.0, bias=None, act=0,
         resid=None, ldr=0, sR=(0, 0), res_scale=1.0, preact=None, p_drop=0.0, seed=0, precise=False, out_shape=None):
    """C[z] = epilogue(alpha * opA(A[z]) @ opB(B[z])^T) - see csrc/gemm.hip for the layout flags."""
    _need_cuda(A, B, out, bias, resid, preact)
    if out is None:
        out = torch.empty(out_shape if out_shape is not None else (nbatch, M, N) if nbatch > 1 else (M, N),
                          dtype=out_dtype or A.dtype, device=A.device)
    if ldc is None:
        ldc = N
    if nbatch > 1 and sC == (0, 0):
        sC = (M * N * batch_inner, M * N)
    ws = None
    if precise and A.dtype == torch.float32:
        ws = workspace(4 * nbatch * M * N, A.device, "gemm_acc")
    _lib.call("sarssl_gemm", _p(A), _p(B), _p(out), c_int(dt(A)), c_int(dt(B)), c_int(dt(out)),
              c_int(1 if a_kc else 0), c_int(1 if b_kc else 0), c_int(M), c_int(N), c_int(K),
              c_long(lda), c_long(ldb), c_long(ldc), c_int(nbatch), c_int(batch_inner),
              c_long(sA[0]), c_long(sA[1]), c_long(sB[0]), c_long(sB[1]), c_long(sC[0]), c_long(sC[1]),
              c_float(alpha), c_float(out_scale), _p(bias), c_int(act),
              _p(resid), c_long(ldr), c_long(sR[0]), c_long(sR[1]), c_float(res_scale),
              _p(preact), c_float(p_drop), c_ulonglong(seed), c_int(1 if (precise and A.dtype == torch.float32) else 0),
              _p(ws), _stream())
    return out


def stft_frontend(sig, eps=1e-6, win_len=512, hop=256, nfft=512):
    """(B, nsample, nch) f32|int16 -> (B*(nch-1), 2, nfft/2, nt, 2) f32 (data_preprocess output, ch_mode 'M')."""
    _need_cuda(sig)
    sig = sig.contiguous()
    nb, nsample, nch = sig.shape
    nt = (nsample - win_len) // hop + 1
    U = torch.empty((nb, nch, nfft // 2 + 1, nt, 2), dtype=torch.float32, device=sig.device)
    magsum = torch.empty((nb,), dtype=torch.float64, device=sig.device)
    out = torch.empty((nb * (nch - 1), 2, nfft // 2, nt, 2), dtype=torch.float32, device=sig.device)
    _lib.call("sarssl_stft_frontend", _p(sig), c_int(dt(sig)), c_int(nb), c_long(nsample), c_int(nch), c_int(win_len),
              c_int(hop), c_int(nfft), c_int(nt), c_float(eps), _p(U), _p(magsum), _p(out), _stream())
    return out


def stft_raw(sig, win_len=512, hop=256, nfft=512):
    """(B, nsample, nch) -> complex64 (B, nfft/2+1, nt, nch) like STFT.forward."""
    _need_cuda(sig)
    sig = sig.contiguous()
    nb, nsample, nch = sig.shape
    nt = (nsample - win_len) // hop + 1
    U = torch.empty((nb, nch, nfft // 2 + 1, nt, 2), dtype=torch.float32, device=sig.device)
    magsum = torch.empty((nb,), dtype=torch.float64, device=sig.device)
    out = torch.empty((nb, nfft // 2 + 1, nt, nch, 2), dtype=torch.float32, device=sig.device)
    _lib.call("sarssl_stft_raw", _p(sig), c_int(dt(sig)), c_int(nb), c_long(nsample), c_int(nch), c_int(win_len),
              c_int(hop), c_int(nfft), c_int(nt), _p(U), _p(magsum), _p(out), _stream())
    return torch.view_as_complex(out)
