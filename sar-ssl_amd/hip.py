"""Tensor-level wrappers over the C-ABI kernels (raw ops; the autograd anchor is model._PretrainFn / autograd.py, orchestration engine.py).

PyTorch is used for device memory and streams only: every wrapper passes ``data_ptr()``s and the
current HIP stream to ``libsarssl_hip.so``.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import c_void_p, c_int, c_long, c_float, c_ulonglong

F32, BF16, I16, F16, MIX16, MIXF32 = 0, 1, 2, 3, 4, 5
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.int16: I16, torch.float16: F16}
_16 = (torch.bfloat16, torch.float16)


def dt(t):
    return _DT[t.dtype]


def dt_ga(g, a):
    """dtype code of a backward kernel that reads the gradient tensor ``g`` next to the tensor ``a`` SAVED BY THE FORWARD PASS: their
    common dtype, or MIX16 for bf16 gradients + fp16 saved activations (the fp16-forward / bf16-backward mode, include/sarssl_hip.h).
    Anything else is a plumbing error and raises here instead of being reinterpreted by a kernel."""
    if a is None or g.dtype == a.dtype:
        return _DT[g.dtype]
    if g.dtype == torch.bfloat16 and a.dtype == torch.float16:
        return MIX16
    if g.dtype == torch.bfloat16 and a.dtype == torch.float32:      # hybrid mode: bf16 gradient next to the f32 prediction / stream
        return MIXF32
    raise _lib.SarsslHipError("unsupported gradient / saved-activation dtype pair (%s, %s)" % (g.dtype, a.dtype))


_hybrid = False        # runtime.set_precision('hybrid'): branch gradients are bf16 also next to an f32 tensor (the f32 prediction)


def gdtype_of(a_dtype):
    """Gradient dtype that goes with activations of ``a_dtype``: bf16 for fp16 activations (no loss scaling needed), else the same
    (hybrid mode: bf16 next to the f32 prediction as well - the loss gradient is a matrix-core operand)."""
    if _hybrid and a_dtype == torch.float32:
        return torch.bfloat16
    return torch.bfloat16 if a_dtype == torch.float16 else a_dtype


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    """Kernels are enqueued on the CURRENT device's current stream: tensors must be GPU tensors of that device (a model built on
    ``cuda:1`` needs ``torch.cuda.set_device(1)`` / a ``torch.cuda.device`` context, as run_pretrain.py and bench.py do)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.SarsslHipError("sar_ssl_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise _lib.SarsslHipError("tensor lives on cuda:%d but the current device is cuda:%d: call torch.cuda.set_device(%d) "
                                      "(kernels are launched on the current device's stream)" % (t.device.index, cur, t.device.index))


# ---- optional per-kernel timing with events on the launch stream (bench.py roofline) ---------------------------------
_prof = None
_prof_shapes = os.environ.get("SARSSL_PROF_SHAPES", "0") == "1"      # tools: label every GEMM launch with its shape


def profile_start(all_calls=False):
    """Event-time the labelled launches (_Timed) until profile_stop(); all_calls: additionally EVERY C-ABI call under its entry point's
    name ("call:<symbol>": a step's whole kernel-time budget by family - run single-stream, where a launch has the GPU to itself)."""
    global _prof
    _prof = {}
    _lib.call_timer = (lambda name: _Timed("call:" + name)) if all_calls else None


def profile_stop():
    """-> {name: (launches, total_ms)}; synchronises."""
    global _prof
    p, _prof = _prof, None
    _lib.call_timer = None
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in (p or {}).items()}


_runway_cycles_per_ms = None


def gpu_runway(ms):
    """Enqueue ~``ms`` milliseconds of spinning on the current stream (torch.cuda._sleep, calibrated once with events).  Event-bracketed
    launches only measure kernel durations while the GPU queue never runs dry: an eager step that is being profiled call by call is
    host-bound (two event records per launch), and every interval would include the host's gaps - so the profiled step is enqueued
    BEHIND a runway long enough for the host to finish enqueueing it."""
    global _runway_cycles_per_ms
    if _runway_cycles_per_ms is None:
        torch.cuda._sleep(1000)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        torch.cuda._sleep(2000000)
        b.record()
        torch.cuda.synchronize()
        _runway_cycles_per_ms = 2000000.0 / max(a.elapsed_time(b), 1e-3)
    torch.cuda._sleep(int(ms * _runway_cycles_per_ms))


class _Timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _prof is not None and self.name is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()                      # current stream == the stream the kernel is launched on
        return self

    def __exit__(self, *exc):
        if _prof is not None and self.name is not None:
            self.b.record()
            _prof.setdefault(self.name, []).append((self.a, self.b))
        return False


# ---- pre-zeroed f64 arena for the reductions' accumulators (csrc/api.hip sarssl_ctx_zero_arena): one memset per forward / backward
#      pass instead of one in front of every reduction launch
_ARENA_DOUBLES = 1 << 16
_arena = {}                 # device index -> [tensor, cursor]


def sums_arena_reset(device):
    """Zero the arena (one memset on the current stream) and start handing out slices from its beginning.  Call at the start of a
    forward and of a backward pass: every slice handed out before has been consumed inside the pass that took it."""
    if device.type != "cuda":
        return
    ent = _arena.get(device.index)
    if ent is None:
        t = torch.zeros((_ARENA_DOUBLES,), dtype=torch.float64, device=device)
        ent = _arena[device.index] = [t, 0]
        _lib.call("sarssl_ctx_zero_arena", c_void_p(_lib.ctx(device.index)), _p(t), c_long(t.numel() * 8))
    ent[0].zero_()
    ent[1] = 0


def _sums(n, device):
    """f64[n] accumulator: a zeroed arena slice when the arena is active for this device, else uninitialised memory (the launch
    wrappers memset whatever lies outside the arena)."""
    ent = _arena.get(device.index)
    if ent is not None and ent[1] + n <= _ARENA_DOUBLES:
        o = ent[1]
        ent[1] = o + (n + 15) // 16 * 16
        return ent[0][o:o + n]
    return torch.empty((n,), dtype=torch.float64, device=device)


_ws_cache = {}


def workspace(nbytes, device, tag="default"):
    key = (tag, device, torch.cuda.current_stream().cuda_stream)      # one scratch buffer per stream: no cross-stream reuse
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = w
    return w


def gemm(A, B, *, a_kc=True, b_kc=True, M, N, K, lda, ldb, out=None, out_dtype=None, ldc=None,
         nbatch=1, batch_inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, out_scale=1.0, bias=None, act=0,
         resid=None, ldr=0, sR=(0, 0), res_scale=1.0, preact=None, aux=None, aux_act=0, p_drop=0.0, seed=0, precise=False,
         out_shape=None, split_k=0, c_row_shift=False):
    """C[z] = epilogue(alpha * opA(A[z]) @ opB(B[z])^T) - see csrc/gemm.hip for the layout flags."""
    _need_cuda(A, B, out, bias, resid, preact, aux)
    if out is None:
        out = torch.empty(out_shape if out_shape is not None else (nbatch, M, N) if nbatch > 1 else (M, N),
                          dtype=out_dtype or A.dtype, device=A.device)
    if ldc is None:
        ldc = N
    if nbatch > 1 and sC == (0, 0):
        sC = (M * N * batch_inner, M * N)
    ws = None
    deferred = None
    if split_k > 0:
        if (_splitk_batch is not None and nbatch == 1 and A.dtype == torch.bfloat16 and B.dtype in _16 and out.dtype == torch.float32 and N % 4 == 0
                and ldc % 4 == 0):
            # weight-gradient product inside splitk_batched(): write the partials only, fold them into `out` together with the
            # stage's other products when the batch closes
            per = ((K + split_k - 1) // split_k + 63) // 64 * 64
            nsplit = (K + per - 1) // per
            ws = torch.empty((nsplit * M * N,), dtype=torch.float32, device=A.device)
            deferred = (ws, nsplit, M, N, out, ldc)
        else:
            ws = workspace(4 * nbatch * split_k * M * N, A.device, "gemm_split")
    elif precise and A.dtype == torch.float32:
        ws = workspace(4 * nbatch * M * N, A.device, "gemm_acc")
    with _Timed("gemm[%d,%d,%d x%d %s%s %s%s%s%s]" % (M, N, K, nbatch, "k" if a_kc else "m", "k" if b_kc else "n", str(out.dtype)[6:],
                                                 " sk%d" % split_k if split_k else "", " act%d" % act if act else "",
                                                 " aux" if aux is not None else "") if _prof_shapes and _prof is not None else None):
      _lib.call("sarssl_gemm", _p(A), _p(B), _p(out) if deferred is None else c_void_p(0), c_int(dt(A)), c_int(dt(B)), c_int(dt(out)),
              c_int(1 if a_kc else 0), c_int(1 if b_kc else 0), c_int(M), c_int(N), c_int(K),
              c_long(lda), c_long(ldb), c_long(ldc), c_int(nbatch), c_int(batch_inner),
              c_long(sA[0]), c_long(sA[1]), c_long(sB[0]), c_long(sB[1]), c_long(sC[0]), c_long(sC[1]),
              c_float(alpha), c_float(out_scale), _p(bias), c_int(act),
              _p(resid), c_long(ldr), c_long(sR[0]), c_long(sR[1]), c_float(res_scale),
              _p(preact), _p(aux), c_int(aux_act), c_int(dt(aux) if aux is not None else dt(out)), c_float(p_drop), c_ulonglong(seed),
              c_int(1 if (precise and A.dtype == torch.float32) else 0),
              _p(ws), c_int(split_k), c_int(1 if c_row_shift else 0), _stream())
    if deferred is not None:
        _splitk_batch.append(deferred)
        if len(_splitk_batch) == 32:
            splitk_flush()
    return out


# ---- downstream heads (csrc/head.hip): mean over frames, small f32 Linear layers ----------------------------------------------------
def mean_rows(x3d):
    """(B, T, d) of any storage dtype -> f32 (B, d): mean over the frames."""
    _need_cuda(x3d)
    B, T, d = x3d.shape
    x3d = x3d.contiguous()
    out = torch.empty((B, d), dtype=torch.float32, device=x3d.device)
    _lib.call("sarssl_mean_rows", _p(x3d), c_int(B), c_int(T), c_int(d), _p(out), c_int(dt(x3d)), _stream())
    return out


def mean_rows_bwd(dy2d, T, dtype=torch.float32):
    B, d = dy2d.shape
    dx = torch.empty((B, T, d), dtype=dtype, device=dy2d.device)
    _lib.call("sarssl_mean_rows_bwd", _p(dy2d.contiguous().float()), c_int(B), c_int(T), c_int(d), _p(dx), c_int(_DT[dtype]), _stream())
    return dx


def small_linear_fwd(x2d, W, bias, act=0):
    """act(x [M, K] @ W[N, K]^T + bias) in f32, any N >= 1 (act 1 = relu)."""
    _need_cuda(x2d, W, bias)
    M, K = x2d.shape
    N = W.shape[0]
    assert x2d.dtype == torch.float32 and W.dtype == torch.float32 and x2d.is_contiguous() and W.is_contiguous()
    y = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
    _lib.call("sarssl_small_linear_fwd", _p(x2d), _p(W), _p(bias), c_int(M), c_int(N), c_int(K), c_int(act), _p(y), _stream())
    return y


def small_linear_bwd(dy2d, y2d, x2d, W, act, dW, db, need_dx=True):
    """-> dx [M, K] (or None); dW += dz^T x, db += column sums of dz, dz = dy (act 0) / dy * [y > 0] (act 1)."""
    M, K = x2d.shape
    N = W.shape[0]
    dy2d = dy2d.contiguous().float()
    dz = torch.empty((M, N), dtype=torch.float32, device=x2d.device) if act == 1 else None
    dx = torch.empty((M, K), dtype=torch.float32, device=x2d.device) if need_dx else None
    _lib.call("sarssl_small_linear_bwd", _p(dy2d), _p(y2d), _p(x2d), _p(W), c_int(M), c_int(N), c_int(K), c_int(act), _p(dz), _p(dx), _p(dW), _p(db),
              _stream())
    return dx


# ---- hybrid numeric mode (csrc/hybrid.hip, sarssl_gemm_split): f32 tensors as fp16 pairs ------------------------------------------------
class Pair:
    """An f32 tensor [M, d] given as two fp16 tensors, hi = fp16(x) and lo = fp16(x - hi) (22 significant bits): the form in which the
    hybrid mode's f32 activations enter the matrix cores (gemm_split).  ``hi`` alone is the fp16 rounding of x - what the backward pass
    keeps as the weight-gradient operand."""
    __slots__ = ("hi", "lo", "__dict__")

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo

    shape = property(lambda self: self.hi.shape)
    device = property(lambda self: self.hi.device)
    dtype = torch.float32

    def float(self):
        return self.hi.float() + self.lo.float() if self.lo is not None else self.hi.float()


def gemm_split(A, B, B_lo, *, M, N, K, out=None, out_dtype=None, ldc=None, out_scale=1.0, bias=None, act=0, resid=None, ldr=0, res_scale=1.0,
               preact=None, p_drop=0.0, seed=0):
    """C = epilogue(A B^T + A B_lo^T [+ A_lo B^T]) - nn.Linear forward of the hybrid mode.  A: Pair (f32 activation) or an fp16 tensor
    [M, K]; B, B_lo: fp16 [N, K] (the weight's hi / lo shadows); C / resid / preact: fp16 or f32."""
    pair = isinstance(A, Pair)
    Ah, Al = (A.hi, A.lo) if pair else (A, None)                 # (a Pair without its lo half contracts as the fp16 tensor hi)
    _need_cuda(Ah, Al, B, B_lo, out, bias, resid, preact)
    assert Ah.dtype == torch.float16 and B.dtype == torch.float16 and (B_lo is None or (B_lo.dtype == torch.float16 and B_lo.stride(0) == B.stride(0)))
    assert Ah.stride(1) == 1 and B.stride(1) == 1 and (Al is None or Al.stride(0) == Ah.stride(0))
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or torch.float32, device=Ah.device)
    if ldc is None:
        ldc = out.stride(0)
    assert out.dtype in (torch.float16, torch.float32) and (resid is None or resid.dtype == out.dtype) and (preact is None or preact.dtype == out.dtype)
    with _Timed("gemm_split[%d,%d,%d x%d %s]" % (M, N, K, 1 + (Al is not None) + (B_lo is not None), str(out.dtype)[6:]) if _prof_shapes and _prof is not None else None):
        _lib.call("sarssl_gemm_split", _p(Ah), _p(Al), _p(B), _p(B_lo), _p(out), c_int(dt(out)), c_int(M), c_int(N), c_int(K),
                  c_long(Ah.stride(0)), c_long(B.stride(0)), c_long(ldc), c_float(out_scale), _p(bias), c_int(act), _p(resid), c_long(ldr),
                  c_float(res_scale), _p(preact), c_float(p_drop), c_ulonglong(seed), _stream())
    return out


def split_pair(src, want_hi=True, hi=None, lo=None):
    """src (f32 | fp16 | bf16, contiguous) -> Pair(hi = fp16(src), lo = fp16(src - hi)); want_hi False: the lo part only (-> tensor)."""
    _need_cuda(src, hi, lo)
    assert src.is_contiguous() and src.numel() % 4 == 0
    if lo is None:
        lo = torch.empty(src.shape, dtype=torch.float16, device=src.device)
    if want_hi and hi is None:
        hi = torch.empty(src.shape, dtype=torch.float16, device=src.device)
    _lib.call("sarssl_split_pair", _p(src), c_int(dt(src)), c_long(src.numel()), _p(hi) if want_hi else c_void_p(0), _p(lo), _stream())
    return Pair(hi, lo) if want_hi else lo


def layernorm_fwd_pair(x2d, gamma, beta, eps=1e-5, save=True, want32=False, out=None, want_lo=True):
    """LayerNorm of f32 rows -> (Pair, stats[, y32]).  out = (hi, lo): fp16 [M, d] views with a common row stride to write into.
    want_lo False: the hi half only (Pair.lo is None) - for a consumer that contracts the fp16 rounding of the result."""
    M, d = x2d.shape
    assert x2d.dtype == torch.float32
    if out is not None:
        hi, lo = out
        assert hi.dtype == torch.float16 and lo.dtype == torch.float16 and hi.stride(0) == lo.stride(0) and hi.stride(1) == 1 and lo.stride(1) == 1
    else:
        hi = torch.empty((M, d), dtype=torch.float16, device=x2d.device)
        lo = torch.empty((M, d), dtype=torch.float16, device=x2d.device) if want_lo else None
    y32 = torch.empty((M, d), dtype=torch.float32, device=x2d.device) if want32 else None
    stats = torch.empty((2, M), dtype=torch.float32, device=x2d.device) if save else None
    _lib.call("sarssl_layernorm_fwd_pair", _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(d), _p(gamma), _p(beta), c_float(eps), _p(hi), _p(lo),
              c_long(hi.stride(0)), _p(y32), c_long(d), _p(stats[0]) if save else c_void_p(0), _p(stats[1]) if save else c_void_p(0), _stream())
    return (Pair(hi, lo), stats, y32) if want32 else (Pair(hi, lo), stats)


def layernorm_fwd2_pair(x2d, ga, ba, epsa, gb, bb, epsb, out=None, want_lo=True):
    """y = LN_a(x) (f32), z = LN_b(y) (Pair; want_lo False: hi half only) in one launch -> (y, stats_a, z, stats_b)."""
    M, d = x2d.shape
    assert x2d.dtype == torch.float32
    if out is None:
        out = torch.empty((M, d), dtype=torch.float32, device=x2d.device)
    hi = torch.empty((M, d), dtype=torch.float16, device=x2d.device)
    lo = torch.empty((M, d), dtype=torch.float16, device=x2d.device) if want_lo else None
    sa = torch.empty((2, M), dtype=torch.float32, device=x2d.device)
    sb = torch.empty((2, M), dtype=torch.float32, device=x2d.device)
    _lib.call("sarssl_layernorm_fwd2_pair", _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(d), _p(ga), _p(ba), c_float(epsa), _p(out),
              c_long(out.stride(0)), _p(sa[0]), _p(sa[1]), _p(gb), _p(bb), c_float(epsb), _p(hi), _p(lo), c_long(d), _p(sb[0]), _p(sb[1]), _stream())
    return out, sa, Pair(hi, lo), sb


def layernorm_bwd_stream(dy2d, x2d, gamma, stats, resid=None, dgamma=None, dbeta=None, out=None, drop=None, copy16=False):
    """LayerNorm backward on the f32 stream (x, resid, result f32; dy bf16 or f32).  drop = (p, seed, gscale) / copy16: also returns the
    bf16 tensor out * dropout_mask * gscale (copy16: p = 0, gscale = 1 - the plain bf16 copy) -> (out, out16)."""
    M, d = x2d.shape
    assert x2d.dtype == torch.float32 and dy2d.dtype in (torch.bfloat16, torch.float32) and (resid is None or resid.dtype == torch.float32)
    if out is None:
        out = torch.empty((M, d), dtype=torch.float32, device=x2d.device)
    part = None
    if dgamma is not None:
        fn = _lib.lib().sarssl_layernorm_bwd_workspace_bytes
        fn.restype = c_long
        if _ln_batch is not None:
            part = torch.empty((fn(c_long(M), c_int(d)) // 4,), dtype=torch.float32, device=x2d.device)
            _ln_batch.append((part, int(_lib.lib().sarssl_layernorm_bwd_nparts(c_long(M))), d, dgamma, dbeta))
            dgamma = dbeta = None
        else:
            part = workspace(fn(c_long(M), c_int(d)), x2d.device, "ln_part")
    out2 = None
    p, seed, gscale = drop if drop is not None else (0.0, 0, 1.0)
    if drop is not None or copy16:
        out2 = torch.empty((M, d), dtype=torch.bfloat16, device=x2d.device)
    _lib.call("sarssl_layernorm_bwd_stream", _p(dy2d), c_int(dt(dy2d)), c_long(dy2d.stride(0)), _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(d),
              _p(gamma), _p(stats[0]), _p(stats[1]), _p(resid), c_long(resid.stride(0) if resid is not None else 0), _p(out), c_long(out.stride(0)),
              _p(dgamma), _p(dbeta), _p(part), _p(out2), c_float(p), c_ulonglong(seed), c_float(gscale), _stream())
    return (out, out2) if out2 is not None else out


# ---- fused feed-forward module (csrc/ffn2.hip) ------------------------------------------------------------------------------------
def ffn2_supported(M, d, dtype):
    return dtype in _16 and bool(_lib.lib().sarssl_ffn2_supported(c_long(M), c_int(d)))


def ffn_pack(jobs):
    """jobs: [(src 2-D view (N, K) - any strides, e.g. ``W`` or ``W.t()`` -, dst flat 16-bit tensor of N * K elements)]: every matrix
    into MFMA fragment order (include/sarssl_hip.h) in ONE launch (<= 64 per launch)."""
    for i in range(0, len(jobs), 64):
        part = jobs[i:i + 64]
        n = len(part)
        _need_cuda(*[t for j in part for t in j])
        for src, dst in part:
            assert src.dim() == 2 and src.element_size() == 2 and dst.element_size() == 2 and dst.numel() == src.numel() and dst.is_contiguous()
        VP, I, L = c_void_p * n, c_int * n, c_long * n
        _lib.call("sarssl_ffn_pack", VP(*[s.data_ptr() for s, _ in part]), VP(*[d.data_ptr() for _, d in part]),
                  I(*[s.shape[0] for s, _ in part]), I(*[s.shape[1] for s, _ in part]), L(*[s.stride(0) for s, _ in part]),
                  L(*[s.stride(1) for s, _ in part]), c_int(n), _stream())


def ffn2_fwd(ln, w1p, w2p, b1, b2, resid, d, p1=0.0, s1=0, p2=0.0, s2=0, out_scale=1.0, out=None, ln_in=None):
    """-> (y [M, d], preact [M, 4d], hidden [M, 4d]); y = resid + out_scale * drop2(W2 drop1(swish(W1 ln + b1)) + b2).
    ln_in = (x, gamma, beta, eps): the LayerNorm runs in the launch (``ln`` is ignored) -> (y, preact, hidden, ln, stats)."""
    src = ln if ln_in is None else ln_in[0]
    _need_cuda(src, w1p, w2p, b1, b2, resid, out)
    M = src.shape[0]
    pre = torch.empty((M, 4 * d), dtype=src.dtype, device=src.device)
    hid = torch.empty((M, 4 * d), dtype=src.dtype, device=src.device)
    if out is None:
        out = torch.empty((M, d), dtype=src.dtype, device=src.device)
    lno = stats = None
    if ln_in is not None:
        x, gamma, beta, eps = ln_in
        lno = torch.empty((M, d), dtype=x.dtype, device=x.device)
        stats = torch.empty((2, M), dtype=torch.float32, device=x.device)
    with _Timed("ffn2_fwd[d%d%s]" % (d, " +ln" if ln_in is not None else "") if _prof_shapes and _prof is not None else None):
        _lib.call("sarssl_ffn2_fwd", _p(ln), c_long(ln.stride(0) if ln is not None else 0), _p(w1p), _p(w2p), _p(b1), _p(b2), _p(pre), _p(hid), _p(out),
                  c_long(out.stride(0)), _p(resid), c_long(resid.stride(0) if resid is not None else 0), c_long(M), c_int(d), c_float(p1), c_ulonglong(s1),
                  c_float(p2), c_ulonglong(s2), c_float(out_scale),
                  _p(ln_in[0]) if ln_in is not None else c_void_p(0), c_long(ln_in[0].stride(0) if ln_in is not None else 0),
                  _p(ln_in[1]) if ln_in is not None else c_void_p(0), _p(ln_in[2]) if ln_in is not None else c_void_p(0),
                  c_float(ln_in[3] if ln_in is not None else 0.0), _p(lno), _p(stats[0]) if stats is not None else c_void_p(0),
                  _p(stats[1]) if stats is not None else c_void_p(0), c_int(dt(src)), _stream())
    if ln_in is not None:
        return out, pre, hid, lno, stats
    return out, pre, hid


def ffn2h_supported(M, d):
    return bool(_lib.lib().sarssl_ffn2h_supported(c_long(M), c_int(d)))


def ffn2h_fwd(x, gamma, beta, eps, w1h, w1l, w2h, w2l, b1, b2, d, p1=0.0, s1=0, p2=0.0, s2=0, out_scale=1.0, out=None, act_pair=False):
    """Hybrid mode, d = 256: y = x + out_scale * drop2(W2 drop1(swish(W1 LN(x) + b1)) + b2) on the f32 stream in one launch
    -> (y f32, preact fp16 [M, 4d], hidden fp16 [M, 4d], ln_hi fp16 [M, d], stats f32 [2, M])."""
    _need_cuda(x, w1h, w1l, w2h, w2l, b1, b2, out)
    M = x.shape[0]
    assert x.dtype == torch.float32 and x.stride(1) == 1
    pre = torch.empty((M, 4 * d), dtype=torch.float16, device=x.device)
    hid = torch.empty((M, 4 * d), dtype=torch.float16, device=x.device)
    lno = torch.empty((M, d), dtype=torch.float16, device=x.device)
    stats = torch.empty((2, M), dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty((M, d), dtype=torch.float32, device=x.device)
    with _Timed("ffn2h_fwd[d%d]" % d if _prof_shapes and _prof is not None else None):
        _lib.call("sarssl_ffn2h_fwd", _p(x), c_long(x.stride(0)), _p(gamma), _p(beta), c_float(eps), _p(lno), _p(stats[0]), _p(stats[1]),
                  _p(w1h), _p(w1l), _p(w2h), _p(w2l), _p(b1), _p(b2), _p(pre), _p(hid), _p(out), c_long(out.stride(0)), c_long(M), c_int(d),
                  c_float(p1), c_ulonglong(s1), c_float(p2), c_ulonglong(s2), c_float(out_scale), c_int(1 if act_pair else 0), _stream())
    return out, pre, hid, lno, stats


def ffn2_bwd(dz2, w2tp, w1tp, preact, d, p1=0.0, s1=0, ln_bwd=None):
    """-> (dln [M, d], dh [M, 4d]) in the gradient dtype of dz2.
    ln_bwd = (x, gamma, stats, resid, dgamma, dbeta, drop): the LayerNorm backward of the module's first layer runs in the launch's
    epilogue (hip.layernorm_bwd on dln) -> (dx, dh) or ((dx, dx2), dh) with drop = (p, seed, gscale); dgamma / dbeta are folded from
    per-tile partials - with the block's other LayerNorms inside ``ln_reduce_batched()``, else right away."""
    _need_cuda(dz2, w2tp, w1tp, preact)
    M = dz2.shape[0]
    dh = torch.empty((M, 4 * d), dtype=dz2.dtype, device=dz2.device)
    dln = torch.empty((M, d), dtype=dz2.dtype, device=dz2.device)
    x = gamma = stats = resid = dgamma = dbeta = drop = dx2 = part = None
    flush_now = None
    if ln_bwd is not None:
        x, gamma, stats, resid, dgamma, dbeta, drop = ln_bwd
        assert x.dtype == preact.dtype
        if dgamma is not None:
            part = torch.empty(((M // 64) * 2 * d,), dtype=torch.float32, device=dz2.device)
            if _ln_batch is not None:
                _ln_batch.append((part, M // 64, d, dgamma, dbeta))
            else:
                flush_now = (part, M // 64, d, dgamma, dbeta)
        if drop is not None:
            dx2 = torch.empty((M, d), dtype=dz2.dtype, device=dz2.device)
    p2, s2, g2 = drop if drop is not None else (0.0, 0, 1.0)
    with _Timed("ffn2_bwd[d%d%s]" % (d, " +ln" if ln_bwd is not None else "") if _prof_shapes and _prof is not None else None):
        _lib.call("sarssl_ffn2_bwd", _p(dz2), c_long(dz2.stride(0)), _p(w2tp), _p(w1tp), _p(preact), _p(dh), _p(dln), c_long(dln.stride(0)),
                  c_long(M), c_int(d), c_float(p1), c_ulonglong(s1), _p(x), c_long(x.stride(0) if x is not None else 0), _p(gamma),
                  _p(stats[0]) if stats is not None else c_void_p(0), _p(stats[1]) if stats is not None else c_void_p(0), _p(resid),
                  c_long(resid.stride(0) if resid is not None else 0), _p(dx2), c_float(p2), c_ulonglong(s2), c_float(g2), _p(part),
                  c_int(dt_ga(dz2, preact)), _stream())
    if flush_now is not None:
        import ctypes
        _lib.call("sarssl_ln_param_reduce_multi", (ctypes.c_void_p * 1)(flush_now[0].data_ptr()), (ctypes.c_int * 1)(flush_now[1]),
                  (ctypes.c_int * 1)(flush_now[2]), (ctypes.c_void_p * 1)(flush_now[3].data_ptr()), (ctypes.c_void_p * 1)(flush_now[4].data_ptr()),
                  c_int(1), _stream())
    if dx2 is not None:
        return (dln, dx2), dh
    return dln, dh


# ---- row-tile-resident Linear layers of the d = 256 blocks (csrc/lin256.hip) ------------------------------------------------------------
def lin256_supported(M, N, K, dtype):
    return dtype in _16 and bool(_lib.lib().sarssl_lin256_supported(c_long(M), c_int(N), c_int(K)))


def _ln_partial_register(M, d, dgamma, dbeta, device):
    """partials [M / 64][2][d] of a fused LayerNorm backward: folded with the block's other LayerNorms inside ln_reduce_batched(), else
    by the returned callable right behind the launch."""
    part = torch.empty(((M // 64) * 2 * d,), dtype=torch.float32, device=device)
    item = (part, M // 64, d, dgamma, dbeta)
    if _ln_batch is not None:
        _ln_batch.append(item)
        return part, None

    def flush():
        import ctypes
        _lib.call("sarssl_ln_param_reduce_multi", (ctypes.c_void_p * 1)(item[0].data_ptr()), (ctypes.c_int * 1)(item[1]), (ctypes.c_int * 1)(item[2]),
                  (ctypes.c_void_p * 1)(item[3].data_ptr()), (ctypes.c_void_p * 1)(item[4].data_ptr()), c_int(1), _stream())
    return part, flush


def lin256_fwd(a, wp, bias, N, K, resid=None, p_drop=0.0, seed=0, out_scale=1.0, out=None, ln_in=None):
    """y [M, N] = resid + out_scale * drop(a W^T + bias), wp = pack(W [N, K]).  ln_in = (x, gamma, beta, eps): a = LayerNorm(x) formed in the
    launch -> (y, ln, stats)."""
    src = a if ln_in is None else ln_in[0]
    _need_cuda(src, wp, bias, resid, out)
    M = src.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=src.dtype, device=src.device)
    lno = stats = None
    if ln_in is not None:
        lno = torch.empty((M, K), dtype=src.dtype, device=src.device)
        stats = torch.empty((2, M), dtype=torch.float32, device=src.device)
    with _Timed("lin256_fwd[%d,%d%s]" % (N, K, " +ln" if ln_in is not None else "") if _prof_shapes and _prof is not None else None):
        _lib.call("sarssl_lin256_fwd", _p(a), c_long(a.stride(0) if a is not None else 0), _p(wp), _p(bias), _p(out), c_long(out.stride(0)), _p(resid),
                  c_long(resid.stride(0) if resid is not None else 0), c_long(M), c_int(N), c_int(K), c_float(p_drop), c_ulonglong(seed),
                  c_float(out_scale), _p(ln_in[0]) if ln_in is not None else c_void_p(0), c_long(ln_in[0].stride(0) if ln_in is not None else 0),
                  _p(ln_in[1]) if ln_in is not None else c_void_p(0), _p(ln_in[2]) if ln_in is not None else c_void_p(0),
                  c_float(ln_in[3] if ln_in is not None else 0.0), _p(lno), _p(stats[0]) if stats is not None else c_void_p(0),
                  _p(stats[1]) if stats is not None else c_void_p(0), c_int(dt(src)), _stream())
    return (out, lno, stats) if ln_in is not None else out


def lin256_bwd(dy, wtp, N, K, ln_bwd=None):
    """dx [M, N] = dy [M, K] W (wtp = pack(W^T [N, K])).  ln_bwd = (x, gamma, stats, resid, dgamma, dbeta, drop): the LayerNorm backward of
    the layer's input runs in the epilogue -> dx or (dx, dx2)."""
    _need_cuda(dy, wtp)
    M = dy.shape[0]
    dx = torch.empty((M, N), dtype=dy.dtype, device=dy.device)
    x = gamma = stats = resid = dx2 = part = flush = None
    drop = None
    if ln_bwd is not None:
        x, gamma, stats, resid, dgamma, dbeta, drop = ln_bwd
        if dgamma is not None:
            part, flush = _ln_partial_register(M, N, dgamma, dbeta, dy.device)
        if drop is not None:
            dx2 = torch.empty((M, N), dtype=dy.dtype, device=dy.device)
    p2, s2, g2 = drop if drop is not None else (0.0, 0, 1.0)
    with _Timed("lin256_bwd[%d,%d%s]" % (N, K, " +ln" if ln_bwd is not None else "") if _prof_shapes and _prof is not None else None):
        _lib.call("sarssl_lin256_bwd", _p(dy), c_long(dy.stride(0)), _p(wtp), _p(dx), c_long(dx.stride(0)), c_long(M), c_int(N), c_int(K), _p(x),
                  c_long(x.stride(0) if x is not None else 0), _p(gamma), _p(stats[0]) if stats is not None else c_void_p(0),
                  _p(stats[1]) if stats is not None else c_void_p(0), _p(resid), c_long(resid.stride(0) if resid is not None else 0), _p(dx2),
                  c_float(p2), c_ulonglong(s2), c_float(g2), _p(part), c_int(dt_ga(dy, x) if x is not None else dt(dy)), _stream())
    if flush is not None:
        flush()
    return (dx, dx2) if dx2 is not None else dx


_splitk_batch = None


def gemm_group_tn(items):
    """items: [(dy [K,M] bf16, x [K,N] bf16, out f32 [M,N] view, split, bias_out f32 [M] or None)] - the split-K partial sums of every
    out_q += dy_q^T x_q in ONE launch (csrc/gemm.hip, gemm_group_tn_kernel) and, where bias_out is given, the per-slice column sums of
    dy_q (the layer's bias gradient) from the same launch; the partials join the open splitk_batched() batch and are folded into the
    gradient buffers when it closes.  Returns False (nothing launched) when a shape is ragged or no batch is open."""
    import ctypes
    n = len(items)
    if _splitk_batch is None or n == 0 or n > 12:
        return False
    for dy, x, out, split, _ in items:
        if not (dy.dtype == torch.bfloat16 and x.dtype == items[0][1].dtype and x.dtype in _16 and out.dtype == torch.float32 and dy.stride(1) == 1
                and x.stride(1) == 1 and out.shape[1] % 4 == 0 and out.stride(0) % 4 == 0 and dy.shape[1] % 128 == 0
                and x.shape[1] % 128 == 0):
            return False
    _need_cuda(*[t for it in items for t in it[:3]])
    ws = [torch.empty((it[3] * it[0].shape[1] * it[1].shape[1],), dtype=torch.float32, device=it[0].device) for it in items]
    cs = [torch.empty((it[3] * it[0].shape[1],), dtype=torch.float32, device=it[0].device) if it[4] is not None else None for it in items]
    arr = lambda ty, vals: (ty * n)(*vals)
    split_out = (ctypes.c_int * n)()
    with _Timed("call:sarssl_gemm_group_tn"):
        rc = _lib.lib().sarssl_gemm_group_tn(
            arr(ctypes.c_void_p, [it[0].data_ptr() for it in items]), arr(ctypes.c_void_p, [it[1].data_ptr() for it in items]),
            arr(ctypes.c_void_p, [w.data_ptr() for w in ws]), arr(ctypes.c_int, [it[0].shape[1] for it in items]),
            arr(ctypes.c_int, [it[1].shape[1] for it in items]), arr(ctypes.c_int, [it[0].shape[0] for it in items]),
            arr(ctypes.c_long, [it[0].stride(0) for it in items]), arr(ctypes.c_long, [it[1].stride(0) for it in items]),
            arr(ctypes.c_int, [it[3] for it in items]), split_out, arr(ctypes.c_void_p, [c.data_ptr() if c is not None else None for c in cs]),
            c_int(n), c_int(dt(items[0][1])), _stream())
    _lib.ncalls += 1
    if rc == 1:
        return False
    _lib.check(rc, "sarssl_gemm_group_tn")
    for q, (dy, x, out, split, bias_out) in enumerate(items):
        ns, Mq = int(split_out[q]), dy.shape[1]
        _splitk_batch.append((ws[q], ns, Mq, x.shape[1], out, out.stride(0)))
        if bias_out is not None:
            _splitk_batch.append((cs[q][:ns * Mq], ns, 1, Mq, bias_out, Mq))
    while len(_splitk_batch) > 32:
        splitk_flush_first(32)
    return True


def splitk_flush_first(k):
    """Fold the first k queued partial-sum buffers now (the reduce launch takes at most 32 problems)."""
    global _splitk_batch
    head, _splitk_batch = _splitk_batch[:k], _splitk_batch[k:]
    _splitk_flush_items(head, len(head))


_splitk_ctx = None            # optional context-manager factory (tensors...) the flush launch runs under (engine._WgradSide)


def splitk_flush():
    global _splitk_batch
    if not _splitk_batch:
        return
    items = _splitk_batch
    _splitk_batch = []
    for i in range(0, len(items), 32):
        chunk = items[i:i + 32]
        if _splitk_ctx is not None:
            with _splitk_ctx(*[it[0] for it in chunk]):
                _splitk_flush_items(chunk, len(chunk))
        else:
            _splitk_flush_items(chunk, len(chunk))


def _splitk_flush_items(items, n):
    import ctypes
    # The reduce launch does a non-atomic read-modify-write of every destination from parallel workgroups: two problems with the same
    # destination (a tied / shared parameter, one bias accumulated twice in a block) must not share a launch (advisor, round 3)
    seen = set()
    for i, it in enumerate(items):
        dst = it[4].data_ptr()
        assert dst % 16 == 0, "split-K fold destinations must be 16-byte aligned"
        if dst in seen:
            _splitk_flush_items(items[:i], i)
            _splitk_flush_items(items[i:], n - i)
            return
        seen.add(dst)
    ws = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
    C = (ctypes.c_void_p * n)(*[it[4].data_ptr() for it in items])
    ns = (ctypes.c_int * n)(*[it[1] for it in items])
    Ms = (ctypes.c_int * n)(*[it[2] for it in items])
    Ns = (ctypes.c_int * n)(*[it[3] for it in items])
    ld = (ctypes.c_long * n)(*[it[5] for it in items])
    _lib.call("sarssl_splitk_reduce_multi", ws, ns, Ms, Ns, C, ld, c_int(n), _stream())


class splitk_batched:
    """Context manager: split-K (weight-gradient) products issued inside keep their partial sums in their own buffers and are
    reduced into the gradient buffers by ONE launch at exit, instead of one reduce launch per product.  ``flush_ctx``: context
    manager factory the reduce launch is issued under (a side stream)."""

    def __init__(self, flush_ctx=None):
        self._ctx = flush_ctx

    def __enter__(self):
        global _splitk_batch, _splitk_ctx
        self._outer = _splitk_batch
        if _splitk_batch is None:
            _splitk_batch = []
            _splitk_ctx = self._ctx
        return self

    def __exit__(self, *exc):
        global _splitk_batch, _splitk_ctx
        if self._outer is None:
            if exc[0] is None:
                splitk_flush()
            _splitk_batch = None
            _splitk_ctx = None
        return False


def fp8_quantize(x2d, transpose=False):
    """x [rows, cols] (f32 | bf16, row-strided) -> (q uint8 e4m3fn [rows, cols] or, transposed, [cols, rows]; inv_scale f32[1] on the
    device) with the per-tensor scale 448 / amax chosen on the device (csrc/gemm_fp8.hip)."""
    _need_cuda(x2d)
    rows, cols = x2d.shape
    assert x2d.stride(1) == 1
    q = torch.empty((cols, rows) if transpose else (rows, cols), dtype=torch.uint8, device=x2d.device)
    ws = torch.empty((2,), dtype=torch.float32, device=x2d.device)            # [amax bits | inv_scale]
    _lib.call("sarssl_fp8_quantize", _p(x2d), c_int(dt(x2d)), c_long(rows), c_long(cols), c_long(x2d.stride(0)), _p(q),
              c_long(q.stride(0)), _p(ws[0:1]), _p(ws[1:2]), c_int(1 if transpose else 0), _stream())
    return q, ws[1:2]


def gemm_fp8(A8, sa, B8, sb, *, M, N, K, out=None, out_dtype=torch.bfloat16, ldc=None, alpha=1.0, out_scale=1.0, bias=None, act=0,
             resid=None, ldr=0, res_scale=1.0, preact=None, aux=None, aux_act=0, p_drop=0.0, seed=0):
    """C [M,N] = epilogue(alpha * sa * sb * A8 [M,K] @ B8 [N,K]^T), fp8 operands with device-resident dequantisation scales."""
    _need_cuda(A8, B8, out, bias, resid, preact, aux)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A8.device)
    if ldc is None:
        ldc = out.stride(0)
    _lib.call("sarssl_gemm_fp8", _p(A8), _p(B8), _p(sa), _p(sb), _p(out), c_int(dt(out)), c_int(M), c_int(N), c_int(K),
              c_long(A8.stride(0)), c_long(B8.stride(0)), c_long(ldc), c_float(alpha), c_float(out_scale), _p(bias), c_int(act),
              _p(resid), c_long(ldr), c_float(res_scale), _p(preact), _p(aux), c_int(aux_act), c_float(p_drop), c_ulonglong(seed),
              _stream())
    return out


def stft_frontend(sig, eps=1e-6, win_len=512, hop=256, nfft=512, ch_mode="M", masks=None, dtype=None):
    """(B, nsample, nch) f32|int16 -> (B*npair, 2, nfft/2, nt, 2) f32 (data_preprocess output); ch_mode 'M' pairs mic 0 with
    every other mic (npair = nch-1), 'MM' takes every mic pair (npair = nch(nch-1)/2).
    masks = (mp_u8 (B*npair, nt), mch_i32 (B*npair)): also the two encoders' masked inputs of `dtype` in the same pass (what mask_inputs
    would make of the result) -> (out, spec_in, spat_in)."""
    _need_cuda(sig)
    sig = sig.contiguous()
    nb, nsample, nch = sig.shape
    nt = (nsample - win_len) // hop + 1
    mode = {"M": 0, "MM": 1}[ch_mode]
    npair = nch - 1 if mode == 0 else nch * (nch - 1) // 2
    U = torch.empty((nb, nch, nfft // 2 + 1, nt, 2), dtype=torch.float32, device=sig.device)
    magsum = torch.empty((nb,), dtype=torch.float64, device=sig.device)
    out = torch.empty((nb * npair, 2, nfft // 2, nt, 2), dtype=torch.float32, device=sig.device)
    if masks is not None:
        mp, mch = masks
        _need_cuda(mp, mch)
        assert mp.dtype == torch.uint8 and mch.dtype == torch.int32 and mp.numel() == nb * npair * nt and mch.numel() == nb * npair
        spec = torch.empty((nb * npair, nfft // 2, nt, 4), dtype=dtype, device=sig.device)
        spat = torch.empty_like(spec)
        _lib.call("sarssl_stft_frontend_pairs_masked", _p(sig), c_int(dt(sig)), c_int(nb), c_long(nsample), c_int(nch), c_int(win_len),
                  c_int(hop), c_int(nfft), c_int(nt), c_float(eps), c_int(mode), _p(U), _p(magsum), _p(out), _p(mp), _p(mch), _p(spec),
                  _p(spat), c_int(dt(spec)), _stream())
        return out, spec, spat
    _lib.call("sarssl_stft_frontend_pairs", _p(sig), c_int(dt(sig)), c_int(nb), c_long(nsample), c_int(nch), c_int(win_len),
              c_int(hop), c_int(nfft), c_int(nt), c_float(eps), c_int(mode), _p(U), _p(magsum), _p(out), _stream())
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


def istft(spec, center=False, win_len=512, hop=256, nfft=512):
    """complex64 (B, nfft/2+1, nt, nch) -> (B, nsample, nch) f32 like ISTFT.forward (torch.istft, rectangular window)."""
    _need_cuda(spec)
    sp = torch.view_as_real(spec.to(torch.complex64).contiguous())
    nb, nf, nt, nch, _ = sp.shape
    assert nf == nfft // 2 + 1
    nsample = (nt - 1) * hop if center else (nt + 1) * hop
    ws = _f32ws(nb * nch * nt * nfft, spec.device, "istft")
    sig = torch.empty((nb, nsample, nch), dtype=torch.float32, device=spec.device)
    _lib.call("sarssl_istft", _p(sp), c_int(nb), c_int(nch), c_int(nt), c_int(win_len), c_int(hop), c_int(nfft),
              c_int(1 if center else 0), _p(ws), _p(sig), _stream())
    return sig


# ------------------------------------------------------------------------------------------------
# conv stem
def _f32ws(n, device, tag):
    return workspace(4 * n, device, tag).view(torch.float32)[:n]


def _f64ws(n, device, tag):
    return workspace(8 * n + 8, device, tag)[: 8 * n].view(torch.float64)


def mask_inputs(x, mp_u8, mch_i32, mode, dtype):
    """x (B,2,F,T,2) f32 -> spec_in, spat_in (B,F,T,4) of `dtype` (csrc/stem.hip)."""
    _need_cuda(x)
    B, _, F, T, _ = x.shape
    spec = torch.empty((B, F, T, 4), dtype=dtype, device=x.device)
    spat = torch.empty((B, F, T, 4), dtype=dtype, device=x.device)
    _lib.call("sarssl_mask_inputs", _p(x), _p(mp_u8), _p(mch_i32), c_int(B), c_int(F), c_int(T), c_int(mode), _p(spec), _p(spat),
              c_int(_DT[dtype]), _stream())
    return spec, spat


def stem_c1_fwd(a0, W1, want_stats=False):
    """-> y1, or (y1, sums f64[128]) with the BatchNorm sums of y1 accumulated in the same pass."""
    npix = a0.numel() // 4
    y = torch.empty(a0.shape[:-1] + (64,), dtype=a0.dtype, device=a0.device)
    sums = _sums(128, a0.device) if want_stats else None
    _lib.call("sarssl_stem_c1_fwd", _p(a0), _p(W1), c_long(npix), _p(y), _p(sums), c_int(dt(a0)), _stream())
    return (y, sums) if want_stats else y


def stem_c1_stats(a0, W1, keep_moments=False):
    """BatchNorm sums f64[128] = [sum | sum of squares] of y1 = W1 a0 from the 4 + 10 first / second moments of a0 (y1 is not formed).
    keep_moments: also return the moments f64[14] in memory of their own (the backward pass reads them, see conv3x3_dgrad_c1red)."""
    npix = a0.numel() // 4
    sums = _sums(128, a0.device)
    mom = torch.empty(16, dtype=torch.float64, device=a0.device) if keep_moments else _sums(16, a0.device)      # (zeroed by the launch wrapper)
    _lib.call("sarssl_stem_c1_stats", _p(a0), c_long(npix), _p(W1), _p(mom), _p(sums), c_int(dt(a0)), _stream())
    return (sums, mom) if keep_moments else sums


def stem_c1_bn_affine(a0, W1, gamma, beta, running_mean, running_var, nbt, eps=1e-5, momentum=0.1):
    """Training-mode BatchNorm(1) affine of y1 = W1 a0 from the input's moments: -> (aff (4,64) f32 = scale | shift | mean | rstd, moments
    f64[14]); running statistics and the batch counter are updated like sarssl_bn_finalize does."""
    npix = a0.numel() // 4
    mom = torch.empty(16, dtype=torch.float64, device=a0.device)               # memory of its own (the backward pass reads it); zeroed by the launch wrapper
    aff = torch.empty((4, 64), dtype=torch.float32, device=a0.device)
    _lib.call("sarssl_stem_c1_stats_affine", _p(a0), c_long(npix), _p(W1), _p(mom), _p(gamma), _p(beta), c_float(eps), c_float(momentum),
              _p(running_mean), _p(running_var), _p(nbt), _p(aff), c_int(dt(a0)), _stream())
    return aff, mom


def conv3x3_dgrad_c1red(dy, w_tap_dgrad, a0, W1, aff, mom, train, dW1, dgamma, dbeta):
    """Data gradient of the first 3x3 convolution consumed in its epilogue: dW1 (64,4,1,1) / dgamma / dbeta (64) += the first stem
    layer's parameter gradients; the 64-channel gradient tensor is never stored (mom = the input's moments from stem_c1_stats)."""
    _need_cuda(dy, w_tap_dgrad, a0, W1, aff, mom)
    B, F, T, _ = a0.shape
    assert dy.dtype == torch.bfloat16 and a0.dtype in _16 and dy.shape == (B, F, T, 64)
    red = _sums(644, a0.device)
    with _Timed("conv3x3_dgrad_c1red"):
        _lib.call("sarssl_conv3x3_dgrad_c1red", _p(dy), _p(w_tap_dgrad), _p(a0), _p(W1), _p(aff[0]), _p(aff[1]), c_int(B), c_int(F), c_int(T),
                  _p(red), c_int(dt(a0)), _stream())
    _lib.call("sarssl_stem_c1_bwd_finalize_mom", _p(red), _p(mom), _p(W1), c_long(B * F * T), _p(aff), c_int(1 if train else 0),
              _p(dW1), _p(dgamma), _p(dbeta), _stream())
    return True


def stem_c1_bwd_a0(dz1, a0, W1, aff, train, dW1, dgamma, dbeta):
    """stem_c1_bwd with y1 = W1 a0 recomputed from the input instead of read (bf16, pixel count a multiple of 64)."""
    npix = a0.numel() // 4
    assert a0.dtype in _16 and dz1.dtype == torch.bfloat16 and npix % 64 == 0
    ws = _sums(644, a0.device)
    _lib.call("sarssl_stem_c1_bwd_a0", _p(dz1), _p(a0), _p(W1), c_long(npix), _p(aff), c_int(1 if train else 0), _p(ws), _p(dW1),
              _p(dgamma), _p(dbeta), c_int(dt_ga(dz1, a0)), _stream())


def stem_c1_wgrad(dy1, a0, grad_out):
    """grad_out (64,4,1,1) f32 += dW1."""
    npix = a0.numel() // 4
    ws = _f64ws(256, a0.device, "c1w")
    _lib.call("sarssl_stem_c1_wgrad", _p(dy1), _p(a0), c_long(npix), _p(ws), c_int(dt_ga(dy1, a0)), _stream())
    _lib.call("sarssl_f64_accum", _p(ws), _p(grad_out), c_int(256), c_float(1.0), _stream())


def stem_c1_wgrad_bn(dz1, y1, a0, aff, red, train, grad_out):
    """grad_out (64,4,1,1) f32 += dW1, from the gradient w.r.t. relu(bn1(y1)) - BatchNorm backward folded into the kernel."""
    npix = a0.numel() // 4
    ws = _f64ws(256, a0.device, "c1w")
    _lib.call("sarssl_stem_c1_wgrad_bn", _p(dz1), _p(y1), _p(a0), c_long(npix), _p(aff), _p(red), c_int(1 if train else 0), _p(ws),
              c_int(dt_ga(dz1, a0)), _stream())
    _lib.call("sarssl_f64_accum", _p(ws), _p(grad_out), c_int(256), c_float(1.0), _stream())


def stem_c1_bwd(dz1, y1, a0, aff, train, dW1, dgamma, dbeta):
    """One pass: dW1 (64,4,1,1) += ..., dgamma / dbeta (64) += BatchNorm(1) parameter gradients, from the gradient w.r.t.
    relu(bn1(y1))."""
    npix = a0.numel() // 4
    ws = _sums(644, a0.device)                 # (arena slice: already zero, no memset launch)
    assert y1.dtype == a0.dtype
    _lib.call("sarssl_stem_c1_bwd", _p(dz1), _p(y1), _p(a0), c_long(npix), _p(aff), c_int(1 if train else 0), _p(ws), _p(dW1),
              _p(dgamma), _p(dbeta), c_int(dt_ga(dz1, a0)), _stream())


def stem_c4_fwd(y3, W4, scale, shift, want_stats=False):
    """-> y4 (B,T,F,4), or (y4, sums f64[8]) with the BatchNorm sums of the stored y4 accumulated in the same pass."""
    B, F, T, _ = y3.shape
    y4 = torch.empty((B, T, F, 4), dtype=y3.dtype, device=y3.device)
    if want_stats:
        sums = _sums(8, y3.device)
        _lib.call("sarssl_stem_c4_fwd_stats", _p(y3), _p(W4), _p(scale), _p(shift), c_int(B), c_int(F), c_int(T), _p(y4), _p(sums),
                  c_int(dt(y3)), _stream())
        return y4, sums
    _lib.call("sarssl_stem_c4_fwd", _p(y3), _p(W4), _p(scale), _p(shift), c_int(B), c_int(F), c_int(T), _p(y4), c_int(dt(y3)), _stream())
    return y4


def stem_c4_fwd_pair(y3, W4, scale, shift, want_stats=False):
    """Hybrid mode: the 64 -> 4 convolution's f32 result as a Pair (hi = what stem_c4_fwd stores) -> Pair or (Pair, sums f64[8])."""
    B, F, T, _ = y3.shape
    hi = torch.empty((B, T, F, 4), dtype=y3.dtype, device=y3.device)
    lo = torch.empty((B, T, F, 4), dtype=y3.dtype, device=y3.device)
    sums = _sums(8, y3.device) if want_stats else None
    _lib.call("sarssl_stem_c4_fwd_pair", _p(y3), _p(W4), _p(scale), _p(shift), c_int(B), c_int(F), c_int(T), _p(hi), _p(lo), _p(sums),
              c_int(dt(y3)), _stream())
    return (Pair(hi, lo), sums) if want_stats else Pair(hi, lo)


def cl_affine_act_pair(x, C, aff, act):
    """act(aff[0][c] * x + aff[1][c]) on a channels-last Pair -> Pair."""
    zh, zl = torch.empty_like(x.hi), torch.empty_like(x.lo)
    _lib.call("sarssl_cl_affine_act_pair", _p(x.hi), _p(x.lo), c_long(x.hi.numel()), c_int(C), _p(aff[0]), _p(aff[1]), c_int(act), _p(zh), _p(zl),
              _stream())
    return Pair(zh, zl)


def stem_c4_bwd(y3, dy4, W4, aff):
    """-> g3 (B,F,T,64), red f64[384] = [dW4 (4x64) | s1 (64) | s2 (64)]."""
    B, F, T, _ = y3.shape
    g3 = torch.empty(y3.shape, dtype=dy4.dtype, device=y3.device)
    red = _sums(384, y3.device)
    _lib.call("sarssl_stem_c4_bwd", _p(y3), _p(dy4), _p(W4), _p(aff[0]), _p(aff[1]), _p(aff[2]), _p(aff[3]), c_int(B), c_int(F),
              c_int(T), _p(g3), _p(red), c_int(dt_ga(dy4, y3)), _stream())
    return g3, red


def stem_c4_bwd_two_phase(y3, dy4, W4, aff, train, pgrads=None):
    """-> dy3 (B,F,T,64) = gradient w.r.t. the third BatchNorm's input, red f64[384] = [dW4 | s1 | s2]; two kernels, no
    intermediate 64-channel tensor.  pgrads = (gW4 (4,64), dgamma, dbeta) f32 gradient buffers: accumulated from red by the second
    kernel."""
    B, F, T, _ = y3.shape
    red = _sums(384, y3.device)
    _lib.call("sarssl_stem_c4_bwd_sums", _p(y3), _p(dy4), _p(W4), _p(aff[0]), _p(aff[1]), _p(aff[2]), _p(aff[3]), c_int(B), c_int(F),
              c_int(T), _p(red), c_int(dt_ga(dy4, y3)), _stream())
    dy3 = torch.empty(y3.shape, dtype=dy4.dtype, device=y3.device)
    gw, dg, db = pgrads if pgrads is not None else (None, None, None)
    _lib.call("sarssl_stem_c4_bwd_apply_pg", _p(y3), _p(dy4), _p(W4), _p(aff[0]), _p(aff[1]), _p(aff[2]), _p(aff[3]), c_int(B), c_int(F),
              c_int(T), _p(red), c_int(1 if train else 0), _p(dy3), _p(gw), _p(dg), _p(db), c_int(dt_ga(dy4, y3)), _stream())
    return dy3, red


def f64_accum2(src, dst1, dst2):
    """dst1 += src[:n], dst2 += src[n:2n] (n = dst1.numel()) in one launch."""
    _lib.call("sarssl_f64_accum2", _p(src), _p(dst1), _p(dst2), c_int(dst1.numel()), _stream())


def conv_taps(W, dtype, gdtype=None):
    """(64,64,3,3) f32 weight -> (fwd [9][co][ci] in `dtype`, dgrad [9][ci][co] with flipped taps in `gdtype` (default: `dtype`)), one
    launch.  (fp16, bf16) is the pair of the fp16-forward mode: forward taps fp16, data-gradient taps bf16."""
    _need_cuda(W)
    assert W.shape == (64, 64, 3, 3) and W.dtype == torch.float32 and W.is_contiguous()
    gdtype = gdtype or dtype
    fwd = torch.empty((9, 64, 64), dtype=dtype, device=W.device)
    dgr = torch.empty((9, 64, 64), dtype=gdtype, device=W.device)
    if gdtype != dtype:
        assert (dtype, gdtype) == (torch.float16, torch.bfloat16)
        code = MIX16
    else:
        code = _DT[dtype]
    _lib.call("sarssl_conv_taps", _p(W), _p(fwd), _p(dgr), c_int(code), _stream())
    return fwd, dgr


def patch_w(W, dtype):
    """(d,4,F,1) f32 frame-patch conv weight -> [d][f*4+c] GEMM operand in `dtype`."""
    _need_cuda(W)
    d, _, F, _ = W.shape
    assert W.dtype == torch.float32 and W.is_contiguous() and W.shape[1] == 4 and W.shape[3] == 1
    out = torch.empty((d, F * 4), dtype=dtype, device=W.device)
    _lib.call("sarssl_patch_w", _p(W), _p(out), c_int(d), c_int(F), c_int(_DT[dtype]), _stream())
    return out


def patch_wgrad_accum(g, grad, nslice=1):
    """grad (d,4,F,1) f32 += sum over the nslice slices of g [nslice][d][f*4+c] f32 (split-K partials of gemm_tn_partials)."""
    d, _, F, _ = grad.shape
    assert g.is_contiguous() and grad.is_contiguous() and g.numel() == nslice * grad.numel()
    _lib.call("sarssl_patch_wgrad_accum", _p(g), c_int(nslice), _p(grad), c_int(d), c_int(F), _stream())


def gemm_tn_partials(dy, x, split):
    """Split-K partial products of dy[M,N]^T @ x[M,K] (bf16) -> (ws f32 [nslice][N][K], nslice): the caller folds them."""
    _need_cuda(dy, x)
    Mr, N = dy.shape
    K = x.shape[1]
    assert dy.dtype == torch.bfloat16 and x.dtype in _16 and split > 0
    per = ((Mr + split - 1) // split + 63) // 64 * 64
    nslice = (Mr + per - 1) // per
    ws = torch.empty((nslice, N, K), dtype=torch.float32, device=dy.device)
    _lib.call("sarssl_gemm", _p(dy), _p(x), c_void_p(0), c_int(BF16), c_int(dt(x)), c_int(F32), c_int(0), c_int(0), c_int(N), c_int(K),
              c_int(Mr), c_long(dy.stride(0)), c_long(x.stride(0)), c_long(K), c_int(1), c_int(1),
              c_long(0), c_long(0), c_long(0), c_long(0), c_long(0), c_long(0), c_float(1.0), c_float(1.0), c_void_p(0), c_int(0),
              c_void_p(0), c_long(0), c_long(0), c_long(0), c_float(1.0), c_void_p(0), c_void_p(0), c_int(0), c_int(F32), c_float(0.0),
              c_ulonglong(0), c_int(0), _p(ws), c_int(split), c_int(0), _stream())
    return ws, nslice


def f64_accum(src, dst, scale=1.0):
    _lib.call("sarssl_f64_accum", _p(src), _p(dst), c_int(src.numel()), c_float(scale), _stream())


def conv3x3_fwd(x, w_tap, scale=None, shift=None, precise=False, want_stats=False):
    """x (B,F,T,64); w_tap [9][64][64] ([tap][co][ci]) in x.dtype; optional BN+ReLU prologue.
    want_stats (bf16 only): also returns f64[128] = per-channel sum | sum of squares of the output (fused epilogue)."""
    _need_cuda(x, w_tap)
    B, F, T, C = x.shape
    assert C == 64 and w_tap.dtype == x.dtype and w_tap.is_contiguous() and x.is_contiguous()
    out = torch.empty_like(x)
    ws = _f32ws(x.numel(), x.device, "conv_acc") if (precise and x.dtype == torch.float32) else None
    sums = _sums(128, x.device) if want_stats else None
    # separate timing labels: the BN+ReLU-prologue launches (forward convolutions) and the identity launches (data gradients) are
    # different amounts of work per tile
    with _Timed("conv3x3_fwd:bn_prologue" if scale is not None else "conv3x3_fwd:identity"):
        _lib.call("sarssl_conv3x3_fwd", _p(x), _p(w_tap), _p(out), c_int(dt(x)), c_int(dt(w_tap)), c_int(B), c_int(F), c_int(T),
                  _p(scale), _p(shift), c_int(1 if ws is not None else 0), _p(ws), _p(sums), _stream())
    return (out, sums) if want_stats else out


def conv3x3_fwd_c1(a0, W1, scale, shift, w_tap, want_stats=False):
    """3x3 convolution of relu(scale * (W1 a0) + shift) straight from the stem's 4-channel input a0 (B,F,T,4) bf16 - the first layer's
    64-channel output is formed while staging, never stored.  -> out (B,F,T,64) bf16 (or (out, stats f64[128]))."""
    _need_cuda(a0, W1, scale, shift, w_tap)
    B, F, T, C = a0.shape
    assert C == 4 and a0.dtype in _16 and w_tap.dtype == a0.dtype and a0.is_contiguous() and W1.is_contiguous()
    out = torch.empty((B, F, T, 64), dtype=a0.dtype, device=a0.device)
    stats = _sums(128, a0.device) if want_stats else None
    with _Timed("conv3x3_fwd_c1"):
        _lib.call("sarssl_conv3x3_fwd_c1", _p(a0), _p(W1), _p(scale), _p(shift), _p(w_tap), _p(out), c_int(B), c_int(F), c_int(T), _p(stats),
                  c_int(dt(a0)), _stream())
    return (out, stats) if want_stats else out


_stamps = None              # (int64 device tensor, [labels]) while tools/step_stamps.py records timeline markers


def stamp(label):
    """Timeline marker (tools only): records the device clock on the current stream when markers are being collected, else nothing."""
    if _stamps is None:
        return
    buf, labels = _stamps
    i = len(labels)
    if i < buf.numel():
        labels.append(label)
        _lib.call("sarssl_stamp", c_void_p(buf.data_ptr() + 8 * i), _stream())


def conv_clock_probe(buf):
    """Clock-probe buffer (int64[20] device tensor, or None) of the 3x3 forward / data-gradient launches of this thread's context."""
    _lib.call("sarssl_ctx_set_clock_probe", c_void_p(_lib._make_current()), _p(buf))


def conv_cus_override(ncus):
    """Workgroup count of the 3x3 gradient launches issued under this thread's context (0: the library's default rule)."""
    _lib.call("sarssl_ctx_set_conv_cus", c_void_p(_lib._make_current()), c_int(int(ncus)))


def conv3x3_wgrad_c1(dy, a0, W1, scale, shift, acc_into):
    """Weight gradient of that convolution, added into the (64,64,3,3) f32 parameter-gradient buffer; the input operand is formed from a0
    while staging."""
    _need_cuda(dy, a0, W1)
    B, F, T, _ = a0.shape
    assert a0.dtype in _16 and dy.dtype == torch.bfloat16 and acc_into.shape == (64, 64, 3, 3) and acc_into.is_contiguous()
    nbytes = _lib.lib().sarssl_conv3x3_wgrad_workspace_bytes
    nbytes.restype = c_long
    part = workspace(nbytes(c_int(B), c_int(F), c_int(T)), a0.device, "wgrad_part")
    with _Timed("conv3x3_wgrad_kernel"):
        _lib.call("sarssl_conv3x3_wgrad_c1_acc", _p(dy), _p(a0), _p(W1), c_int(B), c_int(F), c_int(T), _p(scale), _p(shift), _p(acc_into),
                  _p(part), c_int(dt(a0)), _stream())
    return True


def conv3x3_dgrad_bnred(dy, w_tap_dgrad, y, aff):
    """dz = conv3x3(dy, flipped taps) and the BatchNorm-backward sums red f64[128] of the BN+ReLU in front (pre-BN activations
    ``y``, ``aff`` = (4,64) scale|shift|mean|rstd), accumulated in the convolution's epilogue (bf16)."""
    _need_cuda(dy, w_tap_dgrad, y, aff)
    B, F, T, C = dy.shape
    assert C == 64 and dy.dtype == torch.bfloat16 and w_tap_dgrad.dtype == torch.bfloat16 and y.dtype in _16 and y.shape == dy.shape
    assert aff.dtype == torch.float32 and aff.is_contiguous() and aff.numel() == 256 and dy.is_contiguous() and y.is_contiguous()
    dz = torch.empty_like(dy)
    red = _sums(128, dy.device)
    with _Timed("conv3x3_dgrad_bnred"):          # (its own label: this launch also reads y and reduces)
        _lib.call("sarssl_conv3x3_dgrad_bnred", _p(dy), _p(w_tap_dgrad), _p(dz), c_int(B), c_int(F), c_int(T), _p(y), _p(aff), _p(red),
                  c_int(dt(y)), _stream())
    return dz, red


def conv3x3_wgrad(dy, zin, scale=None, shift=None, precise=False, acc_into=None):
    """-> dW f32 [9][64][64] ([tap][co][ci]); or, with acc_into = the (64,64,3,3) f32 parameter-gradient buffer (bf16 operands), the
    weight gradient is added straight into it (nn.Conv2d layout) and None is returned."""
    _need_cuda(dy, zin)
    B, F, T, C = zin.shape
    nbytes = _lib.lib().sarssl_conv3x3_wgrad_workspace_bytes
    nbytes.restype = c_long
    part = workspace(nbytes(c_int(B), c_int(F), c_int(T)), zin.device, "wgrad_part")
    if acc_into is not None and zin.dtype in _16:
        assert acc_into.shape == (64, 64, 3, 3) and acc_into.dtype == torch.float32 and acc_into.is_contiguous() and dy.dtype == torch.bfloat16
        with _Timed("conv3x3_wgrad_kernel"):
            _lib.call("sarssl_conv3x3_wgrad_acc", _p(dy), _p(zin), c_int(B), c_int(F), c_int(T), _p(scale), _p(shift), _p(acc_into),
                      _p(part), c_int(dt(zin)), _stream())
        return None
    dW = torch.empty((9, 64, 64), dtype=torch.float32, device=zin.device)
    with _Timed("conv3x3_wgrad_kernel"):
        _lib.call("sarssl_conv3x3_wgrad", _p(dy), _p(zin), c_int(dt_ga(dy, zin)), c_int(B), c_int(F), c_int(T), _p(scale), _p(shift), _p(dW),
                  _p(part), c_int(1 if (precise and zin.dtype == torch.float32) else 0), _stream())
    return dW


# ------------------------------------------------------------------------------------------------
# channels-last BatchNorm pieces.  aff = (scale, shift, mean, rstd) f32 [C] each
def cl_stats(x, C):
    N = x.numel() // C
    sums = _sums(2 * C, x.device)
    _lib.call("sarssl_cl_stats", _p(x), c_long(N), c_int(C), _p(sums), c_int(dt(x)), _stream())
    return sums, N


def bn_train_affine(x, C, gamma, beta, running_mean, running_var, nbt, eps=1e-5, momentum=0.1, sums=None, N=None):
    """sums: optional precomputed f64[2C] (sum | sum of squares) from a fused producer epilogue (then x may be None with N given)."""
    if sums is None:
        sums, N = cl_stats(x, C)
    elif N is None:
        N = x.numel() // C
    aff = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    _lib.call("sarssl_bn_finalize", _p(sums), c_long(N), c_int(C), _p(gamma), _p(beta), c_float(eps), c_float(momentum),
              _p(running_mean), _p(running_var), _p(nbt), _p(aff[0]), _p(aff[1]), _p(aff[2]), _p(aff[3]), _stream())
    return aff


def bn_eval_affine(C, gamma, beta, running_mean, running_var, eps=1e-5):
    aff = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    _lib.call("sarssl_bn_eval_affine", c_int(C), _p(gamma), _p(beta), c_float(eps), _p(running_mean), _p(running_var),
              _p(aff[0]), _p(aff[1]), _p(aff[2]), _p(aff[3]), _stream())
    return aff


def cl_bn_train_act(x, C, gamma, beta, running_mean, running_var, nbt, sums, act, eps=1e-5, momentum=0.1):
    """bn_train_affine(sums=...) + cl_affine_act in one launch -> (aff (4, C), z)."""
    aff = torch.empty((4, C), dtype=torch.float32, device=x.device)
    z = torch.empty_like(x)
    _lib.call("sarssl_cl_bn_train_act", _p(x), c_long(x.numel() // C), c_int(C), _p(sums), _p(gamma), _p(beta), c_float(eps),
              c_float(momentum), _p(running_mean), _p(running_var), _p(nbt), _p(aff[0]), _p(aff[1]), _p(aff[2]), _p(aff[3]), c_int(act),
              _p(z), c_int(dt(x)), _stream())
    return aff, z


def cl_affine_act(x, C, aff, act):
    z = torch.empty_like(x)
    _lib.call("sarssl_cl_affine_act", _p(x), c_long(x.numel() // C), c_int(C), _p(aff[0]), _p(aff[1]), c_int(act), _p(z),
              c_int(dt(x)), _stream())
    return z


def cl_bn_bwd_reduce(dz, y, C, aff, act):
    red = _sums(2 * C, y.device)
    _lib.call("sarssl_cl_bn_bwd_reduce", _p(dz), _p(y), c_long(y.numel() // C), c_int(C), _p(aff[0]), _p(aff[1]), _p(aff[2]),
              _p(aff[3]), c_int(act), _p(red), c_int(dt_ga(dz, y)), _stream())
    return red


def cl_bn_bwd_apply(dz, y, C, aff, act, g_is_masked, use_stats, red, out=None, pgrads=None):
    """pgrads = (dgamma, dbeta) f32 [C] gradient buffers: dbeta += red[:C], dgamma += red[C:] in the same launch."""
    if out is None:
        out = torch.empty(y.shape, dtype=dz.dtype, device=y.device)
    dg, db = pgrads if pgrads is not None else (None, None)
    _lib.call("sarssl_cl_bn_bwd_apply_pg", _p(dz), _p(y), c_long(y.numel() // C), c_int(C), _p(aff[0]), _p(aff[1]), _p(aff[2]),
              _p(aff[3]), c_int(act), c_int(1 if g_is_masked else 0), c_int(1 if use_stats else 0), _p(red), _p(out), _p(dg), _p(db),
              c_int(dt_ga(dz, y)), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# row / elementwise kernels (csrc/elementwise.hip)
def layernorm_fwd(x2d, gamma, beta, eps=1e-5, out=None, save=True):
    M, d = x2d.shape
    if out is None:
        out = torch.empty((M, d), dtype=x2d.dtype, device=x2d.device)
    stats = torch.empty((2, M), dtype=torch.float32, device=x2d.device) if save else None
    _lib.call("sarssl_layernorm_fwd", _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(d), _p(gamma), _p(beta), c_float(eps),
              _p(out), c_long(out.stride(0)), _p(stats[0]) if save else c_void_p(0), _p(stats[1]) if save else c_void_p(0),
              c_int(dt(x2d)), _stream())
    return out, stats


def layernorm_fwd2(x2d, ga, ba, epsa, gb, bb, epsb, out=None):
    """y = LN_a(x), z = LN_b(y) in one launch -> (y, stats_a, z, stats_b); the results of two layernorm_fwd calls bit for bit."""
    M, d = x2d.shape
    if out is None:
        out = torch.empty((M, d), dtype=x2d.dtype, device=x2d.device)
    z = torch.empty((M, d), dtype=x2d.dtype, device=x2d.device)
    sa = torch.empty((2, M), dtype=torch.float32, device=x2d.device)
    sb = torch.empty((2, M), dtype=torch.float32, device=x2d.device)
    _lib.call("sarssl_layernorm_fwd2", _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(d), _p(ga), _p(ba), c_float(epsa), _p(out),
              c_long(out.stride(0)), _p(sa[0]), _p(sa[1]), _p(gb), _p(bb), c_float(epsb), _p(z), c_long(z.stride(0)), _p(sb[0]), _p(sb[1]),
              c_int(dt(x2d)), _stream())
    return out, sa, z, sb


_ln_batch = None              # list of (partials, nparts, d, dgamma, dbeta) while an ln_reduce_batched() context is open


class ln_reduce_batched:
    """Context manager: the dgamma / dbeta folds of the LayerNorm backward launches issued inside run as ONE launch at exit
    (sarssl_ln_param_reduce_multi) instead of one small reduce per LayerNorm."""

    def __enter__(self):
        global _ln_batch
        self._outer = _ln_batch
        if _ln_batch is None:
            _ln_batch = []
        return self

    def __exit__(self, *exc):
        global _ln_batch
        if self._outer is None:
            if exc[0] is None:
                ln_reduce_flush()
            _ln_batch = None
        return False


def ln_reduce_flush():
    global _ln_batch
    if not _ln_batch:
        return
    import ctypes
    items, _ln_batch = _ln_batch, []
    for i in range(0, len(items), 8):
        ch = items[i:i + 8]
        n = len(ch)
        _lib.call("sarssl_ln_param_reduce_multi", (ctypes.c_void_p * n)(*[c[0].data_ptr() for c in ch]),
                  (ctypes.c_int * n)(*[c[1] for c in ch]), (ctypes.c_int * n)(*[c[2] for c in ch]),
                  (ctypes.c_void_p * n)(*[c[3].data_ptr() for c in ch]), (ctypes.c_void_p * n)(*[c[4].data_ptr() for c in ch]),
                  c_int(n), _stream())


def layernorm_bwd(dy2d, x2d, gamma, stats, resid=None, dgamma=None, dbeta=None, out=None, drop=None):
    """drop = (p, seed, gscale): also returns out2 = out * dropout_mask(seed) * gscale (contiguous) - what ``act_bwd(out, None, 0, p, seed,
    gscale)`` would compute in its own pass; then the result is (out, out2)."""
    M, d = x2d.shape
    if out is None:
        out = torch.empty((M, d), dtype=dy2d.dtype, device=x2d.device)
    part = None
    if dgamma is not None:
        fn = _lib.lib().sarssl_layernorm_bwd_workspace_bytes
        fn.restype = c_long
        if _ln_batch is not None:              # partials only; folded with the block's other LayerNorms when the batch closes
            part = torch.empty((fn(c_long(M), c_int(d)) // 4,), dtype=torch.float32, device=x2d.device)
            _ln_batch.append((part, int(_lib.lib().sarssl_layernorm_bwd_nparts(c_long(M))), d, dgamma, dbeta))
            dgamma = dbeta = None
        else:
            part = workspace(fn(c_long(M), c_int(d)), x2d.device, "ln_part")
    if drop is not None:
        p, seed, gscale = drop
        out2 = torch.empty((M, d), dtype=dy2d.dtype, device=x2d.device)
        _lib.call("sarssl_layernorm_bwd_drop", _p(dy2d), c_long(dy2d.stride(0)), _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(d),
                  _p(gamma), _p(stats[0]), _p(stats[1]), _p(resid), c_long(resid.stride(0) if resid is not None else 0), _p(out),
                  c_long(out.stride(0)), _p(dgamma), _p(dbeta), _p(part), _p(out2), c_float(p), c_ulonglong(seed), c_float(gscale),
                  c_int(dt_ga(dy2d, x2d)), _stream())
        return out, out2
    _lib.call("sarssl_layernorm_bwd", _p(dy2d), c_long(dy2d.stride(0)), _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(d),
              _p(gamma), _p(stats[0]), _p(stats[1]), _p(resid), c_long(resid.stride(0) if resid is not None else 0), _p(out),
              c_long(out.stride(0)), _p(dgamma), _p(dbeta), _p(part), c_int(dt_ga(dy2d, x2d)), _stream())
    return out


def glu_fwd(h):
    M, d2 = h.shape
    g = torch.empty((M, d2 // 2), dtype=h.dtype, device=h.device)
    _lib.call("sarssl_glu_fwd", _p(h), c_long(M), c_int(d2 // 2), _p(g), c_int(dt(h)), _stream())
    return g


def glu_bwd(dg, h):
    dh = torch.empty(h.shape, dtype=dg.dtype, device=h.device)
    _lib.call("sarssl_glu_bwd", _p(dg), _p(h), c_long(h.shape[0]), c_int(h.shape[1] // 2), _p(dh), c_int(dt_ga(dg, h)), _stream())
    return dh


def dwconv(x3d, w, flip=False):
    """x (B,T,d), w f32 (d,31)."""
    B, T, d = x3d.shape
    y = torch.empty_like(x3d)
    _lib.call("sarssl_dwconv_fwd", _p(x3d), _p(w), c_int(B), c_int(T), c_int(d), c_int(w.shape[-1]), c_int(1 if flip else 0), _p(y),
              c_int(dt(x3d)), _stream())
    return y


def dwglu_fwd(h2d, w, B, T, want_stats=False):
    """h [B*T, 2d] -> c [B*T, d] = depthwise conv (k = 31) of GLU(h) over frames, (+ f64[2d] BatchNorm sums of c)."""
    _need_cuda(h2d, w)
    d = h2d.shape[1] // 2
    c = torch.empty((B * T, d), dtype=h2d.dtype, device=h2d.device)
    sums = _sums(2 * d, h2d.device) if want_stats else None
    _lib.call("sarssl_dwglu_fwd", _p(h2d), _p(w), c_int(B), c_int(T), c_int(d), c_int(w.shape[-1]), _p(c), _p(sums), c_int(dt(h2d)),
              _stream())
    return (c, sums) if want_stats else c


def dwglu_bwd(dc2d, h2d, w, B, T):
    """dc [B*T, d], h [B*T, 2d] -> dh [B*T, 2d] (depthwise-conv data gradient + GLU backward)."""
    d = dc2d.shape[1]
    dh = torch.empty(h2d.shape, dtype=dc2d.dtype, device=h2d.device)
    _lib.call("sarssl_dwglu_bwd", _p(dc2d), _p(h2d), _p(w), c_int(B), c_int(T), c_int(d), c_int(w.shape[-1]), _p(dh), c_int(dt_ga(dc2d, h2d)),
              _stream())
    return dh


def dwglu_wgrad(dc2d, h2d, dw_out, B, T):
    """dw_out f32 (d, 31) += depthwise-conv weight gradient, GLU(h) recomputed."""
    d = dc2d.shape[1]
    fn = _lib.lib().sarssl_dwglu_wgrad_workspace_bytes
    fn.restype = c_long
    nbytes = fn(c_int(B), c_int(T), c_int(d))
    n = dw_out.numel()
    if _splitk_batch is not None and dw_out.is_contiguous() and n % 4 == 0 and dw_out.data_ptr() % 16 == 0:
        # inside a block's backward: the per-tile-group partials join the block's ONE fold launch (split-K partials, bias column sums)
        part = torch.empty((nbytes // 4,), dtype=torch.float32, device=dc2d.device)
        _lib.call("sarssl_dwglu_wgrad", _p(dc2d), _p(h2d), c_int(B), c_int(T), c_int(d), c_int(dw_out.shape[-1]), c_void_p(0), _p(part),
                  c_int(dt_ga(dc2d, h2d)), _stream())
        _splitk_batch.append((part, nbytes // 4 // n, 1, n, dw_out.view(1, n), n))
        return
    part = workspace(nbytes, dc2d.device, "dwglu_part")
    _lib.call("sarssl_dwglu_wgrad", _p(dc2d), _p(h2d), c_int(B), c_int(T), c_int(d), c_int(dw_out.shape[-1]), _p(dw_out), _p(part),
              c_int(dt_ga(dc2d, h2d)), _stream())


def dwconv_wgrad(dy3d, x3d, dw_out):
    B, T, d = x3d.shape
    fn = _lib.lib().sarssl_dwconv_wgrad_workspace_bytes
    fn.restype = c_long
    part = workspace(fn(c_int(B), c_int(T), c_int(d)), x3d.device, "dw_part")
    _lib.call("sarssl_dwconv_wgrad", _p(dy3d), _p(x3d), c_int(B), c_int(T), c_int(d), c_int(dw_out.shape[-1]), _p(dw_out), _p(part),
              c_int(dt_ga(dy3d, x3d)), _stream())


def softmax_relshift_fwd(content, pos, scale, dtype, p_drop=0.0, seed=0):
    nmat = content.numel() // (content.shape[-1] ** 2)
    T = content.shape[-1]
    p = torch.empty(content.shape, dtype=dtype, device=content.device)
    pd = torch.empty_like(p) if p_drop > 0 else p
    _lib.call("sarssl_softmax_relshift_fwd", _p(content), _p(pos), c_long(nmat), c_int(T), c_float(scale), _p(p), _p(pd),
              c_float(p_drop), c_ulonglong(seed), c_int(_DT[dtype]), _stream())
    return p, pd


def softmax_bwd(dpd, p, scale, p_drop=0.0, seed=0):
    T = p.shape[-1]
    nmat = p.numel() // (T * T)
    ds = torch.empty(p.shape, dtype=gdtype_of(p.dtype), device=p.device)
    _lib.call("sarssl_softmax_bwd", _p(dpd), _p(p), c_long(nmat), c_int(T), c_float(scale), c_float(p_drop), c_ulonglong(seed),
              _p(ds), c_int(dt_ga(ds, p)), _stream())
    return ds


def relshift_bwd(dscore):
    T = dscore.shape[-1]
    dpos = torch.empty_like(dscore)
    _lib.call("sarssl_relshift_bwd", _p(dscore), c_long(dscore.numel() // (T * T)), c_int(T), _p(dpos), c_int(dt(dscore)), _stream())
    return dpos


def relpos_attn_supported(T, dh, dtype):
    """Shapes / dtypes the fused attention kernels take (csrc/attention.hip); otherwise the caller uses the GEMM + softmax path."""
    return dtype in _16 and bool(_lib.lib().sarssl_relpos_attn_supported(c_int(T), c_int(dh)))


def relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, p_drop=0.0, seed=0, need_bwd=True, want_ctx32=False):
    """qu [B*T, d], k / v [B*T, d] (row-strided views), bias (B,H,T,T) shifted positional score -> ctx [B*T, d] bf16,
    (ctx32 [B*T, d] f32 unrounded, lse (B,H,T)) for backward."""
    _need_cuda(qu, k, v, bias)
    assert k.stride(0) == v.stride(0) and bias.is_contiguous() and qu.dtype in _16 and k.dtype == qu.dtype and v.dtype == qu.dtype and \
        bias.dtype == qu.dtype
    ctx = torch.empty((B * T, H * dh), dtype=qu.dtype, device=qu.device)
    ctx32 = torch.empty((B * T, H * dh), dtype=torch.float32, device=qu.device) if (need_bwd or want_ctx32) else None
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qu.device)
    _lib.call("sarssl_relpos_attn_fwd", _p(qu), c_long(qu.stride(0)), _p(k), _p(v), c_long(k.stride(0)), _p(bias), _p(ctx),
              c_long(ctx.stride(0)), _p(ctx32), _p(lse), c_int(B), c_int(H), c_int(T), c_int(dh), c_float(scale), c_float(p_drop),
              c_ulonglong(seed), c_int(dt(qu)), _stream())
    return ctx, (ctx32, lse)


def relpos_attn_pos_supported(T, dh, dtype):
    """Shapes the forward with the in-kernel positional score takes (T <= 256)."""
    return dtype in _16 and bool(_lib.lib().sarssl_relpos_attn_pos_supported(c_int(T), c_int(dh)))


def relpos_attn_fwd_pos(qu, qv, k, v, pos, B, H, T, dh, scale, p_drop=0.0, seed=0, need_bwd=True, biases=None, want_ctx32=False, pair=False):
    """relpos_attn_fwd with the shifted positional score (qv pos^T, relative shift) formed inside the kernel.  Returns
    ctx, (ctx32, lse), bias ((B,H,T,T) as the kernel used it; None when need_bwd is False).
    biases = (u, v) f32 [H*dh]: qu and qv are then the same plain query projection q, the kernel adds the biases while loading.
    pair (fp16 tensors): the context as a Pair - the kernel also writes fp16(c - fp16(c)) of its unrounded f32 context."""
    _need_cuda(qu, qv, k, v, pos)
    ub, vb = biases if biases is not None else (None, None)
    assert biases is None or (qu.data_ptr() == qv.data_ptr() and ub.dtype == torch.float32 and vb.dtype == torch.float32 and
                              ub.numel() == H * dh and vb.numel() == H * dh)
    assert k.stride(0) == v.stride(0) and qu.stride(0) == qv.stride(0) and qu.dtype in _16 and \
        all(t.dtype == qu.dtype for t in (qv, k, v, pos)) and pos.stride(1) == 1
    ctx = torch.empty((B * T, H * dh), dtype=qu.dtype, device=qu.device)
    ctx32 = torch.empty((B * T, H * dh), dtype=torch.float32, device=qu.device) if (need_bwd or want_ctx32) else None
    bias = torch.empty((B, H, T, T), dtype=qu.dtype, device=qu.device) if need_bwd else None
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qu.device)
    if pair:
        assert qu.dtype == torch.float16
        lo = torch.empty_like(ctx)
        _lib.call("sarssl_relpos_attn_fwd_pos_pair", _p(qu), _p(qv), c_long(qu.stride(0)), _p(k), _p(v), c_long(k.stride(0)), _p(pos),
                  c_long(pos.stride(0)), _p(bias), _p(ctx), _p(lo), c_long(ctx.stride(0)), _p(ctx32), _p(lse), c_int(B), c_int(H), c_int(T),
                  c_int(dh), c_float(scale), c_float(p_drop), c_ulonglong(seed), _p(ub), _p(vb), _stream())
        return Pair(ctx, lo), (ctx32, lse), bias
    _lib.call("sarssl_relpos_attn_fwd_pos", _p(qu), _p(qv), c_long(qu.stride(0)), _p(k), _p(v), c_long(k.stride(0)), _p(pos),
              c_long(pos.stride(0)), _p(bias), _p(ctx), c_long(ctx.stride(0)), _p(ctx32), _p(lse), c_int(B), c_int(H), c_int(T), c_int(dh),
              c_float(scale), c_float(p_drop), c_ulonglong(seed), _p(ub), _p(vb), c_int(dt(qu)), _stream())
    return ctx, (ctx32, lse), bias


def relpos_attn_pos_long_supported(T, dh, dtype):
    """Shapes the long-sequence forward with the in-kernel positional score takes (T > 256, one 256-key block of the slab at a time)."""
    return dtype in _16 and bool(_lib.lib().sarssl_relpos_attn_pos_long_supported(c_int(T), c_int(dh)))


def relpos_attn_fwd_pos_long(qu, qv, k, v, pos, B, H, T, dh, scale, p_drop=0.0, seed=0, need_bwd=True, want_ctx32=False):
    """relpos_attn_fwd_pos for T > 256 (biased projections qu / qv; no u / v bias form).  Returns ctx, (ctx32, lse), bias - the (B,H,T,T)
    shifted score as relpos_attn_bwd reads it (None when need_bwd is False)."""
    _need_cuda(qu, qv, k, v, pos)
    assert k.stride(0) == v.stride(0) and qu.stride(0) == qv.stride(0) and qu.dtype in _16 and \
        all(t.dtype == qu.dtype for t in (qv, k, v, pos)) and pos.stride(1) == 1
    ctx = torch.empty((B * T, H * dh), dtype=qu.dtype, device=qu.device)
    ctx32 = torch.empty((B * T, H * dh), dtype=torch.float32, device=qu.device) if (need_bwd or want_ctx32) else None
    bias = torch.empty((B, H, T, T), dtype=qu.dtype, device=qu.device) if need_bwd else None
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qu.device)
    _lib.call("sarssl_relpos_attn_fwd_pos_long", _p(qu), _p(qv), c_long(qu.stride(0)), _p(k), _p(v), c_long(k.stride(0)), _p(pos),
              c_long(pos.stride(0)), _p(bias), _p(ctx), c_long(ctx.stride(0)), _p(ctx32), _p(lse), c_int(B), c_int(H), c_int(T), c_int(dh),
              c_float(scale), c_float(p_drop), c_ulonglong(seed), c_int(dt(qu)), _stream())
    return ctx, (ctx32, lse), bias


def relpos_attn_bwd(qu, k, v, bias, aux, dctx, dqu, dk, dv, B, H, T, dh, scale, p_drop=0.0, seed=0):
    """aux = (ctx32, lse) from relpos_attn_fwd.  Writes dqu / dk / dv (row-strided [B*T, d] views, dk and dv with the same row
    stride) and returns dbias (B,H,T,T)."""
    ctx32, lse = aux
    _need_cuda(qu, k, v, bias, ctx32, lse, dctx, dqu, dk, dv)
    assert k.stride(0) == v.stride(0) and dk.stride(0) == dv.stride(0) and k.dtype == qu.dtype and v.dtype == qu.dtype and \
        bias.dtype == qu.dtype and dqu.dtype == dctx.dtype and dk.dtype == dctx.dtype and dv.dtype == dctx.dtype
    dbias = torch.empty(bias.shape, dtype=dctx.dtype, device=bias.device)
    dsum = _f32ws(B * H * T, qu.device, "attn_dsum")
    _lib.call("sarssl_relpos_attn_bwd", _p(qu), c_long(qu.stride(0)), _p(k), _p(v), c_long(k.stride(0)), _p(bias), _p(ctx32),
              _p(lse), _p(dctx), c_long(dctx.stride(0)), _p(dqu), c_long(dqu.stride(0)), _p(dk), _p(dv),
              c_long(dk.stride(0)), _p(dbias), _p(dsum), c_int(B), c_int(H), c_int(T), c_int(dh), c_float(scale), c_float(p_drop),
              c_ulonglong(seed), c_int(dt_ga(dctx, qu)), _stream())
    return dbias


def relpos_attn_bwd_pos(qu, qv, k, v, pos, bias, aux, dctx, dqu, dqv, dk, dv, B, H, T, dh, scale, p_drop=0.0, seed=0, biases=None,
                        dq_sum=None):
    """Backward of relpos_attn_fwd_pos with the positional-score gradients formed in the dQ kernel.  Writes dqu / dqv / dk / dv and
    returns dpos_part (B * ntile, T, d): partial gradients of the positional projection (sum over axis 0)."""
    ctx32, lse = aux
    _need_cuda(qu, qv, k, v, pos, bias, ctx32, lse, dctx, dqu, dqv, dk, dv)
    assert k.stride(0) == v.stride(0) and dk.stride(0) == dv.stride(0) and qu.stride(0) == qv.stride(0) and pos.stride(1) == 1 and \
        all(t.dtype == qu.dtype for t in (qv, k, v, pos, bias)) and all(t.dtype == dctx.dtype for t in (dqu, dqv, dk, dv)) and \
        bias.is_contiguous()
    ub, vb = biases if biases is not None else (None, None)
    assert biases is None or qu.data_ptr() == qv.data_ptr()
    ntile = (T + 127) // 128
    dpos_part = torch.empty((B * ntile, T, H * dh), dtype=dctx.dtype, device=qu.device)
    dsum = _f32ws(B * H * T, qu.device, "attn_dsum")
    fix = _f32ws(B * H * ntile * 2 * dh, qu.device, "attn_dqv_fix")
    _lib.call("sarssl_relpos_attn_bwd_pos", _p(qu), _p(qv), c_long(qu.stride(0)), _p(k), _p(v), c_long(k.stride(0)), _p(pos),
              c_long(pos.stride(0)), _p(bias), _p(ctx32), _p(lse), _p(dctx), c_long(dctx.stride(0)), _p(dqu), c_long(dqu.stride(0)),
              _p(dqv), c_long(dqv.stride(0)), _p(dk), _p(dv), c_long(dk.stride(0)), _p(dpos_part), _p(fix), _p(dsum), c_int(B), c_int(H),
              c_int(T), c_int(dh), c_float(scale), c_float(p_drop), c_ulonglong(seed), _p(ub), _p(vb), _p(dq_sum),
              c_long(dq_sum.stride(0) if dq_sum is not None else 0), c_int(dt_ga(dctx, qu)), _stream())
    return dpos_part


def bias2(q2d, u, v):
    M, d = q2d.shape
    qu = torch.empty((M, d), dtype=q2d.dtype, device=q2d.device)
    qv = torch.empty((M, d), dtype=q2d.dtype, device=q2d.device)
    _lib.call("sarssl_bias2", _p(q2d), c_long(q2d.stride(0)), c_long(M), c_int(d), _p(u), _p(v), _p(qu), _p(qv), c_int(dt(q2d)), _stream())
    return qu, qv


def axpby(x, y, a=1.0, b=1.0, out=None):
    if out is None:
        out = torch.empty_like(x)
    _lib.call("sarssl_axpby", _p(x), _p(y), c_float(a), c_float(b), c_long(x.numel()), _p(out), c_int(dt(x)), _stream())
    return out


def axpby2d(x2d, y2d, a=1.0, b=1.0, out=None):
    """out = a*x + b*y on row-strided [M, N] views (out may alias x)."""
    M, N = x2d.shape
    if out is None:
        out = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    _lib.call("sarssl_axpby2d", _p(x2d), c_long(x2d.stride(0)), _p(y2d), c_long(y2d.stride(0)), c_float(a), c_float(b), c_long(M),
              c_int(N), _p(out), c_long(out.stride(0)), c_int(dt(x2d)), _stream())
    return out


_colsum_batch = None          # list of (x2d, out) while a batch is open


def colsum(x2d, out_f32, now=False):
    """out_f32[n] += sum_m x2d[m][n].  Inside ``colsum_batched()`` the request is queued and executed with the others in one
    launch when the batch closes: x2d must stay unmodified and out_f32 unread until then; ``now=True`` opts out.
    Deterministic: per-row-slice partial sums are written and folded in slice order (no floating-point atomics)."""
    if _colsum_batch is not None and not now and x2d.dtype == torch.bfloat16:
        _colsum_batch.append((x2d, out_f32))
        if len(_colsum_batch) == 24:
            colsum_flush()
        return
    _colsum_flush_items([(x2d, out_f32)], 1, fold_now=True)


_colsum_ctx = None


def colsum_store(x2d, out=None):
    """out[n] = sum_m x2d[m][n] in x2d's dtype (one launch, nothing to zero)."""
    M, N = x2d.shape
    if out is None:
        out = torch.empty((N,), dtype=x2d.dtype, device=x2d.device)
    _lib.call("sarssl_colsum_store", _p(x2d), c_long(x2d.stride(0)), c_long(M), c_int(N), _p(out), c_int(dt(x2d)), _stream())
    return out


def colsum_flush():
    """Runs the queued column sums (one launch) on the current stream (or under the batch's flush context)."""
    global _colsum_batch
    if not _colsum_batch:
        return
    items, n = _colsum_batch, len(_colsum_batch)
    _colsum_batch = []
    if _colsum_ctx is not None:
        with _colsum_ctx(*[x for x, _ in items]):
            _colsum_flush_items(items, n)
    else:
        _colsum_flush_items(items, n)


def _colsum_flush_items(items, n, fold_now=False):
    """One launch writes every problem's per-row-slice column sums; the fold into the gradient buffers joins the open split-K batch
    (one reduce launch per backward stage for weight AND bias gradients) or, outside a batch / fold_now, runs right away."""
    import ctypes
    _need_cuda(*[t for it in items for t in it])
    nsl = [_lib.lib().sarssl_colsum_slices(c_long(x.shape[0]), c_int(x.shape[1])) for x, _ in items]
    offs, tot = [], 0
    for (x, _), g in zip(items, nsl):
        offs.append(tot)
        tot += g * x.shape[1]
    part = torch.empty((tot,), dtype=torch.float32, device=items[0][0].device)
    xs = (ctypes.c_void_p * n)(*[x.data_ptr() for x, _ in items])
    parts = (ctypes.c_void_p * n)(*[part.data_ptr() + 4 * o for o in offs])
    lds = (ctypes.c_long * n)(*[x.stride(0) for x, _ in items])
    Ms = (ctypes.c_long * n)(*[x.shape[0] for x, _ in items])
    Ns = (ctypes.c_int * n)(*[x.shape[1] for x, _ in items])
    _lib.call("sarssl_colsum_multi_partials", xs, lds, Ms, Ns, parts, c_int(n), c_int(dt(items[0][0])), _stream())
    folds = [(part[o:o + g * x.shape[1]], g, 1, x.shape[1], out, x.shape[1]) for (x, out), g, o in zip(items, nsl, offs)]
    if _splitk_batch is not None and not fold_now:
        for f in folds:
            _splitk_batch.append(f)
            if len(_splitk_batch) == 32:
                splitk_flush()
    else:
        for i in range(0, len(folds), 32):
            _splitk_flush_items(folds[i:i + 32], len(folds[i:i + 32]))


class colsum_batched:
    """Context manager: queue the column sums issued inside and run them in one launch at exit (per backward stage).
    ``flush_ctx``: context manager factory the launch is issued under (a side stream)."""

    def __init__(self, flush_ctx=None):
        self._ctx = flush_ctx

    def __enter__(self):
        global _colsum_batch, _colsum_ctx
        self._outer = _colsum_batch
        if _colsum_batch is None:
            _colsum_batch = []
            _colsum_ctx = self._ctx
        return self

    def __exit__(self, *exc):
        global _colsum_batch, _colsum_ctx
        if self._outer is None:
            if exc[0] is None:
                colsum_flush()
            _colsum_batch = None
            _colsum_ctx = None
        return False


def act_bwd(dz, h, act, p_drop=0.0, seed=0, gscale=1.0, out=None):
    if out is None:
        out = torch.empty_like(dz)
    _lib.call("sarssl_act_bwd", _p(dz), _p(h), c_long(dz.numel()), c_int(act), c_float(p_drop), c_ulonglong(seed), c_float(gscale),
              _p(out), c_int(dt_ga(dz, h)), _stream())
    return out


def cast(src, dtype, out=None):
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    _lib.call("sarssl_cast", _p(src), c_int(dt(src)), _p(out), c_int(_DT[dtype]), c_long(src.numel()), _stream())
    return out


def masked_mse_fwd(pred, x, idx_i32, mch_i32, sink=None, with_grad=False):
    """pred (B,T,F*4) -> f32[2] device tensor (loss, diff).  sink = (out_keep f32[2], acc f64[2]): the finalize launch also copies the
    two values into out_keep and adds them to acc (graph.py's running sums).  with_grad: -> (out, dpred), the loss gradient for an
    incoming gradient of 1 from the same pass (what masked_mse_bwd would compute in a pass of its own)."""
    B, _, F, T, _ = x.shape
    nm = idx_i32.shape[1]
    sums = _sums(128, x.device)
    out = torch.empty((2,), dtype=torch.float32, device=x.device)
    if with_grad:
        dpred = torch.empty(pred.shape, dtype=gdtype_of(pred.dtype), device=pred.device)
        keep, acc = sink if sink is not None else (None, None)
        _lib.call("sarssl_masked_mse_fwd_bwd", _p(pred), _p(x), _p(idx_i32), _p(mch_i32), c_int(B), c_int(F), c_int(T), c_int(nm),
                  _p(sums), _p(out), _p(keep), _p(acc), _p(dpred), c_int(dt_ga(dpred, pred)), _stream())
        return out, dpred
    if sink is not None:
        keep, acc = sink
        assert keep.dtype == torch.float32 and acc.dtype == torch.float64 and keep.numel() == 2 and acc.numel() == 2
        _lib.call("sarssl_masked_mse_fwd_acc", _p(pred), _p(x), _p(idx_i32), _p(mch_i32), c_int(B), c_int(F), c_int(T), c_int(nm),
                  _p(sums), _p(out), _p(keep), _p(acc), c_int(dt(pred)), _stream())
        return out
    _lib.call("sarssl_masked_mse_fwd", _p(pred), _p(x), _p(idx_i32), _p(mch_i32), c_int(B), c_int(F), c_int(T), c_int(nm), _p(sums),
              _p(out), c_int(dt(pred)), _stream())
    return out


# ---- decoder on the masked frames only (include/sarssl_hip.h): gather / scatter of the masked rows, loss on the compact prediction
def gather_rows(src2d, idx_i32, B, T):
    """src [B*T, d] (row stride any) -> [B*nm, d]: rows (b, idx[b][j]); idx ascending per item."""
    nm, d = idx_i32.shape[1], src2d.shape[1]
    dst = torch.empty((B * nm, d), dtype=src2d.dtype, device=src2d.device)
    _lib.call("sarssl_gather_rows", _p(src2d), c_long(src2d.stride(0)), _p(idx_i32), c_int(B), c_int(T), c_int(nm), c_int(d), _p(dst),
              c_int(dt(src2d)), _stream())
    return dst


def scatter_rows(src2d, idx_i32, B, T):
    """[B*nm, d] -> [B*T, d] with the rows at (b, idx[b][j]) and zeros elsewhere."""
    nm, d = idx_i32.shape[1], src2d.shape[1]
    dst = torch.empty((B * T, d), dtype=src2d.dtype, device=src2d.device)
    _lib.call("sarssl_scatter_rows", _p(src2d), _p(idx_i32), c_int(B), c_int(T), c_int(nm), c_int(d), _p(dst), c_long(d), c_int(dt(src2d)), _stream())
    return dst


def scatter_rows_drop16(src2d, idx_i32, B, T, p_drop, seed, gscale=1.0):
    """scatter_rows of an f32 [B*nm, d] gradient -> (dst f32 [B*T, d], dst16 bf16 [B*T, d]): dst16 = the bf16 copy with the dropout backward
    (p_drop, seed, gscale; mask of the FULL tensor's element indices) applied - bit-identical to act_bwd(cast(scatter_rows(src)))."""
    assert src2d.dtype == torch.float32 and src2d.is_contiguous()
    nm, d = idx_i32.shape[1], src2d.shape[1]
    dst = torch.empty((B * T, d), dtype=torch.float32, device=src2d.device)
    dst16 = torch.empty((B * T, d), dtype=torch.bfloat16, device=src2d.device)
    _lib.call("sarssl_scatter_rows_drop16", _p(src2d), _p(idx_i32), c_int(B), c_int(T), c_int(nm), c_int(d), _p(dst), _p(dst16), c_float(p_drop),
              c_ulonglong(seed), c_float(gscale), _stream())
    return dst, dst16


def masked_mse_compact(pred_c, x, idx_i32, mch_i32, sink=None, with_grad=False):
    """masked_mse_fwd on the compact prediction pred_c (B*nm, F*4) (rows in ascending frame order) -> out f32[2] or (out, dpred_c)."""
    B, _, F, T, _ = x.shape
    nm = idx_i32.shape[1]
    sums = _sums(128, x.device)
    out = torch.empty((2,), dtype=torch.float32, device=x.device)
    keep, acc = sink if sink is not None else (None, None)
    dpred = torch.empty(pred_c.shape, dtype=gdtype_of(pred_c.dtype), device=pred_c.device) if with_grad else None
    _lib.call("sarssl_masked_mse_compact", _p(pred_c), _p(x), _p(idx_i32), _p(mch_i32), c_int(B), c_int(F), c_int(T), c_int(nm), _p(sums), _p(out),
              _p(keep), _p(acc), _p(dpred), c_int(dt_ga(dpred, pred_c) if with_grad else dt(pred_c)), _stream())
    return (out, dpred) if with_grad else out


def masked_mse_bwd_compact(pred_c, x, idx_i32, mch_i32, gscale=1.0, gscale_dev=None):
    B, _, F, T, _ = x.shape
    nm = idx_i32.shape[1]
    dpred = torch.empty(pred_c.shape, dtype=gdtype_of(pred_c.dtype), device=pred_c.device)
    _lib.call("sarssl_masked_mse_bwd_compact", _p(pred_c), _p(x), _p(idx_i32), _p(mch_i32), c_int(B), c_int(F), c_int(T), c_int(nm), c_float(gscale),
              _p(gscale_dev), _p(dpred), c_int(dt_ga(dpred, pred_c)), _stream())
    return dpred


def masked_mse_bwd(pred, x, mp_u8, mch_i32, nm, gscale=1.0, gscale_dev=None):
    """gscale_dev: optional f32 device scalar (the incoming d(loss)); multiplied in-kernel, no host sync."""
    B, _, F, T, _ = x.shape
    dpred = torch.empty(pred.shape, dtype=gdtype_of(pred.dtype), device=pred.device)
    _lib.call("sarssl_masked_mse_bwd", _p(pred), _p(x), _p(mp_u8), _p(mch_i32), c_int(B), c_int(F), c_int(T), c_int(nm),
              c_float(gscale), _p(gscale_dev), _p(dpred), c_int(dt_ga(dpred, pred)), _stream())
    return dpred


def adam_step(p, g, m, v, p16, lr, step, gscale=1.0, betas=(0.9, 0.999), eps=1e-8, ph16=None, guard=None, nskipped=None, pl16=None):
    """p16 / ph16: bf16 / fp16 shadow copies of the parameters rewritten by the same pass (either may be None).  guard: f32 device
    tensor holding the step's loss - not finite -> the update is skipped on the device (nskipped: int32 device counter, += 1)."""
    assert (p16 is None or p16.dtype == torch.bfloat16) and (ph16 is None or ph16.dtype == torch.float16)
    assert guard is None or (guard.dtype == torch.float32 and guard.is_cuda)
    if pl16 is not None:              # hybrid mode: the lo shadow (fp16(p - fp16(p))) in the same pass
        assert pl16.dtype == torch.float16
        _lib.call("sarssl_adam_step_guard_lo", _p(p), _p(g), _p(m), _p(v), _p(p16), _p(ph16), _p(pl16), c_long(p.numel()), c_float(gscale), c_float(lr),
                  c_float(betas[0]), c_float(betas[1]), c_float(eps), c_int(step), _p(guard), _p(nskipped), _stream())
        return
    _lib.call("sarssl_adam_step_guard", _p(p), _p(g), _p(m), _p(v), _p(p16), _p(ph16), c_long(p.numel()), c_float(gscale), c_float(lr),
              c_float(betas[0]), c_float(betas[1]), c_float(eps), c_int(step), _p(guard), _p(nskipped), _stream())


# ---- device-resident step state (graph replay: dropout salt, Adam step count / bias corrections; csrc/api.hip) --------------
def step_state_new(device, salt, lr, betas=(0.9, 0.999)):
    nbytes = _lib.lib().sarssl_step_state_bytes
    nbytes.restype = c_long
    st = torch.zeros((int(nbytes()) + 7) // 8 * 8, dtype=torch.uint8, device=device)
    _lib.call("sarssl_step_state_init", _p(st), c_ulonglong(salt & 0xFFFFFFFFFFFFFFFF), c_float(lr), c_float(betas[0]), c_float(betas[1]),
              _stream())
    return st


def step_state_reset(st, lr, betas=(0.9, 0.999)):
    """Adam step count back to 0 (a freshly constructed optimizer), new learning rate; the dropout salt keeps running."""
    _lib.call("sarssl_step_state_reset", _p(st), c_float(lr), c_float(betas[0]), c_float(betas[1]), _stream())


def step_state_attach(st):
    """While attached (st not None) every launch that draws dropout masks adds the state's salt to its seed: attach only around
    graph capture - the pointer is baked into the captured launches, eager launches afterwards run unsalted again."""
    _lib.call("sarssl_ctx_attach_step_state", c_void_p(_lib._make_current()), _p(st))


# ---- collectives behind the C ABI (csrc/comm.hip): RCCL resolved with dlopen, one communicator per (process, device)
def comm_available():
    return bool(_lib.lib().sarssl_comm_available())


def comm_rccl_version():
    return int(_lib.lib().sarssl_comm_rccl_version())


def comm_unique_id():
    """128-byte id for comm_create (rank 0 generates it and hands it to the other ranks)."""
    import ctypes
    buf = ctypes.create_string_buffer(128)
    _lib.call("sarssl_comm_unique_id", buf)
    return buf.raw


def comm_create(nranks, rank, id128):
    """Communicator of `rank` among `nranks` on torch's current device (collective: every rank calls it with the same id)."""
    import ctypes
    assert len(id128) == 128
    L = _lib.lib()
    L.sarssl_comm_create.restype = c_void_p
    c = L.sarssl_comm_create(c_int(nranks), c_int(rank), ctypes.create_string_buffer(bytes(id128), 128))
    if not c:
        raise _lib.SarsslHipError("sarssl_comm_create: %s" % L.sarssl_last_error().decode())
    return c


def comm_destroy(comm):
    _lib.call("sarssl_comm_destroy", c_void_p(comm))


def comm_size(comm):
    return int(_lib.lib().sarssl_comm_size(c_void_p(comm)))


def allreduce_bucket(comm, bucket, stream=None):
    """In-place sum over the communicator's ranks of a contiguous f32 tensor, enqueued on `stream` (default: torch's current)."""
    _need_cuda(bucket)
    assert bucket.dtype == torch.float32 and bucket.is_contiguous()
    st = c_void_p(stream.cuda_stream) if stream is not None else _stream()
    _lib.call("sarssl_allreduce_bucket", c_void_p(comm), _p(bucket), c_long(bucket.numel()), st)


def step_tick(st):
    _lib.call("sarssl_step_tick", _p(st), _stream())


def adam_step_dev(p, g, m, v, p16, st, gscale=1.0, eps=1e-8, zero_grad=False, ph16=None, guard=None, pl16=None):
    assert (p16 is None or p16.dtype == torch.bfloat16) and (ph16 is None or ph16.dtype == torch.float16)
    assert guard is None or (guard.dtype == torch.float32 and guard.is_cuda)
    if pl16 is not None:
        assert pl16.dtype == torch.float16
        _lib.call("sarssl_adam_step_dev_guard_lo", _p(p), _p(g), _p(m), _p(v), _p(p16), _p(ph16), _p(pl16), c_long(p.numel()), c_float(gscale), _p(st),
                  c_float(eps), c_int(1 if zero_grad else 0), _p(guard), _stream())
        return
    _lib.call("sarssl_adam_step_dev_guard", _p(p), _p(g), _p(m), _p(v), _p(p16), _p(ph16), c_long(p.numel()), c_float(gscale), _p(st), c_float(eps),
              c_int(1 if zero_grad else 0), _p(guard), _stream())


def fp16_overflow(clear=True, device=None):
    """True when a kernel of this process's context has met a value outside fp16's range while encoding the network input since the last
    loss launch / the last call (include/sarssl_hip.h: sarssl_ctx_fp16_overflow).  Synchronises the current stream."""
    r = _lib.lib().sarssl_ctx_fp16_overflow(c_void_p(_lib.ctx(device)), c_int(1 if clear else 0), _stream())
    if r < 0:
        raise _lib.SarsslHipError("sarssl_ctx_fp16_overflow failed")
    return bool(r)


def step_state_skipped(st):
    """Optimizer steps the guarded Adam launch skipped since the state was created (non-finite loss).  Synchronises the current stream."""
    n = _lib.lib().sarssl_step_state_skipped(_p(st), _stream())
    if n < 0:
        raise _lib.SarsslHipError("sarssl_step_state_skipped failed")
    return int(n)
