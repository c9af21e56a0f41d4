"""Forward/backward orchestration of the MC-Conformer pretraining path on the HIP kernels.

Every function here takes activations as 2-D ``[M = B*T, d]`` (or channels-last 4-D) tensors in the runtime
dtype, the owning ``nn.Module`` (for its parameters, which keep the reference's state_dict layout) and a list to
push saved tensors on; the ``*_bwd`` twin consumes what was saved, accumulates parameter gradients straight into
``p.grad`` (views of the flat gradient buffer when the model is flattened) and returns the input gradient.
Reference semantics are cited per function; the math is restated in oracle/sarssl_oracle.py.
"""
import math

import os

import torch

from . import hip
from .runtime import RT, wt, wt_lo, wtg, gbuf, weights_version

_16 = (torch.bfloat16, torch.float16)          # 16-bit activation storage: the fused kernels' domain

RELU, SWISH = 1, 2


# ------------------------------------------------------------------------------------------------ GEMM helpers
def mm_nt(x, W, fp8=False, **kw):
    """x [M,K] @ W[N,K]^T -> [M,N]   (nn.Linear forward).  (``fp8``: accepted and ignored - the whole-step fp8 mode of rounds 2-5 was removed
    in round 6: its just-in-time quantisation made the step 7-12 % slower than bf16; the e4m3 GEMM kernels remain library entry points,
    hip.gemm_fp8 / tests/test_gpu_fp8.py.)"""
    M, K = x.shape
    return hip.gemm(x, W, M=M, N=W.shape[0], K=K, lda=x.stride(0), ldb=W.stride(0), precise=RT.precise, **kw)


def mm_nn(dy, W, fp8=False, **kw):
    """dy [M,N] @ W[N,K] -> [M,K]   (input gradient of nn.Linear)."""
    M, N = dy.shape
    return hip.gemm(dy, W, a_kc=True, b_kc=False, M=M, N=W.shape[1], K=N, lda=dy.stride(0), ldb=W.stride(0),
                    precise=RT.precise, **kw)


# Grouped launch: the weight-gradient products of one block (bf16) are collected while its backward runs and issued as ONE launch when
# it ends (hip.gemm_group_tn); SARSSL_WGRAD_GROUP=0 issues them one by one where they occur (A/B runs).  (Round 2 also measured
# companion-stream schedules for these products - per product and per block; neither beat keeping them on the chain, and a stream
# forked off a forked stream crashes hipStreamEndCapture on ROCm 7.2 (tools/capture_nested_fork_repro.py).  Removed in round 3.)
_WGRAD_GROUP = os.environ.get("SARSSL_WGRAD_GROUP", "1") != "0"
_WGRAD_CSUM = os.environ.get("SARSSL_WGRAD_CSUM", "1") != "0"   # bias gradients from the grouped weight-gradient launch (0: separate column-sum launch)
_STEM_LAST_ALL_CUS = os.environ.get("SARSSL_STEM_LAST_ALL_CUS", "1") != "0"   # gradient convolutions of the stem that runs last (spat) on every CU (model.py)
_WGRAD_SPLIT_BIG = int(os.environ.get("SARSSL_WGRAD_SPLIT_BIG", "4"))            # K-slices of a grouped launch with >= .._TILES output tiles
_WGRAD_SPLIT_BIG_TILES = int(os.environ.get("SARSSL_WGRAD_SPLIT_BIG_TILES", "128"))
_wg_blocks = []              # stack of pending-product lists (wgrad_block)


class wgrad_block:
    """Backward of one block: weight-gradient products issued inside are collected and enqueued at exit as one grouped launch,
    followed by the block's bias-gradient column sums (one launch of per-slice partials) and ONE fold launch for both.  Their
    operands must stay unmodified until then (true for every product in this file: operands are saved activations or fresh
    gradients)."""

    def __enter__(self):
        _wg_blocks.append([])
        return self

    def __exit__(self, *exc):
        items = _wg_blocks.pop()
        if exc[0] is not None:
            return False
        # products whose contraction length is no multiple of a K-tile (the positional projection's T rows when T % 64 != 0: config 5)
        # cannot join the grouped launch (it has no ragged instantiation) - they must not keep the block's other products out of it
        ragged = [it for it in items if it[0].shape[0] % 64 != 0]
        items = [it for it in items if it[0].shape[0] % 64 == 0]
        if _WGRAD_SPLIT_BIG != 8:
            # a group whose products already have plenty of 256 x 128 output tiles needs fewer K-slices to fill the chip: half the
            # partial-sum traffic of the products and of the fold launch
            tiles = sum(((it[0].shape[1] + 255) // 256) * ((it[1].shape[1] + 127) // 128) for it in items)
            if tiles >= _WGRAD_SPLIT_BIG_TILES:
                items = [(dy, x, g2, _WGRAD_SPLIT_BIG if (split == 8 and dy.shape[0] >= 4096) else split, bias) for dy, x, g2, split, bias in items]
        for dy, x, g2, split, bias in ragged:
            _wgrad_gemm(dy, x, g2, split)
            if bias is not None:
                hip.colsum(dy, bias)
        for i in range(0, len(items), 12):                 # one grouped launch per <= 12 products (csrc/gemm.hip)
            chunk = items[i:i + 12]
            if not (_WGRAD_GROUP and len(chunk) > 1 and hip.gemm_group_tn(chunk)):
                for dy, x, g2, split, bias in chunk:
                    _wgrad_gemm(dy, x, g2, split)
                    if bias is not None:
                        hip.colsum(dy, bias)
        hip.colsum_flush()                                 # partial sums join the split-K batch ...
        hip.splitk_flush()                                 # ... and are folded with the products' partials in one launch
        return False


def _wgrad_gemm(dy, x, g2, split):
    M, N = dy.shape
    K = x.shape[1]
    hip.gemm(dy, x, a_kc=False, b_kc=False, M=N, N=K, K=M, lda=dy.stride(0), ldb=x.stride(0), out=g2, ldc=K, precise=RT.precise,
             split_k=split)


def _wgrad_split(M, N, K, grouped):
    """Number of K-slices (M = rows of dy / x = the contraction length) of the weight-gradient product dy[M,N]^T x[M,K].  Multiples
    of 8 select the slice-major workgroup order of csrc/gemm.hip (each XCD's L2 reads one eighth of both operands once instead of
    the whole second operand).  Inside a grouped launch the other products of the block fill the chip, so 8 slices of >= 8 K-tiles
    are enough and keep the partial-sum traffic (2 x slices x N x K x 4 bytes) small; alone, ~2 workgroups per CU."""
    if grouped and M >= 4096:
        return 8
    tiles = ((N + 255) // 256) * ((K + 127) // 128)                          # 256 x 128 output tiles (csrc/gemm.hip)
    split = max(1, min((M + 511) // 512, (512 + tiles - 1) // tiles))       # ~2 workgroups per CU, >= 8 K-tiles each
    split = max(1, min(split, (1 << 23) // (N * K)))                         # partial-sum workspace <= 32 MB (reduce pass cost)
    return split // 8 * 8 if split >= 8 else split


def mm_tn_acc(dy, x, gW, group=True, bias=None):
    """gW[N,K] += dy[M,N]^T @ x[M,K]   (weight gradient of nn.Linear, f32 accumulate into the grad buffer) and, with ``bias``,
    bias[N] += column sums of dy (the layer's bias gradient): inside a grouped launch they come out of the same kernel (one extra MFMA
    per dy fragment against a ones fragment) instead of a second pass over dy."""
    M, N = dy.shape
    K = x.shape[1]
    g2 = gW.view(N, K)
    grouped = bool(group and _wg_blocks and RT.replay is None and _WGRAD_GROUP and RT.dtype in _16)
    split = _wgrad_split(M, N, K, grouped)
    if grouped:
        if bias is not None and not _WGRAD_CSUM:             # A/B: bias gradient by the stand-alone column-sum launch
            hip.colsum(dy, bias)
            bias = None
        _wg_blocks[-1].append((dy, x, g2, split, bias))      # enqueued when the block's backward ends (wgrad_block)
    else:
        _wgrad_gemm(dy, x, g2, split)
        if bias is not None:
            hip.colsum(dy, bias)


# ------------------------------------------------------------------------------------------------ hybrid mode (runtime.set_precision)
# fp16 stem and module-internal tensors, f32 residual stream: an f32 activation is handed to a Linear layer as an fp16 pair (hip.Pair,
# written by the LayerNorm in front of it), a weight as its (hi, lo) fp16 shadows; hip.gemm_split contracts hi hi + lo hi + hi lo.
_F32 = torch.float32
_H_STEM4 = os.environ.get("SARSSL_HYBRID_STEM4", "1") != "0"    # the stem's 4-channel tensors (64 -> 4 result, BatchNorm(4) + ReLU of it) as pairs (0: fp16)
_H_CTX = os.environ.get("SARSSL_HYBRID_CTX", "1") != "0"        # the attention context enters the output projection as a pair (0: fp16)
# Which Linear layers fed by a LayerNorm contract the ACTIVATION as a pair (three products) instead of its fp16 rounding (two).  The CPU
# study (profiles/r06_operand_rounding_study.txt) says the pair buys no per-bin accuracy where the layer's OUTPUT is an fp16 tensor (q / k /
# v, the feed-forward hidden layer: rms 1.02e-4 -> 1.06e-4 eval, 2.02e-4 -> 2.05e-4 train) - measured on the GPU (profiles/
# r06_hybrid_alo_ab.txt) the per-bin figures hold (F13 train max 5.6e-4 -> 7.3e-4 on that build, inside the 1e-3 gate) but the GRADIENT deviation of the
# f32-stream parameters grows 2.1e-3 -> 5.7e-3 / 6.9e-3 (the positional projection's weight, the convolution module's LayerNorm bias) for
# 0.05 ms per family of an 11.1 ms step.  Not worth it: every family keeps the pair by default; "pw1,dec1" is the 0.1 ms faster setting.
_H_ALO = set(v for v in os.environ.get("SARSSL_HYBRID_ALO", "ffn1,qkv,pw1,dec1").split(",") if v != "none" and v)
_H_PREP = os.environ.get("SARSSL_HYBRID_PREP", "1") != "0"      # 0: the hybrid mode's positional projections / patch pair formed at their point of use
_H_FFN2_FWD = os.environ.get("SARSSL_HYBRID_FFN2_FWD", "1") != "0"   # the feed-forward module's forward on the f32 stream in one launch (d = 256, csrc/ffn2h.hip; 0: LayerNorm + two GEMMs)
_H_FFN2_BWD = os.environ.get("SARSSL_HYBRID_FFN2_BWD", "1") != "0"   # the feed-forward module's data gradients in the fused launch (d = 256; 0: two GEMMs)
_H_DLN32 = os.environ.get("SARSSL_HYBRID_DLN32", "1") != "0"    # branch gradients entering the LayerNorm backward in f32 (0: bf16)


def wpair(p, view=None):
    """(hi, lo) fp16 shadows of parameter ``p`` (optionally viewed as ``view``)."""
    hi, lo = wt(p), wt_lo(p)
    return (hi.view(view), lo.view(view)) if view is not None else (hi, lo)


def mm_nt_h(x, w, out_dtype, **kw):
    """x [M,K] (hip.Pair or an fp16 tensor) @ (w_hi + w_lo)[N,K]^T -> [M,N] of ``out_dtype`` (fp16 | f32): nn.Linear forward, hybrid mode."""
    M, K = x.shape
    return hip.gemm_split(x, w[0], w[1], M=M, N=w[0].shape[0], K=K, out_dtype=out_dtype, **kw)


def _as_stream(x):
    """x as an f32 stream tensor (module-level entry points hand over whatever dtype their caller used)."""
    return x if x.dtype == _F32 else hip.cast(x.contiguous(), _F32)


def _ln_bwd_hd(dln, x, ln_mod, stats, dy, drop, want16):
    """LayerNorm backward of the hybrid mode: f32 stream gradient out; with ``want16`` also the bf16 operand of the next module of the
    backward chain - with its dropout backward applied when ``drop`` = (p, seed, gscale) is given, else the plain copy (the module
    multiplies a replayed tensor mask itself) -> (dx, dx16), or dx alone without ``want16``."""
    r = hip.layernorm_bwd_stream(dln, x, ln_mod.weight.data, stats, resid=dy, dgamma=gbuf(ln_mod.weight), dbeta=gbuf(ln_mod.bias),
                                 drop=drop if want16 else None, copy16=want16 and drop is None)
    if isinstance(r, tuple):
        r[1]._dropped = drop is not None
    return r


def _ln_bwd_h(dln, x, ln_mod, stats, dy, saved, next_kind, want16):
    return _ln_bwd_hd(dln, x, ln_mod, stats, dy, _next_drop(next_kind, saved) if want16 else None, want16)


def _grad16(dy, dy16, p, seed, gscale=1.0):
    """bf16 matrix-core operand of a module's backward pass from its incoming stream gradient: ``dy16`` when the previous LayerNorm backward
    already wrote it (with this module's dropout backward applied when dy16._dropped), else formed here."""
    if dy16 is not None:
        if getattr(dy16, "_dropped", False) or torch.is_tensor(seed) or (p <= 0 and gscale == 1.0):
            return dy16
        return hip.act_bwd(dy16, None, 0, p_drop=p, seed=seed, gscale=gscale)
    own = dy.dtype != torch.bfloat16              # (module-level entry points hand the gradient over in bf16 already)
    d16 = hip.cast(dy.contiguous(), torch.bfloat16) if own else dy.contiguous()
    if torch.is_tensor(seed) or (p <= 0 and gscale == 1.0):
        return d16
    return hip.act_bwd(d16, None, 0, p_drop=p, seed=seed, gscale=gscale, out=d16 if own else None)


def to_rt(x):
    return x if x.dtype == RT.dtype else hip.cast(x.contiguous(), RT.dtype)


def to_g(x):
    """x in the storage dtype of gradients (RT.gdtype)."""
    return x if x.dtype == RT.gdtype else hip.cast(x.contiguous(), RT.gdtype)


def _cached(module, name, builder):
    """Per-module cache of re-laid-out weights, invalidated whenever parameters change: by the fused optimizer / flat-buffer
    refresh (global version) or through torch itself (load_state_dict, torch.optim, in-place init: the parameter's own version
    counter and storage address)."""
    w = module.weight
    key = (weights_version(), RT.dtype, w._version, w.data_ptr())
    c = module.__dict__.setdefault("_wcache", {})
    hit = c.get(name)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            c[name] = (key, builder())
    return c[name][1]


# ------------------------------------------------------------------------------------------------ BatchNorm plumbing
def bn_affine(x, C, bn, train, sums=None, N=None):
    """BatchNorm{1,2}d affine for channels-last x: batch statistics (+ running-stat update) in train mode, running
    statistics in eval mode.  Returns aff = [scale, shift, mean, rstd] (4, C) f32.  ``sums``: statistics already
    accumulated by the producing kernel's epilogue (saves one pass over x; x may then be None with the element count N)."""
    if train:
        return hip.bn_train_affine(x, C, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, eps=bn.eps, momentum=bn.momentum, sums=sums, N=N)
    return hip.bn_eval_affine(C, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, eps=bn.eps)


def bn_param_grads(bn, red, C):
    """red = [sum g | sum g*xhat] (f64, 2C) -> dbeta, dgamma."""
    hip.f64_accum2(red, gbuf(bn.bias), gbuf(bn.weight))


# ------------------------------------------------------------------------------------------------ CNN stem
def _taps(conv):
    """(co,ci,3,3) -> forward taps [9][co][ci] (activation dtype) and data-gradient taps [9][ci][co] (flipped; gradient dtype)."""
    def build():                                              # one launch (rebuilt every step: the weights move)
        return hip.conv_taps(conv.weight.data.contiguous(), RT.dtype, RT.gdtype)
    return _cached(conv, "taps", build)


def _patch_w(conv, F, grad=False):
    """(d,4,F,1) -> [d][f*4+c] so the patch conv is a plain GEMM over the (B,T,F,4) tensor (grad: the gradient-side copy)."""
    dtype = RT.gdtype if grad else RT.dtype

    def build():
        return hip.patch_w(conv.weight.data.contiguous(), dtype)
    return _cached(conv, "patchw_g" if (grad and dtype != RT.dtype) else "patchw", build)


def _patch_w_pair(conv, F):
    """(hi, lo) fp16 pair of the [d][f*4+c] patch matrix (hybrid mode)."""
    def build():
        p = hip.split_pair(hip.patch_w(conv.weight.data.contiguous(), _F32))
        return p.hi, p.lo
    return _cached(conv, "patchw_pair", build)


def stem_fwd(a0, pe, train, saved):
    """``patch_embed`` (code/model.py:50-64) on channels-last a0 (B,F,T,4) -> [B*T, d]."""
    B, F, T, _ = a0.shape
    fuse = train and RT.dtype in _16                     # BatchNorm sums come out of the producing kernel's epilogue
    W1 = pe[0].weight.data.view(64, 4)
    y1 = y2 = mom1 = None
    if fuse and _C1IN and (B * F * T) % 64 == 0 and not RT.inference:
        # the first layer's 64-channel output is never stored: its BatchNorm statistics follow from the 4 + 10 moments of the
        # 4-channel input, the first 3x3 convolution (and, in backward, its weight gradient and the layer's own backward pass) form
        # relu(bn1(W1 a0)) from a0 while staging - 4 x 537 MB less HBM traffic per encoder and step at B = 64
        bn1 = pe[1]
        aff1, mom1 = hip.stem_c1_bn_affine(a0, W1, bn1.weight.data, bn1.bias.data, bn1.running_mean, bn1.running_var,
                                           bn1.num_batches_tracked, eps=bn1.eps, momentum=bn1.momentum)
        y2, s2 = hip.conv3x3_fwd_c1(a0, W1, aff1[0], aff1[1], _taps(pe[3])[0], want_stats=True)
    else:
        y1, s1 = hip.stem_c1_fwd(a0, W1, want_stats=True) if train else (hip.stem_c1_fwd(a0, W1), None)
        aff1 = bn_affine(y1, 64, pe[1], train, sums=s1)
        y2, s2 = hip.conv3x3_fwd(y1, _taps(pe[3])[0], aff1[0], aff1[1], want_stats=True) if fuse else \
            (hip.conv3x3_fwd(y1, _taps(pe[3])[0], aff1[0], aff1[1], precise=RT.precise), None)
    aff2 = bn_affine(y2, 64, pe[4], train, sums=s2)
    y3, s3 = hip.conv3x3_fwd(y2, _taps(pe[6])[0], aff2[0], aff2[1], want_stats=True) if fuse else \
        (hip.conv3x3_fwd(y2, _taps(pe[6])[0], aff2[0], aff2[1], precise=RT.precise), None)
    aff3 = bn_affine(y3, 64, pe[7], train, sums=s3)
    if RT.hybrid and _H_STEM4 and RT.dtype == torch.float16:
        # the f32 stream starts at the 4-channel tensors (0.5 MB per segment): the 64 -> 4 result and BatchNorm(4) + ReLU of it travel as
        # fp16 pairs into a three-segment frame-patch product; the hi halves are exactly the fp16 mode's y4 / z4 (what backward reads)
        if train:
            y4p, s4 = hip.stem_c4_fwd_pair(y3, pe[9].weight.data.view(4, 64), aff3[0], aff3[1], want_stats=True)
            aff4 = bn_affine(y4p.hi, 4, pe[10], train, sums=s4)
        else:
            y4p = hip.stem_c4_fwd_pair(y3, pe[9].weight.data.view(4, 64), aff3[0], aff3[1])
            aff4 = bn_affine(y4p.hi, 4, pe[10], train)
        z4p = hip.cl_affine_act_pair(y4p, 4, aff4, RELU)
        y4, z4 = y4p.hi, z4p.hi.view(B * T, F * 4)
        e = mm_nt_h(hip.Pair(z4, z4p.lo.view(B * T, F * 4)), _patch_w_pair(pe[12], F), _F32)
        saved.append((a0, (y1, mom1), aff1, y2, aff2, y3, aff3, y4, aff4, z4, train))
        return e
    if train:                       # BatchNorm(4) sums ride in the 64->4 pass (no statistics pass over y4)
        y4, s4 = hip.stem_c4_fwd(y3, pe[9].weight.data.view(4, 64), aff3[0], aff3[1], want_stats=True)        # (B,T,F,4)
        aff4 = bn_affine(y4, 4, pe[10], train, sums=s4)
    else:
        y4 = hip.stem_c4_fwd(y3, pe[9].weight.data.view(4, 64), aff3[0], aff3[1])
        aff4 = bn_affine(y4, 4, pe[10], train)
    z4 = hip.cl_affine_act(y4, 4, aff4, RELU).view(B * T, F * 4)
    if RT.hybrid:          # the stream starts here: f32 result, the weight as its fp16 pair
        e = mm_nt_h(z4, _patch_w_pair(pe[12], F), _F32)
    else:
        e = mm_nt(z4, _patch_w(pe[12], F), fp8=False)
    saved.append((a0, (y1, mom1), aff1, y2, aff2, y3, aff3, y4, aff4, z4, train))
    return e


# A/B switches of the round-2 work-removal passes (NOTES.md 4.1a / 4.6); bench.py prints their resolved state.  The paths measured
# slower in round 2 (BatchNorm-backward transform inside the convolution staging, consumer-side BatchNorm finalize, one-pass 64->4
# backward, the separate / half-fused first-layer backward) were removed from the engine in round 3.
_DGRAD_BNRED = os.environ.get("SARSSL_DGRAD_BNRED", "1") != "0"
_DWGLU = os.environ.get("SARSSL_DWGLU", "1") != "0"             # 0: separate glu / dwconv / cl_stats kernels (A/B runs)
_FUSED_ATTN = os.environ.get("SARSSL_FUSED_ATTN", "1") != "0"   # 0: GEMM + softmax-kernel attention core also in bf16 mode (A/B runs)
_ATTN_POS = os.environ.get("SARSSL_ATTN_POS", "1") != "0"       # 0: positional score by its own GEMM launch instead of inside the attention kernel (A/B runs)
# T > 256 (config 5): the forward kernel forms the shifted score itself, one 256-key block of its slab at a time (round 6; bit-identical to
# the score-GEMM sequence).  OFF by default: measured SLOWER on config 5 (same box, two interleaved rounds: 20.65 vs 20.33 ms fp16, 22.46 vs
# 22.16 ms hybrid) - per (query tile, key block) the band of position tiles is re-staged through LDS behind two barriers each at one wave
# per SIMD, which costs more than the chip-filling score GEMM it replaces plus the read of its 150 MB result
_ATTN_POS_LONG = os.environ.get("SARSSL_ATTN_POS_LONG", "0") != "0"
_C1IN = os.environ.get("SARSSL_C1IN", "1") != "0"             # 0: store the first layer's 64-channel output (A/B runs)
_C1RED = os.environ.get("SARSSL_C1RED", "1") != "0"           # 0: store the gradient w.r.t. that output and reduce it in a pass of its own


def knobs():
    """Resolved state of every environment switch that selects a compute path (bench.py prints it; a benchmark line is only
    comparable with another one under the same knobs)."""
    from . import runtime
    return {"SARSSL_WGRAD_GROUP": int(_WGRAD_GROUP), "SARSSL_WGRAD_CSUM": int(_WGRAD_CSUM), "SARSSL_DGRAD_BNRED": int(_DGRAD_BNRED), "SARSSL_DWGLU": int(_DWGLU),
            "SARSSL_FUSED_ATTN": int(_FUSED_ATTN), "SARSSL_ATTN_POS": int(_ATTN_POS), "SARSSL_ATTN_POS_LONG": int(_ATTN_POS_LONG), "SARSSL_C1IN": int(_C1IN), "SARSSL_C1RED": int(_C1RED),
            "SARSSL_FUSE_DROP_BWD": int(_FUSE_DROP_BWD), "SARSSL_FFN2": int(_FFN2), "SARSSL_FFN2_FWD": sorted(_FFN2_FWD), "SARSSL_FFN2_BWD": sorted(_FFN2_BWD), "SARSSL_FFN2_LN": int(_FFN2_LN), "SARSSL_HYBRID_ALO": sorted(_H_ALO), "SARSSL_LIN256": int(_LIN256), "SARSSL_DEC_MASKED": int(_DEC_MASKED), "SARSSL_TAIL_MASKED": int(_TAIL_MASKED), "SARSSL_PREP_ASYNC": int(_PREP_ASYNC), "SARSSL_WGRAD_SPLIT_BIG": [_WGRAD_SPLIT_BIG, _WGRAD_SPLIT_BIG_TILES], "SARSSL_TWO_STREAMS": os.environ.get("SARSSL_TWO_STREAMS", "1"),
            "SARSSL_STEM_LAST_ALL_CUS": int(_STEM_LAST_ALL_CUS), "SARSSL_WGRAD_WS": os.environ.get("SARSSL_WGRAD_WS", "1"),
            "SARSSL_CONV_WS": os.environ.get("SARSSL_CONV_WS", "4"),
            "SARSSL_CONV_CUS_FWD": os.environ.get("SARSSL_CONV_CUS_FWD", os.environ.get("SARSSL_CONV_CUS", "default(256)")),
            "SARSSL_CONV_CUS_BWD": os.environ.get("SARSSL_CONV_CUS_BWD", os.environ.get("SARSSL_CONV_CUS", "default(224)")),
            "SARSSL_GRAPH": os.environ.get("SARSSL_GRAPH", "default"),
            "precision": runtime.get_precision()}


def patch_bwd(de, pe, saved):
    """Backward of the frame-patch convolution (the last layer of ``patch_embed``, a plain GEMM here): accumulates its weight
    gradient - the stem's only sizeable parameter, so the data-parallel bucket holding it can be reduced before the long
    parameter-poor remainder of the stem backward - and returns the gradient w.r.t. the (B,T,F,4) activations."""
    a0, z4 = saved[-1][0], saved[-1][9]
    B, F, T, _ = a0.shape
    d = de.shape[1]
    if RT.hybrid and de.dtype == _F32:        # the stream's gradient leaves f32 here: bf16 operand of the two products below
        de = getattr(de, "_g16", None) if getattr(de, "_g16", None) is not None else hip.cast(de.contiguous(), torch.bfloat16)
    if RT.dtype in _16 and RT.replay is None:   # split-K partials folded and re-laid-out in one pass (nothing zeroed)
        ws, nslice = hip.gemm_tn_partials(de, z4, _wgrad_split(de.shape[0], d, F * 4, False))
        hip.patch_wgrad_accum(ws, gbuf(pe[12].weight), nslice)
    else:
        gtmp = torch.zeros((d, F * 4), dtype=torch.float32, device=de.device)
        mm_tn_acc(de, z4, gtmp, group=False)
        hip.patch_wgrad_accum(gtmp, gbuf(pe[12].weight))
    return mm_nn(de, _patch_w(pe[12], F, grad=True), fp8=False)                            # (B,T,F,4)


def stem_bwd(dz4, pe, saved):
    """Backward of the CNN stem below the patch GEMM (``dz4`` = patch_bwd's result)."""
    a0, (y1, mom1), aff1, y2, aff2, y3, aff3, y4, aff4, z4, train = saved.pop()
    B, F, T, _ = a0.shape
    W1 = pe[0].weight.data.view(64, 4)
    red4 = hip.cl_bn_bwd_reduce(dz4, y4, 4, aff4, RELU)
    # (BatchNorm / 1x1-conv parameter gradients are added from the finished sums by workgroup 0 of the pass that consumes them)
    dy4 = hip.cl_bn_bwd_apply(dz4, y4, 4, aff4, RELU, False, train, red4, pgrads=(gbuf(pe[10].weight), gbuf(pe[10].bias)))
    # 64->4 conv + BN3/ReLU backward: sums pass + direct dy3 pass over y3 (1.7 GB per encoder)
    dy3, red = hip.stem_c4_bwd_two_phase(y3, dy4, pe[9].weight.data.view(4, 64), aff3, train,
                                         pgrads=(gbuf(pe[9].weight), gbuf(pe[7].weight), gbuf(pe[7].bias)))
    # second 3x3 conv
    dW = hip.conv3x3_wgrad(dy3, y2, aff2[0], aff2[1], precise=RT.precise, acc_into=gbuf(pe[6].weight))
    if dW is not None:
        gbuf(pe[6].weight).add_(dW.view(3, 3, 64, 64).permute(2, 3, 0, 1))
    red2 = None
    if _DGRAD_BNRED and RT.dtype in _16:              # BatchNorm-backward sums accumulated in the data-gradient kernel's epilogue
        dz2, red2 = hip.conv3x3_dgrad_bnred(dy3, _taps(pe[6])[1], y2, aff2)
    else:
        dz2 = hip.conv3x3_fwd(dy3, _taps(pe[6])[1], precise=RT.precise)
        red2 = hip.cl_bn_bwd_reduce(dz2, y2, 64, aff2, RELU)
    # first 3x3 conv
    dy2 = hip.cl_bn_bwd_apply(dz2, y2, 64, aff2, RELU, False, train, red2, out=dz2, pgrads=(gbuf(pe[4].weight), gbuf(pe[4].bias)))
    if y1 is None:                     # first layer never stored (stem_fwd): its operand is formed from a0 while staging
        hip.conv3x3_wgrad_c1(dy2, a0, W1, aff1[0], aff1[1], gbuf(pe[3].weight))
        if _C1RED and hip.conv3x3_dgrad_c1red(dy2, _taps(pe[3])[1], a0, W1, aff1, mom1, train, gbuf(pe[0].weight),
                                              gbuf(pe[1].weight), gbuf(pe[1].bias)):
            return None     # the data gradient of the first 3x3 convolution was consumed in its epilogue: the first layer is done
    else:
        dW = hip.conv3x3_wgrad(dy2, y1, aff1[0], aff1[1], precise=RT.precise, acc_into=gbuf(pe[3].weight))
        if dW is not None:
            gbuf(pe[3].weight).add_(dW.view(3, 3, 64, 64).permute(2, 3, 0, 1))
    dz1 = hip.conv3x3_fwd(dy2, _taps(pe[3])[1], precise=RT.precise)
    # everything the first layer needs from (dz1, y1, a0) in one pass: BN sums, dgamma/dbeta, dW1
    if y1 is None:
        hip.stem_c1_bwd_a0(dz1, a0, W1, aff1, train, gbuf(pe[0].weight), gbuf(pe[1].weight), gbuf(pe[1].bias))
    else:
        hip.stem_c1_bwd(dz1, y1, a0, aff1, train, gbuf(pe[0].weight), gbuf(pe[1].weight), gbuf(pe[1].bias))
    return None        # the stem input is data


# ------------------------------------------------------------------------------------------------ Conformer modules
def _p(drop, train):
    return float(drop.p) if train else 0.0


def _replaying(train):
    return train and RT.replay is not None


_DEC_MASKED = os.environ.get("SARSSL_DEC_MASKED", "1") != "0"   # training steps run the decoder on the masked frames only (model._PretrainFn; 0: every frame)
_PREP_ASYNC = os.environ.get("SARSSL_PREP_ASYNC", "1") != "0"    # weight-only launches of a step on the side stream, next to the front-end (model.py)
_TAIL_MASKED = os.environ.get("SARSSL_TAIL_MASKED", "1") != "0"  # ... and the row-wise tail of each encoder's last block (second feed-forward module + closing LayerNorm)
_FFN2 = os.environ.get("SARSSL_FFN2", "1") != "0"             # 0: the feed-forward module as two GEMM launches (A/B runs)
# Model widths that take the fused forward / backward launch.  Default: d = 256 only (the spat encoder - the step's critical chain).  The
# fused launch owns whole CUs (512 threads, 101-134 KB of LDS); at d = 512 (spec encoder) it runs 100-160 us during which the other
# encoder's stream gets no CU, and the step is SLOWER with it (same box, two rounds: off 10.69 / 10.70 ms, d = 256 only 10.64 / 10.59,
# d = 512 only 10.86 / 10.83, both 10.78 / 10.71) although the launch itself beats its two GEMMs alone (101 vs 121 us forward).
_FFN2_LN = os.environ.get("SARSSL_FFN2_LN", "1") != "0"       # the module's LayerNorm (forward / backward) inside the fused launches (0: own launches)
_FFN2_FWD = set(int(v) for v in os.environ.get("SARSSL_FFN2_FWD", "256").split(",") if v)
_FFN2_BWD = set(int(v) for v in os.environ.get("SARSSL_FFN2_BWD", "256").split(",") if v)


def prepare_ffn_packs(ffs, need_bwd=True):
    """Fragment-order packs of the feed-forward modules' weights for the fused kernel (csrc/ffn2.hip): W1, W2 in the forward dtype and,
    for the backward launch, W2^T, W1^T in the gradient dtype - all modules of ``ffs`` whose packs are stale in ONE launch.  Buffers are
    persistent per module (a captured step rewrites them in place from the shadow weights the Adam kernel wrote)."""
    jobs, fresh = [], []
    for ff in ffs:
        l1, l2 = ff.sequential[1].linear, ff.sequential[4].linear
        if l1.weight.shape[1] not in _FFN2_FWD and l1.weight.shape[1] not in _FFN2_BWD:
            continue
        w1, w2 = wt(l1.weight), wt(l2.weight)
        if w1.dtype not in _16 or not w1.is_cuda:
            continue
        hyb = RT.hybrid and _H_FFN2_FWD                    # hybrid mode: packs of the lo shadows as well (csrc/ffn2h.hip)
        key = (weights_version(), w1.dtype, l1.weight._version, l2.weight._version, w1.data_ptr(), bool(need_bwd), hyb)
        c = ff.__dict__.get("_ffn2_packs")
        if c is not None and (c[0] == key or (c[0][:5] == key[:5] and c[0][5] and c[0][6] == hyb)):
            continue
        bufs = ff.__dict__.get("_ffn2_bufs")
        if bufs is None or bufs[0].dtype != w1.dtype or bufs[0].device != w1.device or (hyb and len(bufs) < 6):
            gd = RT.gdtype
            bufs = ff.__dict__["_ffn2_bufs"] = (torch.empty(w1.numel(), dtype=w1.dtype, device=w1.device), torch.empty(w2.numel(), dtype=w1.dtype, device=w1.device),
                                                torch.empty(w2.numel(), dtype=gd, device=w1.device), torch.empty(w1.numel(), dtype=gd, device=w1.device))
            if hyb:
                bufs = ff.__dict__["_ffn2_bufs"] = bufs + (torch.empty(w1.numel(), dtype=w1.dtype, device=w1.device), torch.empty(w2.numel(), dtype=w1.dtype, device=w1.device))
        jobs += [(w1, bufs[0]), (w2, bufs[1])]
        if hyb:
            jobs += [(wt_lo(l1.weight), bufs[4]), (wt_lo(l2.weight), bufs[5])]
        if need_bwd:
            jobs += [(wtg(l2.weight).t(), bufs[2]), (wtg(l1.weight).t(), bufs[3])]
        fresh.append((ff, key, bufs))
    if jobs:
        hip.ffn_pack(jobs)
    for ff, key, bufs in fresh:
        ff.__dict__["_ffn2_packs"] = (key, bufs)


# LayerNorm + q/k/v / first pointwise convolution (and their data gradient + LayerNorm backward) of the d = 256 blocks as one tile-resident
# launch each (csrc/lin256.hip).  OFF by default: alone the launches beat the pairs they replace (LayerNorm + q/k/v 21.2 vs 27.4 us, data
# gradient + LayerNorm backward 33.0 vs 37.7 us; tools/bench_lin256.py), inside the two-stream step they do not - same box, three interleaved
# rounds: 10.67 / 10.67 / 10.66 ms with, 10.64 / 10.62 / 10.65 ms without.  Like the d = 512 feed-forward launch, a 512-thread / 100 KB
# workgroup owns its CU, and what it displaces from the other encoder's stream costs what it saves (NOTES.md section 10).
_LIN256 = os.environ.get("SARSSL_LIN256", "0") != "0"


def _lin_pack_jobs(owner, name, w_fwd, w_bwd, need_bwd):
    """Pack jobs (and the cache entry to commit) of one Linear weight [N, K]: forward pack of W in the forward dtype, pack of W^T in the
    gradient dtype; persistent buffers on ``owner`` (an nn.Module), stale when the weights' version moved."""
    key = (weights_version(), w_fwd.dtype, w_fwd.data_ptr(), bool(need_bwd))
    c = owner.__dict__.get("_lin256_" + name)
    if c is not None and (c[0] == key or (c[0][:3] == key[:3] and c[0][3])):
        return [], None
    bufs = owner.__dict__.get("_lin256_bufs_" + name)
    if bufs is None or bufs[0].dtype != w_fwd.dtype or bufs[0].device != w_fwd.device or bufs[0].numel() != w_fwd.numel():
        bufs = owner.__dict__["_lin256_bufs_" + name] = (torch.empty(w_fwd.numel(), dtype=w_fwd.dtype, device=w_fwd.device),
                                                         torch.empty(w_fwd.numel(), dtype=RT.gdtype, device=w_fwd.device))
    jobs = [(w_fwd, bufs[0])]
    if need_bwd:
        jobs.append((w_bwd.t(), bufs[1]))
    return jobs, ("_lin256_" + name, key, bufs)


def _lin256_weights(blk_mod, kind):
    """(forward view [N, K], gradient-side view) of the layer a tile-resident launch replaces: 'qkv' of an attention module (the three
    projections back to back in the flat buffers) or 'pw1' of a convolution module; None when the layout does not allow it."""
    if kind == "qkv":
        att = blk_mod.attention
        f, gsd = _qkv_views(att), _qkv_views(att, grad=True)
        return None if f is None or gsd is None else (f[0], gsd[0])
    pw1 = blk_mod.sequential[2].conv
    d = pw1.weight.shape[1]
    return wt(pw1.weight).view(2 * d, d), wtg(pw1.weight).view(2 * d, d)


def prepare_lin256_packs(mods, need_bwd=True):
    """mods: [(module, kind)] - packs of every stale weight in one launch (model._PretrainFn: once per step for the whole spat encoder)."""
    jobs, commits = [], []
    for m, kind in mods:
        ws = _lin256_weights(m, kind)
        if ws is None or ws[0].dtype not in _16 or not ws[0].is_cuda or ws[0].shape[1] != 256:
            continue
        j, c = _lin_pack_jobs(m, kind, ws[0], ws[1], need_bwd)
        jobs += j
        if c is not None:
            commits.append((m, c))
    if jobs:
        hip.ffn_pack(jobs)
    for m, (name, key, bufs) in commits:
        m.__dict__[name] = (key, bufs)


def _lin256_pack(m, kind, need_bwd=True):
    prepare_lin256_packs([(m, kind)], need_bwd)
    c = m.__dict__.get("_lin256_" + kind)
    return None if c is None else c[1]


def block_lin256_mods(enc):
    return [(blk.sequential[1].module, "qkv") for blk in enc.layers] + [(blk.sequential[2].module, "pw1") for blk in enc.layers]


def _ffn_packs(ff, need_bwd=True):
    prepare_ffn_packs([ff], need_bwd)
    return ff.__dict__["_ffn2_packs"][1]


def block_ffns(enc):
    """The feed-forward modules of a ConformerEncoder (two per block) - for one pack launch per step (model._PretrainFn)."""
    return [blk.sequential[i].module for blk in enc.layers for i in (0, 3)]


def _ffn_fwd_h(x, ff, factor, train, saved, out=None):
    """ffn_fwd in the hybrid mode: x / result f32, LayerNorm output as an fp16 pair, hidden activation and saved pre-activation fp16."""
    seq = ff.sequential
    x = _as_stream(x)
    pre = x.__dict__.pop("_pre_ln", None)
    p1, p2 = _p(seq[3], train), _p(seq[5], train)
    l1, l2 = seq[1].linear, seq[4].linear
    d = x.shape[1]
    if (_FFN2 and _H_FFN2_FWD and d in _FFN2_FWD and pre is None and not _replaying(train) and x.stride(1) == 1 and hip.ffn2h_supported(x.shape[0], d)
            and l1.weight.shape[0] == 4 * d and wt(l1.weight).is_cuda):
        # LayerNorm + Linear + Swish + Dropout + Linear + Dropout + scaled residual on the f32 stream in one launch (csrc/ffn2h.hip)
        packs = _ffn_packs(ff, need_bwd=not RT.inference)
        s1, s2 = (RT.next_seed() if p1 > 0 else 0), (RT.next_seed() if p2 > 0 else 0)
        y, hpre, a, lnh, stats = hip.ffn2h_fwd(x, seq[0].weight.data, seq[0].bias.data, seq[0].eps, packs[0], packs[4], packs[1], packs[5],
                                               l1.bias.data, l2.bias.data, d, p1=p1, s1=s1, p2=p2, s2=s2, out_scale=factor, out=out,
                                               act_pair="ffn1" in _H_ALO)
        saved.append((x, lnh, stats, hpre, a, p1, s1, p2, s2, factor))
        return y
    if pre is not None and pre[0] is seq[0]:
        ln, stats = pre[1], pre[2]
    else:
        ln, stats = hip.layernorm_fwd_pair(x, seq[0].weight.data, seq[0].bias.data, seq[0].eps, want_lo="ffn1" in _H_ALO)
    hpre = torch.empty((x.shape[0], l1.weight.shape[0]), dtype=torch.float16, device=x.device)
    if _replaying(train) and (p1 > 0 or p2 > 0):          # host-drawn masks in the reference's order: hidden, then output
        a = mm_nt_h(ln, wpair(l1.weight), torch.float16, bias=l1.bias.data, act=SWISH, preact=hpre)
        s1 = RT.replay.mask(tuple(a.shape), p1, x.device, a.dtype) if p1 > 0 else 0
        if p1 > 0:
            a = a * s1
        y = mm_nt_h(a, wpair(l2.weight), _F32, bias=l2.bias.data)
        s2 = RT.replay.mask(tuple(y.shape), p2, x.device, _F32) if p2 > 0 else 0
        y = torch.add(x, y * s2 if p2 > 0 else y, alpha=factor)
        if out is not None:
            out.copy_(y)
            y = out
    else:
        s1, s2 = (RT.next_seed() if p1 > 0 else 0), (RT.next_seed() if p2 > 0 else 0)
        a = mm_nt_h(ln, wpair(l1.weight), torch.float16, bias=l1.bias.data, act=SWISH, preact=hpre, p_drop=p1, seed=s1)
        y = mm_nt_h(a, wpair(l2.weight), _F32, bias=l2.bias.data, p_drop=p2, seed=s2, out_scale=factor, resid=x, ldr=x.stride(0),
                    res_scale=1.0, out=out, ldc=(out.stride(0) if out is not None else None))
    saved.append((x, ln.hi, stats, hpre, a, p1, s1, p2, s2, factor))
    return y


def _ffn_bwd_h(dy, ff, saved, dy16=None, next_kind=None, want16=True):
    """-> (dx f32, dx16) - the stream gradient and its bf16 copy for the next module of the backward chain - or dx alone (want16 False)."""
    x, ln, stats, hpre, a, p1, s1, p2, s2, factor = saved.pop()
    seq = ff.sequential
    l1, l2 = seq[1].linear, seq[4].linear
    dz2 = _grad16(dy, dy16, p2, s2, factor)
    if torch.is_tensor(s2):
        dz2 = dz2 * (s2 * factor).to(dz2.dtype)
    mm_tn_acc(dz2, a, gbuf(l2.weight), bias=gbuf(l2.bias))
    d = x.shape[1]
    if (_FFN2 and _H_FFN2_BWD and d in _FFN2_BWD and not torch.is_tensor(s1) and dz2.stride(1) == 1 and hip.ffn2_supported(x.shape[0], d, dz2.dtype)
            and hpre.shape[1] == 4 * d and ff.__dict__.get("_ffn2_packs") is not None and ff.__dict__["_ffn2_packs"][0][5]
            and ff.__dict__["_ffn2_packs"][0][0] == weights_version()):
        # both data-gradient products in the fused launch of the fp16 mode (csrc/ffn2.hip; bf16 gradients, fp16 saved pre-activation):
        # the branch gradient dln reaches the stream's LayerNorm backward in bf16
        dln, dh = hip.ffn2_bwd(dz2, _ffn_packs(ff)[2], _ffn_packs(ff)[3], hpre, d, p1=p1, s1=s1)
        mm_tn_acc(dh, ln, gbuf(l1.weight), bias=gbuf(l1.bias))
        return _ln_bwd_h(dln, x, seq[0], stats, _as_stream(dy), saved, next_kind, want16)
    if torch.is_tensor(s1):
        dh = mm_nn(dz2, wtg(l2.weight), aux=hpre, aux_act=SWISH)
        dh = dh * s1.to(dh.dtype)
    else:
        dh = mm_nn(dz2, wtg(l2.weight), aux=hpre, aux_act=SWISH, p_drop=p1, seed=s1)
    mm_tn_acc(dh, ln, gbuf(l1.weight), bias=gbuf(l1.bias))
    dln = mm_nn(dh, wtg(l1.weight), out_dtype=_F32 if _H_DLN32 else None)
    return _ln_bwd_h(dln, x, seq[0], stats, _as_stream(dy), saved, next_kind, want16)


def ffn_fwd(x, ff, factor, train, saved, out=None):
    """x + factor * FeedForwardModule(x)  (conformer/feed_forward.py:47-57, Conformer.py:60-67)."""
    if RT.hybrid:
        return _ffn_fwd_h(x, ff, factor, train, saved, out=out)
    seq = ff.sequential
    pre = x.__dict__.pop("_pre_ln", None)            # (block_fwd of the previous block already normalised this very tensor for us)
    p1, p2 = _p(seq[3], train), _p(seq[5], train)
    d = x.shape[1]
    fused = (_FFN2 and d in _FFN2_FWD and not _replaying(train) and hip.ffn2_supported(x.shape[0], d, x.dtype)
             and seq[1].linear.weight.shape[0] == 4 * d)
    if pre is not None and pre[0] is seq[0]:
        ln, stats = pre[1], pre[2]
    elif fused and _FFN2_LN and x.stride(1) == 1:
        ln = stats = None                           # the fused launch normalises its rows itself (bit-identical to the stand-alone kernel)
    else:
        ln, stats = hip.layernorm_fwd(x, seq[0].weight.data, seq[0].bias.data, seq[0].eps)
    if fused:
        # one launch for (LayerNorm +) Linear + Swish + Dropout + Linear + Dropout + scaled residual: the hidden tile stays on the CU (csrc/ffn2.hip)
        packs = _ffn_packs(ff, need_bwd=not RT.inference)
        s1, s2 = (RT.next_seed() if p1 > 0 else 0), (RT.next_seed() if p2 > 0 else 0)
        r = hip.ffn2_fwd(ln, packs[0], packs[1], seq[1].linear.bias.data, seq[4].linear.bias.data, x, d, p1=p1, s1=s1, p2=p2, s2=s2,
                         out_scale=factor, out=out, ln_in=None if ln is not None else (x, seq[0].weight.data, seq[0].bias.data, seq[0].eps))
        y, hpre, a = r[0], r[1], r[2]
        if ln is None:
            ln, stats = r[3], r[4]
        saved.append((x, ln, stats, hpre, a, p1, s1, p2, s2, factor))
        return y
    hpre = torch.empty((x.shape[0], seq[1].linear.weight.shape[0]), dtype=x.dtype, device=x.device)
    if _replaying(train) and (p1 > 0 or p2 > 0):          # host-drawn masks in the reference's order: hidden, then output
        a = mm_nt(ln, wt(seq[1].linear.weight), bias=seq[1].linear.bias.data, act=SWISH, preact=hpre)
        s1 = RT.replay.mask(tuple(a.shape), p1, x.device, x.dtype) if p1 > 0 else 0
        if p1 > 0:
            a = a * s1
        y = mm_nt(a, wt(seq[4].linear.weight), bias=seq[4].linear.bias.data)
        s2 = RT.replay.mask(tuple(y.shape), p2, x.device, x.dtype) if p2 > 0 else 0
        y = torch.add(x, y * s2 if p2 > 0 else y, alpha=factor)
        if out is not None:
            out.copy_(y)
            y = out
        saved.append((x, ln, stats, hpre, a, p1, s1, p2, s2, factor))
        return y
    s1, s2 = (RT.next_seed() if p1 > 0 else 0), (RT.next_seed() if p2 > 0 else 0)
    a = mm_nt(ln, wt(seq[1].linear.weight), bias=seq[1].linear.bias.data, act=SWISH, preact=hpre, p_drop=p1, seed=s1)
    y = mm_nt(a, wt(seq[4].linear.weight), bias=seq[4].linear.bias.data, p_drop=p2, seed=s2, out_scale=factor,
              resid=x, ldr=x.stride(0), res_scale=1.0, out=out, ldc=(out.stride(0) if out is not None else None))
    saved.append((x, ln, stats, hpre, a, p1, s1, p2, s2, factor))
    return y


_FUSE_DROP_BWD = os.environ.get("SARSSL_FUSE_DROP_BWD", "1") != "0"


def _next_drop(kind, saved):
    """(p, seed, gscale) of the dropout backward the NEXT module of the backward chain (``kind``: 'ffn' | 'conv' | 'mhsa', its entry
    is on top of ``saved``) applies to its incoming gradient, or None when there is nothing to apply (or the masks are replayed host
    tensors).  The LayerNorm backward that produces that gradient then writes the dropped copy as a second output
    (hip.layernorm_bwd(drop=...)) instead of a separate act_bwd pass."""
    if not _FUSE_DROP_BWD or kind is None or kind == "copy" or not saved:
        return None
    e = saved[-1]
    if kind == "ffn":
        p, seed, g = e[7], e[8], e[9]
    elif kind == "conv":
        p, seed, g = e[8], e[9], 1.0
    else:
        p, seed, g = e[-4], e[-3], 1.0
    if torch.is_tensor(seed) or torch.is_tensor(p) or (p <= 0 and g == 1.0):
        return None
    return (float(p), int(seed), float(g))


def ffn_bwd(dy, ff, saved, dy_dropped=None, next_kind=None):
    if RT.hybrid:
        return _ffn_bwd_h(dy, ff, saved, dy16=dy_dropped, next_kind=next_kind, want16=next_kind is not None)
    x, ln, stats, hpre, a, p1, s1, p2, s2, factor = saved.pop()
    seq = ff.sequential
    if torch.is_tensor(s1) or torch.is_tensor(s2):        # replayed masks (see ffn_fwd)
        dz2 = (dy * s2.to(dy.dtype) if torch.is_tensor(s2) else dy) * factor
        mm_tn_acc(dz2, a, gbuf(seq[4].linear.weight), bias=gbuf(seq[4].linear.bias))
        dh = mm_nn(dz2, wtg(seq[4].linear.weight), aux=hpre, aux_act=SWISH)
        if torch.is_tensor(s1):
            dh = dh * s1.to(dh.dtype)
    else:
        if dy_dropped is not None:
            dz2 = dy_dropped                                  # written by the previous LayerNorm backward (see _next_drop)
        else:
            dz2 = hip.act_bwd(dy, None, 0, p_drop=p2, seed=s2, gscale=factor) if (p2 > 0 or factor != 1.0) else dy
        mm_tn_acc(dz2, a, gbuf(seq[4].linear.weight), bias=gbuf(seq[4].linear.bias))
        d = x.shape[1]
        if (_FFN2 and d in _FFN2_BWD and dz2.dtype in _16 and dz2.stride(1) == 1 and hip.ffn2_supported(x.shape[0], d, dz2.dtype)
                and hpre.shape[1] == 4 * d and ff.__dict__.get("_ffn2_packs") is not None and ff.__dict__["_ffn2_packs"][0][5]):
            # both data-gradient products in one launch: dh = (dz2 W2) * mask * swish'(hpre) leaves the chip once (the two weight-gradient
            # products read it), dln = dh W1 is formed from the LDS-resident tile (csrc/ffn2.hip, packs of the transposed weights)
            packs = _ffn_packs(ff)
            if _FFN2_LN and x.dtype == hpre.dtype and x.stride(1) == 1 and dy.stride(1) == 1:
                # ... and the LayerNorm backward of the module's first layer in the same launch's epilogue (dln never leaves the chip)
                dx, dh = hip.ffn2_bwd(dz2, packs[2], packs[3], hpre, d, p1=p1, s1=s1,
                                      ln_bwd=(x, seq[0].weight.data, stats, dy, gbuf(seq[0].weight), gbuf(seq[0].bias), _next_drop(next_kind, saved)))
                mm_tn_acc(dh, ln, gbuf(seq[1].linear.weight), bias=gbuf(seq[1].linear.bias))
                return dx
            dln, dh = hip.ffn2_bwd(dz2, packs[2], packs[3], hpre, d, p1=p1, s1=s1)
            mm_tn_acc(dh, ln, gbuf(seq[1].linear.weight), bias=gbuf(seq[1].linear.bias))
            return hip.layernorm_bwd(dln, x, seq[0].weight.data, stats, resid=dy, dgamma=gbuf(seq[0].weight), dbeta=gbuf(seq[0].bias),
                                     drop=_next_drop(next_kind, saved))
        # dh = (dz2 @ W2) * dropout_mask1 * swish'(hpre): activation backward fused into the GEMM epilogue
        dh = mm_nn(dz2, wtg(seq[4].linear.weight), aux=hpre, aux_act=SWISH, p_drop=p1, seed=s1)
    mm_tn_acc(dh, ln, gbuf(seq[1].linear.weight), bias=gbuf(seq[1].linear.bias))
    dln = mm_nn(dh, wtg(seq[1].linear.weight))
    return hip.layernorm_bwd(dln, x, seq[0].weight.data, stats, resid=dy, dgamma=gbuf(seq[0].weight), dbeta=gbuf(seq[0].bias),
                             drop=_next_drop(next_kind, saved))


def _pe(mod, T):
    """First T rows of the sinusoid table in the runtime dtype (persistent buffer 'pe', embedding.py:31-39)."""
    key = (T, RT.dtype)
    c = mod.__dict__.setdefault("_pecache", {})
    if key not in c:
        c[key] = to_rt(mod.positional_encoding.pe[0, :T].contiguous())
    return c[key]


def _pos_proj(mod, T):
    """The module's positional projection linear_pos(PE[:T]) [T, d] (attention.py:84): batch-invariant and a function of the weights only,
    so it is cached per weight version like the re-laid-out taps - prepare_step_weights forms it off the encoders' chains."""
    lin = mod.attention.pos_proj.linear
    return _cached(lin, "pos%d" % T, lambda: mm_nt(_pe(mod, T), wt(lin.weight), fp8=False))


def prepare_step_weights(net, F, T, need_bwd=True):
    """Everything a step derives from the PARAMETERS ALONE - re-laid-out 3x3 taps, patch-conv matrices, fragment-order feed-forward
    packs, the positional projections - in one place, so that model._PretrainFn can issue it on the side stream while the front-end /
    masking launches run (a dozen 5-20 us launches that otherwise sit in the two encoders' chains).  The point-of-use helpers
    (_taps, _patch_w, _pos_proj, _ffn_packs) then hit their caches."""
    encs = (net.spec_encoder, net.spat_encoder)
    if _FFN2 and RT.dtype in _16 and RT.replay is None:
        # fragment-order packs of every feed-forward module's weights for the fused kernel: one launch per step
        prepare_ffn_packs(block_ffns(encs[0].embed) + block_ffns(encs[1].embed), need_bwd=need_bwd)
        if _LIN256:
            prepare_lin256_packs(block_lin256_mods(encs[0].embed) + block_lin256_mods(encs[1].embed), need_bwd=need_bwd)
    for enc in encs:
        pe = enc.patch_embed
        _taps(pe[3]); _taps(pe[6])
        if RT.hybrid and _H_PREP:
            _patch_w_pair(pe[12], F)                         # (the forward's operand; the backward takes the gradient-side matrix below)
        else:
            _patch_w(pe[12], F)
        if need_bwd:
            _patch_w(pe[12], F, grad=True)
        for blk in enc.embed.layers:
            # hybrid: the pair-accurate projection is what both passes use (round 6: it used to be formed inside the blocks' chains - four
            # 12-19 us launches - while the fp16 mode's projection was still formed here, unused)
            (_pos_proj_h if (RT.hybrid and _H_PREP) else _pos_proj)(blk.sequential[1].module, T)


def _adjacent(ts):
    """Back-to-back views of ONE storage (as laid out by runtime.FlatParams)."""
    base = ts[0].untyped_storage().data_ptr()
    return all(t.untyped_storage().data_ptr() == base for t in ts) and \
        all(ts[i + 1].data_ptr() == ts[i].data_ptr() + ts[i].numel() * ts[i].element_size() for i in range(len(ts) - 1))


def _qkv_views(att, grad=False):
    """([3d,d] weight view, [3d] bias view, their gradient views) when q/k/v parameters are contiguous in memory
    (runtime.FlatParams lays them out that way), else None.  grad: the weight view of the backward pass (gradient-side dtype)."""
    projs = (att.query_proj.linear, att.key_proj.linear, att.value_proj.linear)
    ws = [(wtg if grad else wt)(l.weight) for l in projs]
    bs = [l.bias.data for l in projs]
    gw = [gbuf(l.weight) for l in projs]
    gb = [gbuf(l.bias) for l in projs]
    if not (_adjacent(ws) and _adjacent(bs) and _adjacent(gw) and _adjacent(gb)):
        return None
    d = att.d_model
    return (torch.as_strided(ws[0], (3 * d, d), (d, 1)), torch.as_strided(bs[0], (3 * d,), (1,)),
            torch.as_strided(gw[0], (3 * d, d), (d, 1)), torch.as_strided(gb[0], (3 * d,), (1,)))


def _pos_proj_h(mod, T):
    """_pos_proj in the hybrid mode: the sinusoid table and the weight as fp16 pairs, result fp16 (an attention operand)."""
    lin = mod.attention.pos_proj.linear

    def build():
        pe = hip.split_pair(mod.positional_encoding.pe[0, :T].contiguous().float())
        return mm_nt_h(pe, wpair(lin.weight), torch.float16)
    return _cached(lin, "posh%d" % T, build)


def _qkv_views_lo(att):
    """[3d, d] view of the q / k / v weights' lo shadows (laid out like the hi shadows, see _qkv_views)."""
    ws = [wt_lo(l.weight) for l in (att.query_proj.linear, att.key_proj.linear, att.value_proj.linear)]
    if not _adjacent(ws):
        return None
    d = att.d_model
    return torch.as_strided(ws[0], (3 * d, d), (d, 1))


def _mhsa_fwd_h(x, mod, B, T, train, saved):
    """mhsa_fwd in the hybrid mode: x / result f32; q, k, v, the attention core and its context are the fp16 kernels of the fp16 mode."""
    att = mod.attention
    H, dh, d = att.num_heads, att.d_head, att.d_model
    M = B * T
    x = _as_stream(x)
    ln, stats = hip.layernorm_fwd_pair(x, mod.layer_norm.weight.data, mod.layer_norm.bias.data, mod.layer_norm.eps, want_lo="qkv" in _H_ALO)
    fused = _qkv_views(att)
    fused_lo = _qkv_views_lo(att) if fused is not None else None
    if fused is not None and fused_lo is not None:
        qkv = mm_nt_h(ln, (fused[0], fused_lo), torch.float16, bias=fused[1])
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    else:
        q = mm_nt_h(ln, wpair(att.query_proj.linear.weight), torch.float16, bias=att.query_proj.linear.bias.data)
        k = mm_nt_h(ln, wpair(att.key_proj.linear.weight), torch.float16, bias=att.key_proj.linear.bias.data)
        v = mm_nt_h(ln, wpair(att.value_proj.linear.weight), torch.float16, bias=att.value_proj.linear.bias.data)
    ldk = k.stride(0)
    pe = _pe(mod, T)
    pos = _pos_proj_h(mod, T)
    nbh = B * H
    pa = _p(att.dropout, train)
    sa = RT.next_seed() if pa > 0 else 0
    scale = 1.0 / math.sqrt(d)
    replay = _replaying(train) and (pa > 0 or _p(mod.dropout, train) > 0)
    fused_attn = _FUSED_ATTN and not replay and hip.relpos_attn_supported(T, dh, RT.dtype)
    in_kernel = fused_attn and _ATTN_POS and hip.relpos_attn_pos_supported(T, dh, RT.dtype)
    ub, vb = att.u_bias.data.view(-1), att.v_bias.data.view(-1)
    qu, qv = (q, None) if in_kernel else hip.bias2(q, ub, vb)
    po = _p(mod.dropout, train)
    wo = wpair(att.out_proj.linear.weight)
    if fused_attn:
        if in_kernel:
            ctx, lse, bias = hip.relpos_attn_fwd_pos(q, q, k, v, pos, B, H, T, dh, scale, pa, sa, need_bwd=not RT.inference, biases=(ub, vb),
                                                     pair=_H_CTX)            # (the kernel writes the context's lo half itself)
        elif _ATTN_POS_LONG and hip.relpos_attn_pos_long_supported(T, dh, RT.dtype):
            ctx, lse, bias = hip.relpos_attn_fwd_pos_long(qu, qv, k, v, pos, B, H, T, dh, scale, pa, sa, need_bwd=not RT.inference, want_ctx32=_H_CTX)
        else:
            bias = torch.empty((B, H, T, T), dtype=RT.dtype, device=x.device)
            hip.gemm(qv, pos, M=T, N=T, K=dh, lda=d, ldb=d, nbatch=nbh, batch_inner=H, sA=(T * d, dh), sB=(0, dh), out=bias, ldc=T,
                     sC=(H * T * T, T * T), c_row_shift=True)
            ctx, lse = hip.relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, pa, sa, need_bwd=not RT.inference, want_ctx32=_H_CTX)
        so = RT.next_seed() if po > 0 else 0
        # the kernel's unrounded f32 context (it keeps it for the backward pass anyway) enters the output projection as a pair
        if isinstance(ctx, hip.Pair):
            cin, ctx = ctx, ctx.hi
        else:
            cin = hip.Pair(ctx, hip.split_pair(lse[0], want_hi=False)) if (_H_CTX and lse[0] is not None) else ctx
        y = mm_nt_h(cin, wo, _F32, bias=att.out_proj.linear.bias.data, p_drop=po, seed=so, resid=x, ldr=x.stride(0), res_scale=1.0)
        saved.append((x, ln.hi, stats, qu, qv, k, v, pos, pe, bias, lse, pa, sa, ctx, po, so, B, T))
        return y
    content = hip.gemm(qu, k, M=T, N=T, K=dh, lda=d, ldb=ldk, nbatch=nbh, batch_inner=H, sA=(T * d, dh), sB=(T * ldk, dh),
                       out_dtype=torch.float32, out_shape=(B, H, T, T))
    pscore = hip.gemm(qv, pos, M=T, N=T, K=dh, lda=d, ldb=d, nbatch=nbh, batch_inner=H, sA=(T * d, dh), sB=(0, dh),
                      out_dtype=torch.float32, out_shape=(B, H, T, T))
    if replay:
        p, pd = hip.softmax_relshift_fwd(content, pscore, scale, RT.dtype, 0.0, 0)
        if pa > 0:
            sa = RT.replay.mask((B, H, T, T), pa, x.device, RT.dtype)
            pd = p * sa
    else:
        p, pd = hip.softmax_relshift_fwd(content, pscore, scale, RT.dtype, pa, sa)
    del content, pscore
    ctx = torch.empty((M, d), dtype=RT.dtype, device=x.device)
    hip.gemm(pd, v, a_kc=True, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=ldk, nbatch=nbh, batch_inner=H,
             sA=(H * T * T, T * T), sB=(T * ldk, dh), out=ctx, ldc=d, sC=(T * d, dh))
    if replay:
        y = mm_nt_h(ctx, wo, _F32, bias=att.out_proj.linear.bias.data)
        so = RT.replay.mask(tuple(y.shape), po, x.device, _F32) if po > 0 else 0
        y = x + (y * so if po > 0 else y)
    else:
        so = RT.next_seed() if po > 0 else 0
        y = mm_nt_h(ctx, wo, _F32, bias=att.out_proj.linear.bias.data, p_drop=po, seed=so, resid=x, ldr=x.stride(0), res_scale=1.0)
    saved.append((x, ln.hi, stats, qu, qv, k, v, pos, pe, p, pd, pa, sa, ctx, po, so, B, T))
    return y


def mhsa_fwd(x, mod, B, T, train, saved):
    """x + MultiHeadedSelfAttentionModule(x)  (conformer/attention.py:143-151, 72-113).

    bf16: the positional-score GEMM writes (q+v)P^T directly in the reference's relative-shift layout and ONE flash-style kernel
    (csrc/attention.hip) does content score + shifted positional score + 1/sqrt(d_model) + softmax + dropout + PV - no (B,H,T,T)
    score / probability tensor.  f32 mode, replayed dropout masks and shapes the fused kernel does not take: batched MFMA GEMMs over
    (b, head) + one scale/relative-shift/softmax/dropout kernel.  The positional projection is computed once per call (it is
    batch-invariant)."""
    att = mod.attention
    H, dh, d = att.num_heads, att.d_head, att.d_model
    M = B * T
    if RT.hybrid:
        return _mhsa_fwd_h(x, mod, B, T, train, saved)
    fused = _qkv_views(att)
    lin = (_LIN256 and fused is not None and d == 256 and not _replaying(train) and x.stride(1) == 1
           and hip.lin256_supported(M, 3 * d, d, x.dtype))
    pk = _lin256_pack(mod, "qkv", need_bwd=not RT.inference) if lin else None
    if pk is not None:      # LayerNorm + the [M, 3d] projection in one tile-resident launch (csrc/lin256.hip)
        qkv, ln, stats = hip.lin256_fwd(None, pk[0], fused[1], 3 * d, d, ln_in=(x, mod.layer_norm.weight.data, mod.layer_norm.bias.data, mod.layer_norm.eps))
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    else:
        ln, stats = hip.layernorm_fwd(x, mod.layer_norm.weight.data, mod.layer_norm.bias.data, mod.layer_norm.eps)
    if pk is not None:
        pass
    elif fused is not None:                     # one [M, 3d] GEMM; q / k / v are column slices (row stride 3d)
        qkv = mm_nt(ln, fused[0], bias=fused[1])
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    else:
        q = mm_nt(ln, wt(att.query_proj.linear.weight), bias=att.query_proj.linear.bias.data)
        k = mm_nt(ln, wt(att.key_proj.linear.weight), bias=att.key_proj.linear.bias.data)
        v = mm_nt(ln, wt(att.value_proj.linear.weight), bias=att.value_proj.linear.bias.data)
    ldk = k.stride(0)
    pe = _pe(mod, T)
    pos = _pos_proj(mod, T)                                                                  # [T, d]
    nbh = B * H
    pa = _p(att.dropout, train)
    sa = RT.next_seed() if pa > 0 else 0
    scale = 1.0 / math.sqrt(d)                                                               # 1/sqrt(d_model), attention.py:57
    replay = _replaying(train) and (pa > 0 or _p(mod.dropout, train) > 0)
    fused_attn = _FUSED_ATTN and not replay and hip.relpos_attn_supported(T, dh, RT.dtype)
    in_kernel = fused_attn and _ATTN_POS and hip.relpos_attn_pos_supported(T, dh, RT.dtype)
    ub, vb = att.u_bias.data.view(-1), att.v_bias.data.view(-1)
    # (positional score in the kernels: they also form q + u / q + v while loading the query rows - no biased copies of q)
    qu, qv = (q, None) if in_kernel else hip.bias2(q, ub, vb)
    if fused_attn:
        # fused path (csrc/attention.hip): the positional-score GEMM writes its product directly in the relative-shift layout and
        # one flash-style kernel does content score + shifted bias + softmax + dropout + PV; no (B,H,T,T) score / probability tensor
        if in_kernel:
            # T <= 256: the kernel forms the shifted positional score itself (position tiles stream through its K buffer)
            ctx, lse, bias = hip.relpos_attn_fwd_pos(q, q, k, v, pos, B, H, T, dh, scale, pa, sa, need_bwd=not RT.inference,
                                                     biases=(ub, vb))
        elif _ATTN_POS_LONG and hip.relpos_attn_pos_long_supported(T, dh, RT.dtype):
            # T > 256 (config 5): the forward kernel forms the shifted score itself, one 256-key block of the slab at a time, and writes
            # it out for the backward kernels - no positional-score GEMM launch, no read of its result
            ctx, lse, bias = hip.relpos_attn_fwd_pos_long(qu, qv, k, v, pos, B, H, T, dh, scale, pa, sa, need_bwd=not RT.inference)
        else:
            bias = torch.empty((B, H, T, T), dtype=RT.dtype, device=x.device)
            hip.gemm(qv, pos, M=T, N=T, K=dh, lda=d, ldb=d, nbatch=nbh, batch_inner=H, sA=(T * d, dh), sB=(0, dh), out=bias, ldc=T,
                     sC=(H * T * T, T * T), c_row_shift=True)
            ctx, lse = hip.relpos_attn_fwd(qu, k, v, bias, B, H, T, dh, scale, pa, sa, need_bwd=not RT.inference)   # lse = (ctx32, lse)
        po = _p(mod.dropout, train)
        so = RT.next_seed() if po > 0 else 0
        y = mm_nt(ctx, wt(att.out_proj.linear.weight), bias=att.out_proj.linear.bias.data, p_drop=po, seed=so,
                  resid=x, ldr=x.stride(0), res_scale=1.0)
        saved.append((x, ln, stats, qu, qv, k, v, pos, pe, bias, lse, pa, sa, ctx, po, so, B, T))
        return y
    content = hip.gemm(qu, k, M=T, N=T, K=dh, lda=d, ldb=ldk, nbatch=nbh, batch_inner=H, sA=(T * d, dh), sB=(T * ldk, dh),
                       out_dtype=torch.float32, precise=RT.precise, out_shape=(B, H, T, T))
    pscore = hip.gemm(qv, pos, M=T, N=T, K=dh, lda=d, ldb=d, nbatch=nbh, batch_inner=H, sA=(T * d, dh), sB=(0, dh),
                      out_dtype=torch.float32, precise=RT.precise, out_shape=(B, H, T, T))
    if replay:                                             # host-drawn masks: attention probabilities, then the module output
        p, pd = hip.softmax_relshift_fwd(content, pscore, scale, RT.dtype, 0.0, 0)
        if pa > 0:
            sa = RT.replay.mask((B, H, T, T), pa, x.device, RT.dtype)
            pd = p * sa
    else:
        p, pd = hip.softmax_relshift_fwd(content, pscore, scale, RT.dtype, pa, sa)
    del content, pscore
    ctx = torch.empty((M, d), dtype=RT.dtype, device=x.device)
    hip.gemm(pd, v, a_kc=True, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=ldk, nbatch=nbh, batch_inner=H,
             sA=(H * T * T, T * T), sB=(T * ldk, dh), out=ctx, ldc=d, sC=(T * d, dh), precise=RT.precise)
    po = _p(mod.dropout, train)
    if replay:
        y = mm_nt(ctx, wt(att.out_proj.linear.weight), bias=att.out_proj.linear.bias.data)
        so = RT.replay.mask(tuple(y.shape), po, x.device, RT.dtype) if po > 0 else 0
        y = x + (y * so if po > 0 else y)
    else:
        so = RT.next_seed() if po > 0 else 0
        y = mm_nt(ctx, wt(att.out_proj.linear.weight), bias=att.out_proj.linear.bias.data, p_drop=po, seed=so,
                  resid=x, ldr=x.stride(0), res_scale=1.0)
    saved.append((x, ln, stats, qu, qv, k, v, pos, pe, p, pd, pa, sa, ctx, po, so, B, T))
    return y


def mhsa_bwd(dy, mod, saved, dy_dropped=None, next_kind=None):
    x, ln, stats, qu, qv, k, v, pos, pe, p, pd, pa, sa, ctx, po, so, B, T = saved.pop()
    att = mod.attention
    H, dh, d = att.num_heads, att.d_head, att.d_model
    M, nbh = B * T, B * H
    dev = x.device
    fused_attn = isinstance(pd, tuple)                                # fused forward saved (bias, (ctx32, lse)) in place of (p, pd)
    if RT.hybrid:                                          # dy: f32 stream gradient, dy_dropped: its bf16 operand copy
        dout = _grad16(dy, dy_dropped, po, so)
        if torch.is_tensor(so):
            dout = dout * so.to(dout.dtype)
    elif torch.is_tensor(so):
        dout = dy * so.to(dy.dtype)                        # replayed mask
    elif dy_dropped is not None:
        dout = dy_dropped
    else:
        dout = hip.act_bwd(dy, None, 0, p_drop=po, seed=so) if po > 0 else dy
    mm_tn_acc(dout, ctx, gbuf(att.out_proj.linear.weight), bias=gbuf(att.out_proj.linear.bias))
    dctx = mm_nn(dout, wtg(att.out_proj.linear.weight))
    fused = _qkv_views(att, grad=True)
    ldk = k.stride(0)
    if fused is not None:                       # dq | dk | dv are written straight into one [M, 3d] buffer
        dqkv = torch.empty((M, 3 * d), dtype=RT.gdtype, device=dev)
        dqu, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        if fused_attn:      # the content-score part of dq in its own buffer: its column sum (u_bias gradient) can then wait for the
            dqu = torch.empty((M, d), dtype=RT.gdtype, device=dev)    # block's batched launch instead of running before dq overwrites it
    else:
        dqu = torch.empty((M, d), dtype=RT.gdtype, device=dev)
        dk = torch.empty((M, d), dtype=RT.gdtype, device=dev)
        dv = torch.empty((M, d), dtype=RT.gdtype, device=dev)
    ldg = dqu.stride(0)
    scale = 1.0 / math.sqrt(d)
    if fused_attn and _ATTN_POS and hip.relpos_attn_pos_supported(T, dh, RT.dtype):
        # T <= 256: the dQ kernel also forms the positional-score gradients (no d(bias) tensor, un-shift pass or batched products)
        dqv = torch.empty((M, d), dtype=RT.gdtype, device=dev)
        plain_q = qv is None                       # forward saved the plain query projection: the kernels add u / v while loading
        dq_out = dqkv[:, :d] if fused is not None else None      # dq = dqu + dqv written by the dK / dV kernel (needs a buffer of its own)
        dposb = hip.relpos_attn_bwd_pos(qu, qu if plain_q else qv, k, v, pos, p, pd, dctx, dqu, dqv, dk, dv, B, H, T, dh, scale, pa, sa,
                                        biases=(att.u_bias.data.view(-1), att.v_bias.data.view(-1)) if plain_q else None, dq_sum=dq_out)
        return _mhsa_bwd_tail(dy, mod, att, x, ln, stats, pe, dqkv if fused is not None else None, dqu, dk, dv, dqv, dposb, fused, B, T, d,
                              dev, drop=_next_drop(next_kind, saved), dq_out=dq_out if dq_out is not None else dqu,
                              dq_done=dq_out is not None, want16=next_kind is not None)
    if fused_attn:
        dbias = hip.relpos_attn_bwd(qu, k, v, p, pd, dctx, dqu, dk, dv, B, H, T, dh, scale, pa, sa)      # p = bias, pd = (ctx32, lse) here
        dps = hip.relshift_bwd(dbias)                                                        # d (unshifted) pos score
        del dbias
    else:
        dps = _mhsa_bwd_scores(dctx, qu, k, v, p, pd, dqu, dk, dv, B, T, H, dh, d, ldk, ldg, scale, pa, sa)
    nbh = B * H
    dqv = torch.empty((M, d), dtype=RT.gdtype, device=dev)
    hip.gemm(dps, pos, a_kc=True, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=nbh, batch_inner=H,
             sA=(H * T * T, T * T), sB=(0, dh), out=dqv, ldc=d, sC=(T * d, dh), precise=RT.precise)
    # d pos (per batch item, then summed over the batch): dpos[b][m][h,:] = sum_i dps[b,h,i,m] * qv[b,i,h,:]
    dposb = torch.empty((B, T, d), dtype=RT.gdtype, device=dev)
    hip.gemm(dps, qv, a_kc=False, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=nbh, batch_inner=H,
             sA=(H * T * T, T * T), sB=(T * d, dh), out=dposb, ldc=d, sC=(T * d, dh), precise=RT.precise)
    del dps
    return _mhsa_bwd_tail(dy, mod, att, x, ln, stats, pe, dqkv if fused is not None else None, dqu, dk, dv, dqv, dposb, fused, B, T, d, dev,
                          drop=_next_drop(next_kind, saved), dq_out=dqkv[:, :d] if fused is not None else dqu, want16=next_kind is not None)


def _mhsa_bwd_scores(dctx, qu, k, v, p, pd, dqu, dk, dv, B, T, H, dh, d, ldk, ldg, scale, pa, sa):
    """Unfused attention-core backward (f32 mode / shapes the fused kernel does not take): dv, dqu, dk from materialised
    probabilities; returns the gradient of the unshifted positional score."""
    nbh = B * H
    # dP = dctx @ v^T ; dv = P^T @ dctx
    dpd = hip.gemm(dctx, v, M=T, N=T, K=dh, lda=d, ldb=ldk, nbatch=nbh, batch_inner=H, sA=(T * d, dh), sB=(T * ldk, dh),
                   out_dtype=torch.float32, precise=RT.precise, out_shape=(B, H, T, T))
    hip.gemm(pd, dctx, a_kc=False, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=nbh, batch_inner=H,
             sA=(H * T * T, T * T), sB=(T * d, dh), out=dv, ldc=ldg, sC=(T * ldg, dh), precise=RT.precise)
    if torch.is_tensor(sa):                                # replayed mask on the probabilities
        ds = hip.softmax_bwd(dpd * sa.float(), p, scale, 0.0, 0)
    else:
        ds = hip.softmax_bwd(dpd, p, scale, pa, sa)                                          # d content score
    del dpd
    dps = hip.relshift_bwd(ds)                                                               # d (unshifted) pos score
    hip.gemm(ds, k, a_kc=True, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=ldk, nbatch=nbh, batch_inner=H,
             sA=(H * T * T, T * T), sB=(T * ldk, dh), out=dqu, ldc=ldg, sC=(T * ldg, dh), precise=RT.precise)
    hip.gemm(ds, qu, a_kc=False, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=nbh, batch_inner=H,
             sA=(H * T * T, T * T), sB=(T * d, dh), out=dk, ldc=ldg, sC=(T * ldg, dh), precise=RT.precise)
    return dps


def _mhsa_bwd_tail(dy, mod, att, x, ln, stats, pe, dqkv, dqu, dk, dv, dqv, dposb, fused, B, T, d, dev, drop=None, dq_out=None,
                   dq_done=False, want16=False):
    """Positional-projection, bias and q/k/v-projection gradients + LayerNorm backward (shared by both attention cores)."""
    if dposb.dtype == RT.gdtype:                                     # batch sum straight into the GEMM operand's dtype: one launch
        dpos_rt = hip.colsum_store(dposb.view(dposb.shape[0], T * d)).view(T, d)      # (B rows, or B * ntile partials of the fused backward)
    else:
        dpos = torch.zeros((T * d,), dtype=torch.float32, device=dev)
        hip.colsum(dposb.view(B, T * d), dpos, now=True)             # consumed right below
        dpos_rt = to_g(dpos.view(T, d))
    mm_tn_acc(dpos_rt, pe, gbuf(att.pos_proj.linear.weight))
    if dq_out is None:
        dq_out = dqu
    inplace = dq_out.data_ptr() == dqu.data_ptr()                    # dq = dqu + dqv overwrites dqu: its column sum has to run first
    hip.colsum(dqu, gbuf(att.u_bias).view(-1), now=inplace)
    hip.colsum(dqv, gbuf(att.v_bias).view(-1))
    dq = dq_out if dq_done else hip.axpby2d(dqu, dqv, 1.0, 1.0, out=dq_out)      # (dq_done: the attention backward wrote dqu + dqv itself)
    if fused is not None:
        mm_tn_acc(dqkv, ln, fused[2], bias=fused[3])
        pk = mod.__dict__.get("_lin256_qkv")
        if (_LIN256 and pk is not None and pk[0][3] and pk[0][0] == weights_version() and d == 256 and dqkv.dtype in _16 and x.dtype == ln.dtype
                and x.stride(1) == 1 and dy.stride(1) == 1 and hip.lin256_supported(dqkv.shape[0], d, 3 * d, dqkv.dtype)):
            # data gradient of the projection + the LayerNorm backward in one tile-resident launch
            return hip.lin256_bwd(dqkv, pk[1][1], d, 3 * d, ln_bwd=(x, mod.layer_norm.weight.data, stats, dy, gbuf(mod.layer_norm.weight),
                                                                     gbuf(mod.layer_norm.bias), drop))
        dln = mm_nn(dqkv, fused[0], out_dtype=_F32 if (RT.hybrid and _H_DLN32) else None)
    else:
        for proj, g in ((att.query_proj, dq), (att.key_proj, dk), (att.value_proj, dv)):
            mm_tn_acc(g, ln, gbuf(proj.linear.weight), bias=gbuf(proj.linear.bias))
        dln = mm_nn(dq, wtg(att.query_proj.linear.weight))
        dln = mm_nn(dk, wtg(att.key_proj.linear.weight), out=dln, ldc=d, resid=dln, ldr=d, res_scale=1.0)
        dln = mm_nn(dv, wtg(att.value_proj.linear.weight), out=dln, ldc=d, resid=dln, ldr=d, res_scale=1.0)
    if RT.hybrid:
        return _ln_bwd_hd(dln, x, mod.layer_norm, stats, _as_stream(dy), drop, want16)
    return hip.layernorm_bwd(dln, x, mod.layer_norm.weight.data, stats, resid=dy, dgamma=gbuf(mod.layer_norm.weight),
                             dbeta=gbuf(mod.layer_norm.bias), drop=drop)


def convmod_fwd(x, cm, B, T, train, saved):
    """x + ConformerConvModule(x)  (conformer/convolution.py:136-149)."""
    seq = cm.sequential
    d = x.shape[1]
    pw1, dw, bn, pw2 = seq[2].conv, seq[4].conv, seq[5], seq[7].conv
    hyb = RT.hybrid
    if hyb:
        x = _as_stream(x)
    lin = (_LIN256 and not hyb and d == 256 and not _replaying(train) and x.stride(1) == 1 and hip.lin256_supported(x.shape[0], 2 * d, d, x.dtype))
    pk = _lin256_pack(cm, "pw1", need_bwd=not RT.inference) if lin else None
    if hyb:                 # f32 stream in / out; the module's inner tensors (h, c, s) are the fp16 tensors of the fp16 mode
        lnp, stats = hip.layernorm_fwd_pair(x, seq[0].weight.data, seq[0].bias.data, seq[0].eps, want_lo="pw1" in _H_ALO)
        ln = lnp.hi
        h = mm_nt_h(lnp, wpair(pw1.weight, (2 * d, d)), torch.float16, bias=pw1.bias.data)
    elif pk is not None:      # LayerNorm + first pointwise convolution in one tile-resident launch (csrc/lin256.hip)
        h, ln, stats = hip.lin256_fwd(None, pk[0], pw1.bias.data, 2 * d, d, ln_in=(x, seq[0].weight.data, seq[0].bias.data, seq[0].eps))
    else:
        ln, stats = hip.layernorm_fwd(x, seq[0].weight.data, seq[0].bias.data, seq[0].eps)
        h = mm_nt(ln, wt(pw1.weight).view(2 * d, d), bias=pw1.bias.data)                     # [M, 2d]
    g = None
    if _DWGLU and d % 8 == 0:            # GLU + depthwise conv + BatchNorm batch sums in one LDS-tiled pass (csrc/dwconv.hip)
        c, sums = hip.dwglu_fwd(h, dw.weight.data.view(d, -1), B, T, want_stats=True) if train else \
            (hip.dwglu_fwd(h, dw.weight.data.view(d, -1), B, T), None)
        if train and d >= 64 and 256 % (d // 8) == 0:       # BatchNorm finalize + affine + Swish in one launch (bit-identical to the two)
            aff, s_act = hip.cl_bn_train_act(c, d, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                             sums, SWISH, eps=bn.eps, momentum=bn.momentum)
        else:
            aff, s_act = bn_affine(c, d, bn, train, sums=sums), None
    else:
        g = hip.glu_fwd(h)
        c = hip.dwconv(g.view(B, T, d), dw.weight.data.view(d, -1))
        aff, s_act = bn_affine(c, d, bn, train), None
    s = (s_act if s_act is not None else hip.cl_affine_act(c, d, aff, SWISH)).view(B * T, d)
    po = _p(seq[8], train)
    if _replaying(train) and po > 0:                       # the reference draws this mask on the (B, d, T) conv output
        y = mm_nt_h(s, wpair(pw2.weight, (d, d)), _F32, bias=pw2.bias.data) if hyb else mm_nt(s, wt(pw2.weight).view(d, d), bias=pw2.bias.data)
        so = RT.replay.mask((B, d, T), po, x.device, y.dtype, to_layout=lambda m: m.permute(0, 2, 1).reshape(B * T, d))
        y = x + y * so
    elif hyb:
        so = RT.next_seed() if po > 0 else 0
        y = mm_nt_h(s, wpair(pw2.weight, (d, d)), _F32, bias=pw2.bias.data, p_drop=po, seed=so, resid=x, ldr=x.stride(0), res_scale=1.0)
    else:
        so = RT.next_seed() if po > 0 else 0
        y = mm_nt(s, wt(pw2.weight).view(d, d), bias=pw2.bias.data, p_drop=po, seed=so, resid=x, ldr=x.stride(0), res_scale=1.0)
    saved.append((x, ln, stats, h, g, c, aff, s, po, so, B, T, train))
    return y


def convmod_bwd(dy, cm, saved, dy_dropped=None, next_kind=None):
    x, ln, stats, h, g, c, aff, s, po, so, B, T, train = saved.pop()
    seq = cm.sequential
    d = x.shape[1]
    pw1, dw, bn, pw2 = seq[2].conv, seq[4].conv, seq[5], seq[7].conv
    if RT.hybrid:
        dout = _grad16(dy, dy_dropped, po, so)
        if torch.is_tensor(so):
            dout = dout * so.to(dout.dtype)
    elif torch.is_tensor(so):
        dout = dy * so.to(dy.dtype)
    elif dy_dropped is not None:
        dout = dy_dropped
    else:
        dout = hip.act_bwd(dy, None, 0, p_drop=po, seed=so) if po > 0 else dy
    mm_tn_acc(dout, s, gbuf(pw2.weight), bias=gbuf(pw2.bias))
    ds = mm_nn(dout, wtg(pw2.weight).view(d, d))
    red = hip.cl_bn_bwd_reduce(ds, c, d, aff, SWISH)
    dc = hip.cl_bn_bwd_apply(ds, c, d, aff, SWISH, False, train, red, out=ds, pgrads=(gbuf(bn.weight), gbuf(bn.bias))).view(B, T, d)
    if g is None:                        # fused forward: the GLU output was never stored
        dh = hip.dwglu_bwd(dc.view(B * T, d), h, dw.weight.data.view(d, -1), B, T)
        hip.dwglu_wgrad(dc.view(B * T, d), h, gbuf(dw.weight).view(d, -1), B, T)
    else:
        dg = hip.dwconv(dc, dw.weight.data.view(d, -1), flip=True)
        hip.dwconv_wgrad(dc, g.view(B, T, d), gbuf(dw.weight).view(d, -1))
        dh = hip.glu_bwd(dg.view(B * T, d), h)
    mm_tn_acc(dh, ln, gbuf(pw1.weight), bias=gbuf(pw1.bias))
    pk = cm.__dict__.get("_lin256_pw1")
    if (_LIN256 and pk is not None and pk[0][3] and pk[0][0] == weights_version() and d == 256 and dh.dtype in _16 and x.dtype == ln.dtype
            and x.stride(1) == 1 and dy.stride(1) == 1 and dh.stride(1) == 1 and hip.lin256_supported(dh.shape[0], d, 2 * d, dh.dtype)):
        return hip.lin256_bwd(dh, pk[1][1], d, 2 * d, ln_bwd=(x, seq[0].weight.data, stats, dy, gbuf(seq[0].weight), gbuf(seq[0].bias),
                                                               _next_drop(next_kind, saved)))
    if RT.hybrid:
        dln = mm_nn(dh, wtg(pw1.weight).view(2 * d, d), out_dtype=_F32 if _H_DLN32 else None)
        return _ln_bwd_h(dln, x, seq[0], stats, _as_stream(dy), saved, next_kind, next_kind is not None)
    dln = mm_nn(dh, wtg(pw1.weight).view(2 * d, d))
    return hip.layernorm_bwd(dln, x, seq[0].weight.data, stats, resid=dy, dgamma=gbuf(seq[0].weight), dbeta=gbuf(seq[0].bias),
                             drop=_next_drop(next_kind, saved))


_LN_PAIR = os.environ.get("SARSSL_LN_PAIR", "1") != "0"       # 0: the two LayerNorms of a block boundary as two launches (A/B runs)


def block_fwd(x, blk, B, T, train, saved, out=None, next_blk=None, rows=None):
    """ConformerBlock (code/common/Conformer.py:59-91).  next_blk: the block that consumes the result - its first LayerNorm (the
    feed-forward module's) is applied by this block's closing LayerNorm launch (two row passes in one kernel, bit-identical).
    rows = idx (int32 [B, nm], ascending): the caller only consumes the rows of these frames (the LAST block of an encoder in a training
    step: the decoder runs on the masked frames) - the second feed-forward module and the closing LayerNorm act on every row separately,
    so they run on the gathered rows only and the result is [B * nm, d]; everything in front of them mixes frames (attention, depthwise
    convolution, BatchNorm statistics) and sees every row."""
    seq = blk.sequential
    x = ffn_fwd(x, seq[0].module, seq[0].module_factor, train, saved)
    x = mhsa_fwd(x, seq[1].module, B, T, train, saved)
    x = convmod_fwd(x, seq[2].module, B, T, train, saved)
    if rows is not None:
        xc = hip.gather_rows(x, rows, B, T)
        xc = ffn_fwd(xc, seq[3].module, seq[3].module_factor, train, saved)
        if isinstance(out, tuple):         # hybrid mode: the decoder's input leaves the closing LayerNorm as the fp16 pair its first product reads
            y, stats = hip.layernorm_fwd_pair(_as_stream(xc), seq[4].weight.data, seq[4].bias.data, seq[4].eps, out=out)
        else:
            y, stats = hip.layernorm_fwd(xc, seq[4].weight.data, seq[4].bias.data, seq[4].eps, out=out)
        saved.append((xc, stats, (rows, B, T), x))          # (x: the full-row input of the tail, for a full prediction on request - vis)
        return y
    x = ffn_fwd(x, seq[3].module, seq[3].module_factor, train, saved)
    if next_blk is not None and out is None and _LN_PAIR and not (RT.hybrid and _FFN2 and _H_FFN2_FWD and x.shape[1] in _FFN2_FWD and not _replaying(train)
                                                               and hip.ffn2h_supported(x.shape[0], x.shape[1])):
        # (hybrid, d = 256: the next block's fused feed-forward launch normalises its rows in its own prologue)
        nln = next_blk.sequential[0].module.sequential[0]
        if RT.hybrid:
            y, stats, z, zstats = hip.layernorm_fwd2_pair(x, seq[4].weight.data, seq[4].bias.data, seq[4].eps, nln.weight.data, nln.bias.data, nln.eps,
                                                          want_lo="ffn1" in _H_ALO)
        else:
            y, stats, z, zstats = hip.layernorm_fwd2(x, seq[4].weight.data, seq[4].bias.data, seq[4].eps, nln.weight.data, nln.bias.data, nln.eps)
        # the paired LayerNorm's result travels ON the tensor object it belongs to (round 4 kept it in a process-global dict keyed by the
        # tensor's address: a stale entry could outlive its tensor and match a recycled address - advisor): whoever consumes y as the next
        # block's input finds it, anything else never sees it, and it dies with y
        y._pre_ln = (nln, z, zstats)
    else:
        y, stats = hip.layernorm_fwd(x, seq[4].weight.data, seq[4].bias.data, seq[4].eps, out=out)
    saved.append((x, stats))
    return y


def block_bwd(dy, blk, saved, first=False):
    """first (hybrid mode): ``blk`` is the encoder's first block - its input gradient goes to the frame-patch GEMMs, which take it in bf16;
    the block's last LayerNorm backward writes that copy next to the f32 gradient (attribute ``_g16``) instead of a cast pass."""
    seq = blk.sequential
    top = saved.pop()
    x, stats = top[0], top[1]
    rows = top[2] if len(top) > 2 else None        # compact tail (block_fwd(rows=...)): dy, x are [B * nm, d]
    # the block's ~9 bias-gradient column sums and the reductions of its 9 split-K weight-gradient products: one launch each, at the end
    with hip.colsum_batched(), hip.splitk_batched(), hip.ln_reduce_batched(), wgrad_block():
        # each module's backward starts with the dropout backward of its incoming gradient: the LayerNorm backward that produces
        # that gradient writes the dropped copy as a second output (d = (gradient, dropped gradient) where a mask applies)
        pair = lambda r: r if isinstance(r, tuple) else (r, None)
        if RT.hybrid:       # f32 stream gradient (d) + the bf16 operand copy of the module that consumes it next (dd)
            d, dd = _ln_bwd_h(_as_stream(dy), x, seq[4], stats, None, saved, "ffn", True)
        else:
            d, dd = pair(hip.layernorm_bwd(dy, x, seq[4].weight.data, stats, dgamma=gbuf(seq[4].weight), dbeta=gbuf(seq[4].bias),
                                           drop=_next_drop("ffn", saved)))
        if rows is not None:    # the tail ran on the gathered rows: its input gradient goes back to its frames, zeros elsewhere
            d = pair(ffn_bwd(d, seq[3].module, saved, dy_dropped=dd, next_kind=None))[0]
            drop = _next_drop("conv", saved) if (RT.hybrid and d.dtype == _F32) else None
            if drop is not None:       # the scatter also writes the convolution module's bf16 operand (its dropout backward applied)
                d, dd = hip.scatter_rows_drop16(d.contiguous(), rows[0], rows[1], rows[2], *drop)
                dd._dropped = True
            else:
                d, dd = hip.scatter_rows(d, rows[0], rows[1], rows[2]), None
        else:
            d, dd = pair(ffn_bwd(d, seq[3].module, saved, dy_dropped=dd, next_kind="conv"))
        d, dd = pair(convmod_bwd(d, seq[2].module, saved, dy_dropped=dd, next_kind="mhsa"))
        d, dd = pair(mhsa_bwd(d, seq[1].module, saved, dy_dropped=dd, next_kind="ffn"))
        if first and RT.hybrid:
            d, d16 = pair(ffn_bwd(d, seq[0].module, saved, dy_dropped=dd, next_kind="copy"))
            if d16 is not None:
                d._g16 = d16
        else:
            d = pair(ffn_bwd(d, seq[0].module, saved, dy_dropped=dd))[0]
    return d


def block_tail_full(x_full, blk, train, out=None):
    """Second feed-forward module + closing LayerNorm of ``blk`` on EVERY row of its input (no tensors saved): what block_fwd(rows=...)
    skipped for the frames nobody consumed - for a full prediction on request (vis).  Dropout (train mode) draws fresh masks."""
    seq = blk.sequential
    y = ffn_fwd(x_full, seq[3].module, seq[3].module_factor, train, [])
    return hip.layernorm_fwd(y, seq[4].weight.data, seq[4].bias.data, seq[4].eps, out=out, save=False)[0]


def encoder_fwd(x, enc, B, T, train, saved, out=None):
    """ConformerEncoder.forward, add_same_one=False (code/common/Conformer.py:165-195)."""
    n = len(enc.layers)
    for i, blk in enumerate(enc.layers):
        x = block_fwd(x, blk, B, T, train, saved, out=out if i == n - 1 else None, next_blk=enc.layers[i + 1] if i + 1 < n else None)
    return x


def encoder_bwd(dy, enc, saved):
    for blk in reversed(enc.layers):
        dy = block_bwd(dy, blk, saved, first=blk is enc.layers[0])
    return dy


# ------------------------------------------------------------------------------------------------ decoder + loss
def decoder_fwd(e, dec, saved):
    """EmbedDecoder ['','fc'] (code/model.py:295-301, 321-334): Linear -> ReLU -> Linear."""
    l1, l2 = dec.proj[0], dec.proj[2]
    if RT.hybrid:           # f32 decoder input as an fp16 pair, fp16 hidden layer, f32 prediction
        ep = e if isinstance(e, hip.Pair) else hip.split_pair(_as_stream(e).contiguous())
        h = mm_nt_h(ep if "dec1" in _H_ALO else ep.hi, wpair(l1.weight), torch.float16, bias=l1.bias.data, act=RELU)
        pred = mm_nt_h(h, wpair(l2.weight), _F32, bias=l2.bias.data)
        saved.append((ep.hi, h))
        return pred
    h = mm_nt(e, wt(l1.weight), fp8=False, bias=l1.bias.data, act=RELU)
    pred = mm_nt(h, wt(l2.weight), fp8=False, bias=l2.bias.data)
    saved.append((e, h))
    return pred


def decoder_bwd(dpred, dec, saved):
    e, h = saved.pop()
    l1, l2 = dec.proj[0], dec.proj[2]
    with hip.colsum_batched(), hip.splitk_batched(), wgrad_block():
        mm_tn_acc(dpred, h, gbuf(l2.weight), bias=gbuf(l2.bias))
        dh = mm_nn(dpred, wtg(l2.weight), fp8=False, aux=h, aux_act=RELU)
        mm_tn_acc(dh, e, gbuf(l1.weight), bias=gbuf(l1.bias))
        return mm_nn(dh, wtg(l1.weight), fp8=False, out_dtype=_F32 if RT.hybrid else None)      # hybrid: the stream's gradient is f32


# ------------------------------------------------------------------------------------------------ downstream heads
def _head_layers(seq):
    """nn.Sequential(LayerNorm, Linear) or (LayerNorm, Linear, ReLU, Linear) (code/model.py:411-419, 806-808) -> (LayerNorm, [Linear, ..])."""
    mods = list(seq)
    assert isinstance(mods[0], torch.nn.LayerNorm) and all(isinstance(m, (torch.nn.Linear, torch.nn.ReLU)) for m in mods[1:]), "unsupported head"
    return mods[0], [m for m in mods[1:] if isinstance(m, torch.nn.Linear)]


def head_fwd(x, seq, saved):
    """Downstream head on pooled embeddings x f32 [B, d]: LayerNorm + Linear (+ ReLU + Linear), f32 through csrc/head.hip."""
    ln_mod, lins = _head_layers(seq)
    h, stats = hip.layernorm_fwd(x, ln_mod.weight.data, ln_mod.bias.data, ln_mod.eps)
    acts = []
    for i, lin in enumerate(lins):
        act = 1 if i + 1 < len(lins) else 0
        y = hip.small_linear_fwd(h, lin.weight.data.contiguous(), lin.bias.data if lin.bias is not None else None, act)
        acts.append((h, y, act))
        h = y
    saved.append((x, stats, acts))
    return h


def head_bwd(dy, seq, saved):
    ln_mod, lins = _head_layers(seq)
    x, stats, acts = saved.pop()
    for lin, (h, y, act) in reversed(list(zip(lins, acts))):
        dy = hip.small_linear_bwd(dy, y, h, lin.weight.data.contiguous(), act, gbuf(lin.weight), gbuf(lin.bias) if lin.bias is not None else None)
    return hip.layernorm_bwd(dy, x, ln_mod.weight.data, stats, dgamma=gbuf(ln_mod.weight), dbeta=gbuf(ln_mod.bias))
