"""Bridge between torch.autograd and the hand-written forward/backward pairs in engine.py.

One autograd node per module call: forward runs the HIP forward and keeps a list of saved tensors, backward
runs the hand-written HIP backward, which accumulates parameter gradients directly into ``p.grad`` (so the
parameters are passed to ``apply`` only to make the output require grad; their returned gradients are None).
"""
import torch

from . import hip
from .runtime import RT, begin_forward


def _to_rt(x):
    return x if x.dtype == RT.dtype else hip.cast(x.contiguous(), RT.dtype)


def _to_g(dy):
    """incoming gradient in the backward pass's storage dtype (bf16 next to fp16 activations)"""
    return dy if dy.dtype == RT.gdtype else hip.cast(dy.contiguous(), RT.gdtype)


class _TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fwd, bwd, x, *params):
        if not x.is_cuda:
            raise hip._lib.SarsslHipError("sar_ssl_amd modules run on the GPU only (no CPU fallback); got a CPU tensor")
        saved = []
        y = fwd(_to_rt(x.detach().contiguous()), saved)
        ctx.bwd, ctx.saved, ctx.in_dtype, ctx.nparams = bwd, saved, x.dtype, len(params)
        if y.dtype == torch.float16:
            # node boundary of the fp16-forward mode: torch.autograd hands a node the gradient in its OUTPUT's dtype, and a gradient
            # must never travel as fp16 (1e-7-sized values: below fp16's range, there is no loss scaling) - the output leaves as f32
            # (exact), the next node's forward re-encodes it (exact round trip), gradients between nodes are f32
            y = hip.cast(y.contiguous(), torch.float32)
        return y

    @staticmethod
    def backward(ctx, dy):
        hip.sums_arena_reset(dy.device)
        dx = ctx.bwd(_to_g(dy.contiguous()), ctx.saved)
        if dx is not None and dx.dtype != ctx.in_dtype:
            dx = hip.cast(dx.contiguous(), ctx.in_dtype)
        return (None, None, dx) + (None,) * ctx.nparams


def tape_apply(module, fwd, bwd, x):
    begin_forward(module.parameters())
    params = [p for p in module.parameters() if p.requires_grad]
    if torch.is_grad_enabled() and (x.requires_grad or params):
        return _TapeFn.apply(fwd, bwd, x, *params)
    if not x.is_cuda:
        raise hip._lib.SarsslHipError("sar_ssl_amd modules run on the GPU only (no CPU fallback); got a CPU tensor")
    RT.inference = True                       # no backward follows (no_grad, or nothing upstream / inside requires a gradient)
    try:
        return fwd(_to_rt(x.detach().contiguous()), [])
    finally:
        RT.inference = False


class _F32Tape(torch.autograd.Function):
    """Node of the downstream heads (engine.head_fwd / hip.mean_rows): f32 in, f32 out in every numeric mode."""

    @staticmethod
    def forward(ctx, fwd, bwd, x, *params):
        if not x.is_cuda:
            raise hip._lib.SarsslHipError("sar_ssl_amd modules run on the GPU only (no CPU fallback); got a CPU tensor")
        saved = []
        y = fwd(x.detach().contiguous(), saved)
        ctx.bwd, ctx.saved, ctx.in_dtype, ctx.nparams = bwd, saved, x.dtype, len(params)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = ctx.bwd(dy.contiguous().float(), ctx.saved)
        if dx is not None and dx.dtype != ctx.in_dtype:
            dx = hip.cast(dx.contiguous(), ctx.in_dtype)
        return (None, None, dx) + (None,) * ctx.nparams


def pool_mean(embed):
    """embed (B, T, d) -> f32 (B, d), mean over the frames (code/model.py:700-708: ``embed.mean(dim=1)``) through csrc/head.hip."""
    T = embed.shape[1]
    fwd = lambda x, saved: hip.mean_rows(x)
    bwd = lambda dy, saved: hip.mean_rows_bwd(dy, T, torch.float32)
    if torch.is_grad_enabled() and embed.requires_grad:
        return _F32Tape.apply(fwd, bwd, embed.float())
    return fwd(embed.detach(), [])


def head_apply(seq, pooled):
    """nn.Sequential(LayerNorm, Linear[, ReLU, Linear]) on pooled f32 embeddings through the library (engine.head_fwd / head_bwd)."""
    from . import engine
    from .runtime import gbuf                                                # noqa: F401  (engine accumulates into p.grad)
    params = [p for p in seq.parameters() if p.requires_grad]
    fwd = lambda x, saved: engine.head_fwd(x.float().contiguous(), seq, saved)
    bwd = lambda dy, saved: engine.head_bwd(dy, seq, saved)
    if torch.is_grad_enabled() and (pooled.requires_grad or params):
        return _F32Tape.apply(fwd, bwd, pooled, *params)
    return fwd(pooled.detach(), [])
