"""Swish / GLU activations (interface of code/common/conformer/activation.py:19-42).  In the fused Conformer path
these never run as separate kernels (Swish lives in the GEMM epilogue); standalone calls use the HIP elementwise
kernels through autograd."""
import torch
import torch.nn as nn

from ... import hip
from ...autograd import tape_apply


class Swish(nn.Module):
    def forward(self, inputs):
        shape = inputs.shape

        def fwd(x, saved):
            saved.append(x)
            assert x.numel() % 4 == 0, "Swish: numel must be a multiple of 4"
            one = torch.ones((4,), dtype=torch.float32, device=x.device)
            zero = torch.zeros((4,), dtype=torch.float32, device=x.device)
            return hip.cl_affine_act(x.view(-1, 4), 4, (one, zero), 2).view(shape)

        def bwd(dy, saved):
            x = saved.pop()
            return hip.act_bwd(dy.view(-1), x.view(-1), 2).view(shape)
        return tape_apply(self, fwd, bwd, inputs)


class GLU(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, inputs):
        # (batch, 2d, time) with dim=1 as used by the conv module, or (..., 2d) with dim=-1
        x = inputs.transpose(self.dim, -1) if self.dim not in (-1, inputs.dim() - 1) else inputs
        lead = x.shape[:-1]
        d = x.shape[-1] // 2

        def fwd(h, saved):
            h2 = h.reshape(-1, 2 * d)
            saved.append(h2)
            return hip.glu_fwd(h2).view(*lead, d)

        def bwd(dg, saved):
            h2 = saved.pop()
            return hip.glu_bwd(dg.reshape(-1, d), h2).view(*lead, 2 * d)
        y = tape_apply(self, fwd, bwd, x.contiguous())
        return y.transpose(self.dim, -1) if self.dim not in (-1, inputs.dim() - 1) else y
