"""Thin building blocks with the reference's names and state_dict keys
(code/common/conformer/modules.py:21-72)."""
import torch
import torch.nn as nn
import torch.nn.init as init

from ... import engine
from ...autograd import tape_apply
from ...runtime import wt, wtg, gbuf
from ... import hip


class ResidualConnectionModule(nn.Module):
    """outputs = module(inputs) * module_factor + inputs * input_factor.  The fused Conformer path folds the
    residual into the last GEMM epilogue of the wrapped module (engine.py); this forward is the standalone form."""

    def __init__(self, module: nn.Module, module_factor: float = 1.0, input_factor: float = 1.0):
        super().__init__()
        self.module = module
        self.module_factor = module_factor
        self.input_factor = input_factor

    def forward(self, inputs):
        fused = getattr(self.module, "forward_residual", None)
        if fused is not None and self.input_factor == 1.0:
            return fused(inputs, self.module_factor)
        return (self.module(inputs) * self.module_factor) + (inputs * self.input_factor)


class Linear(nn.Module):
    """nn.Linear with xavier-uniform weight / zero bias (modules.py:35-48), computed by the MFMA GEMM."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True) -> None:
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        init.xavier_uniform_(self.linear.weight)
        if bias:
            init.zeros_(self.linear.bias)

    def forward(self, x):
        lead = x.shape[:-1]
        lin = self.linear

        def fwd(xx, saved):
            x2 = xx.reshape(-1, lin.in_features)
            saved.append(x2)
            return engine.mm_nt(x2, wt(lin.weight), bias=lin.bias.data if lin.bias is not None else None).view(*lead, -1)

        def bwd(dy, saved):
            x2 = saved.pop()
            d2 = dy.reshape(-1, lin.out_features)
            engine.mm_tn_acc(d2, x2, gbuf(lin.weight))
            if lin.bias is not None:
                hip.colsum(d2, gbuf(lin.bias))
            return engine.mm_nn(d2, wtg(lin.weight)).view(*lead, -1)
        return tape_apply(self, fwd, bwd, x)


class View(nn.Module):
    def __init__(self, shape: tuple, contiguous: bool = False):
        super().__init__()
        self.shape = shape
        self.contiguous = contiguous

    def forward(self, x):
        if self.contiguous:
            x = x.contiguous()
        return x.view(*self.shape)


class Transpose(nn.Module):
    def __init__(self, shape: tuple):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.transpose(*self.shape)
