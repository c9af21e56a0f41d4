"""Conformer convolution module (code/common/conformer/convolution.py:24-149)."""
import torch.nn as nn

from ... import engine
from ...autograd import tape_apply
from .activation import Swish, GLU
from .modules import Transpose


class DepthwiseConv1d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        assert out_channels % in_channels == 0, "out_channels should be constant multiple of in_channels"
        self.conv = nn.Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                              groups=in_channels, stride=stride, padding=padding, bias=bias)


class PointwiseConv1d(nn.Module):
    def __init__(self, in_channels, out_channels, stride=1, padding=0, bias=True):
        super().__init__()
        self.conv = nn.Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=stride,
                              padding=padding, bias=bias)


class ConformerConvModule(nn.Module):
    """LN -> PW conv (GEMM) -> GLU -> depthwise k=31 -> BatchNorm1d -> Swish -> PW conv (GEMM) -> Dropout."""

    def __init__(self, in_channels: int, kernel_size: int = 31, expansion_factor: int = 2, dropout_p: float = 0.1):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0, "kernel_size should be a odd number for 'SAME' padding"
        assert expansion_factor == 2, "Currently, Only Supports expansion_factor 2"
        self.sequential = nn.Sequential(
            nn.LayerNorm(in_channels),
            Transpose(shape=(1, 2)),
            PointwiseConv1d(in_channels, in_channels * expansion_factor, stride=1, padding=0, bias=True),
            GLU(dim=1),
            DepthwiseConv1d(in_channels, in_channels, kernel_size, stride=1, padding=(kernel_size - 1) // 2),
            nn.BatchNorm1d(in_channels),
            Swish(),
            PointwiseConv1d(in_channels, in_channels, stride=1, padding=0, bias=True),
            nn.Dropout(p=dropout_p),
        )

    def forward_residual(self, inputs, factor=1.0):
        assert factor == 1.0
        B, T, d = inputs.shape
        return tape_apply(
            self,
            lambda x, saved: engine.convmod_fwd(x.view(B * T, d), self, B, T, self.training, saved).view(B, T, d),
            lambda dy, saved: engine.convmod_bwd(dy.view(B * T, d), self, saved).view(B, T, d),
            inputs)

    def forward(self, inputs):
        return self.forward_residual(inputs) - inputs
