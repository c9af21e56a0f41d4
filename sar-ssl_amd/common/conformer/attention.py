"""Relative-position multi-head self-attention (code/common/conformer/attention.py:26-151)."""
import math

import torch
import torch.nn as nn

from ... import engine
from ...autograd import tape_apply
from .embedding import PositionalEncoding
from .modules import Linear


class RelativeMultiHeadAttention(nn.Module):
    """Parameter container with the reference layout (attention.py:46-70); the computation (u/v biases, relative
    shift, 1/sqrt(d_model) scaling, softmax, dropout) runs in engine.mhsa_fwd / mhsa_bwd."""

    def __init__(self, d_model: int = 512, num_heads: int = 16, dropout_p: float = 0.1):
        super().__init__()
        assert d_model % num_heads == 0, "d_model % num_heads should be zero."
        self.d_model = d_model
        self.d_head = int(d_model / num_heads)
        self.num_heads = num_heads
        self.sqrt_dim = math.sqrt(d_model)
        self.query_proj = Linear(d_model, d_model)
        self.key_proj = Linear(d_model, d_model)
        self.value_proj = Linear(d_model, d_model)
        self.pos_proj = Linear(d_model, d_model, bias=False)
        self.dropout = nn.Dropout(p=dropout_p)
        self.u_bias = nn.Parameter(torch.Tensor(self.num_heads, self.d_head))
        self.v_bias = nn.Parameter(torch.Tensor(self.num_heads, self.d_head))
        torch.nn.init.xavier_uniform_(self.u_bias)
        torch.nn.init.xavier_uniform_(self.v_bias)
        self.out_proj = Linear(d_model, d_model)


class MultiHeadedSelfAttentionModule(nn.Module):
    def __init__(self, d_model: int, num_heads: int, dropout_p: float = 0.1):
        super().__init__()
        self.positional_encoding = PositionalEncoding(d_model)
        self.layer_norm = nn.LayerNorm(d_model)
        self.attention = RelativeMultiHeadAttention(d_model, num_heads, dropout_p)
        self.dropout = nn.Dropout(p=dropout_p)

    def forward_residual(self, inputs, factor=1.0):
        assert factor == 1.0
        B, T, d = inputs.shape
        return tape_apply(
            self,
            lambda x, saved: engine.mhsa_fwd(x.view(B * T, d), self, B, T, self.training, saved).view(B, T, d),
            lambda dy, saved: engine.mhsa_bwd(dy.view(B * T, d), self, saved).view(B, T, d),
            inputs)

    def forward(self, inputs, mask=None):
        assert mask is None, "attention masks are not used on the SAR-SSL path"
        return self.forward_residual(inputs) - inputs
