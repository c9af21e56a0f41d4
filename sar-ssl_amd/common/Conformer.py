"""ConformerBlock / ConformerEncoder (code/common/Conformer.py:16-195): macaron FFN halves around relative-position
MHSA and the convolution module, final LayerNorm; no subsampling / input projection (commented out upstream)."""
import torch.nn as nn

from .. import engine
from ..autograd import tape_apply
from .conformer.feed_forward import FeedForwardModule
from .conformer.attention import MultiHeadedSelfAttentionModule
from .conformer.convolution import ConformerConvModule
from .conformer.modules import ResidualConnectionModule


class ConformerBlock(nn.Module):
    def __init__(self, encoder_dim: int = 512, num_attention_heads: int = 8, feed_forward_expansion_factor: int = 4,
                 conv_expansion_factor: int = 2, feed_forward_dropout_p: float = 0.1, attention_dropout_p: float = 0.1,
                 conv_dropout_p: float = 0.1, conv_kernel_size: int = 31, half_step_residual: bool = True):
        super().__init__()
        self.feed_forward_residual_factor = 0.5 if half_step_residual else 1
        self.sequential = nn.Sequential(
            ResidualConnectionModule(
                module=FeedForwardModule(encoder_dim=encoder_dim, expansion_factor=feed_forward_expansion_factor,
                                         dropout_p=feed_forward_dropout_p),
                module_factor=self.feed_forward_residual_factor),
            ResidualConnectionModule(module=MultiHeadedSelfAttentionModule(
                d_model=encoder_dim, num_heads=num_attention_heads, dropout_p=attention_dropout_p)),
            ResidualConnectionModule(module=ConformerConvModule(
                in_channels=encoder_dim, kernel_size=conv_kernel_size, expansion_factor=conv_expansion_factor,
                dropout_p=conv_dropout_p)),
            ResidualConnectionModule(
                module=FeedForwardModule(encoder_dim=encoder_dim, expansion_factor=feed_forward_expansion_factor,
                                         dropout_p=feed_forward_dropout_p),
                module_factor=self.feed_forward_residual_factor),
            nn.LayerNorm(encoder_dim),
        )

    def forward(self, inputs):
        B, T, d = inputs.shape
        return tape_apply(
            self,
            lambda x, saved: engine.block_fwd(x.view(B * T, d), self, B, T, self.training, saved).view(B, T, d),
            lambda dy, saved: engine.block_bwd(dy.view(B * T, d), self, saved).view(B, T, d),
            inputs)


class ConformerEncoder(nn.Module):
    def __init__(self, input_dim: int = 80, encoder_dim: int = 256, num_layers: int = 6, num_attention_heads: int = 4,
                 feed_forward_expansion_factor: int = 4, conv_expansion_factor: int = 2, input_dropout_p: float = 0.1,
                 feed_forward_dropout_p: float = 0.1, attention_dropout_p: float = 0.1, conv_dropout_p: float = 0.1,
                 conv_kernel_size: int = 31, half_step_residual: bool = True):
        super().__init__()
        self.layers = nn.ModuleList([
            ConformerBlock(encoder_dim=encoder_dim, num_attention_heads=num_attention_heads,
                           feed_forward_expansion_factor=feed_forward_expansion_factor,
                           conv_expansion_factor=conv_expansion_factor, feed_forward_dropout_p=feed_forward_dropout_p,
                           attention_dropout_p=attention_dropout_p, conv_dropout_p=conv_dropout_p,
                           conv_kernel_size=conv_kernel_size, half_step_residual=half_step_residual)
            for _ in range(num_layers)])

    def count_parameters(self) -> int:
        return sum([p.numel() for p in self.parameters()])

    def update_dropout(self, dropout_p: float) -> None:
        for name, child in self.named_children():
            if isinstance(child, nn.Dropout):
                child.p = dropout_p

    def forward(self, inputs, add_same_one=False):
        assert not add_same_one, "add_same_one is always False on the SAR-SSL path (code/model.py:565)"
        B, T, d = inputs.shape
        return tape_apply(
            self,
            lambda x, saved: engine.encoder_fwd(x.view(B * T, d), self, B, T, self.training, saved).view(B, T, d),
            lambda dy, saved: engine.encoder_bwd(dy.view(B * T, d), self, saved).view(B, T, d),
            inputs)
