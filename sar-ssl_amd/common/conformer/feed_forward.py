"""FeedForwardModule (code/common/conformer/feed_forward.py:23-57): LN -> Linear -> Swish -> Dropout -> Linear ->
Dropout, run as LayerNorm kernel + two MFMA GEMMs with fused bias/Swish/dropout/residual epilogues."""
import torch.nn as nn

from ... import engine
from ...autograd import tape_apply
from .activation import Swish
from .modules import Linear


class FeedForwardModule(nn.Module):
    def __init__(self, encoder_dim: int = 512, expansion_factor: int = 4, dropout_p: float = 0.1) -> None:
        super().__init__()
        self.sequential = nn.Sequential(
            nn.LayerNorm(encoder_dim),
            Linear(encoder_dim, encoder_dim * expansion_factor, bias=True),
            Swish(),
            nn.Dropout(p=dropout_p),
            Linear(encoder_dim * expansion_factor, encoder_dim, bias=True),
            nn.Dropout(p=dropout_p),
        )

    def forward_residual(self, inputs, factor=1.0):
        """inputs + factor * self(inputs) with the residual fused into the second GEMM."""
        shape = inputs.shape
        return tape_apply(
            self,
            lambda x, saved: engine.ffn_fwd(x.view(-1, shape[-1]), self, factor, self.training, saved).view(shape),
            lambda dy, saved: engine.ffn_bwd(dy.view(-1, shape[-1]), self, saved).view(shape),
            inputs)

    def forward(self, inputs):
        return self.forward_residual(inputs, 1.0) - inputs
