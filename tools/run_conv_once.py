"""Runs conv3x3_fwd_kernel (BN+ReLU prologue, B=64, bf16) a few times - target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip
dev = torch.device("cuda:0")
x = torch.randn((64, 256, 256, 64), device=dev).to(torch.bfloat16)
w = (torch.randn((9, 64, 64), device=dev) * 0.05).to(torch.bfloat16)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
for _ in range(5):
    y = hip.conv3x3_fwd(x, w, sc, sh)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
