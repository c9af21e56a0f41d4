"""Library yardstick (torch.matmul -> hipBLASLt) for the hot GEMM shapes of the step at M = 16384: forward NT, dX NN and dW TN products.
Not part of the product path (the step runs csrc/gemm.hip); used to judge how far the hand-written kernel is from the library on the plain
large shapes (NOTES.md 7: library 1.3-1.45x faster on forward / dX, 1.5-3x slower on the weight gradients)."""
import torch
dev = "cuda"
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
M = 16384
for name, N, K in (("ffn1 d512", 2048, 512), ("ffn2 d512", 512, 2048), ("qkv d512", 1536, 512), ("dec1", 3072, 768), ("dec2", 1024, 3072),
                   ("ffn1 d256", 1024, 256), ("ffn2 d256", 256, 1024), ("proj d256", 256, 256), ("qkv d256", 768, 256), ("patch", 512, 1024)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16() * 0.05
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(A, W.t(), out=out))
    dY = torch.randn(M, N, device=dev).bfloat16()
    us_dx = t(lambda: torch.matmul(dY, W))            # dX = dY @ W  (NN)
    us_dw = t(lambda: torch.matmul(dY.t(), A))        # dW = dY^T @ A (TN)
    fl = 2.0 * M * N * K
    print("%-10s N=%4d K=%4d  fwd NT %6.1f us (%4.0f TF)   dX NN %6.1f us (%4.0f TF)   dW TN %6.1f us (%4.0f TF)" % (name, N, K, us, fl / us / 1e6, us_dx, fl / us_dx / 1e6, us_dw, fl / us_dw / 1e6))
