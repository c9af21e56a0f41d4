"""d = 256 FFN shapes (the largest pool of the GEMM family): each epilogue feature's cost in isolation, fp16 forward operands."""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip
from bench_kernels import timeit, dev
M = 16384
def case(name, N, K, dt_=torch.float16, **kw):
    A = torch.randn((M, K), device=dev).to(dt_); B = torch.randn((N, K), device=dev).to(dt_)
    out = torch.zeros((M, N), dtype=dt_, device=dev)
    t = timeit(lambda: hip.gemm(A, B, M=M, N=N, K=K, lda=K, ldb=K, out=out, **kw), n=50)
    byts = 2.0 * (M * K + N * K) + 2.0 * M * N * (2 if "preact" in kw else 1) + (2.0 * M * N if ("resid" in kw or "aux" in kw) else 0)
    print("%-46s N=%5d K=%5d %7.1f us %7.1f TF/s %7.0f GB/s" % (name, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12, byts / t / 1e9), flush=True)
bias = torch.zeros(4096, device=dev)
pre = torch.empty((M, 1024), dtype=torch.float16, device=dev)
res = torch.randn((M, 1024), device=dev).to(torch.float16)
case("ffn1 plain", 1024, 256)
case("ffn1 +bias+swish", 1024, 256, bias=bias[:1024], act=2)
case("ffn1 +bias+swish+preact", 1024, 256, bias=bias[:1024], act=2, preact=pre)
case("ffn1 +bias+swish+preact+dropout (as in step)", 1024, 256, bias=bias[:1024], act=2, preact=pre, p_drop=0.1, seed=5)
case("ffn2 plain", 256, 1024)
case("ffn2 +bias+dropout+resid (as in step)", 256, 1024, bias=bias[:256], p_drop=0.1, seed=5, resid=res[:, :256].contiguous(), ldr=256, res_scale=1.0, out_scale=0.5)
case("qkv-like plain N=768", 768, 256)
case("proj plain N=256 K=256", 256, 256)
a = torch.randn((M * 1024,), device=dev).to(torch.float16); b = torch.empty_like(a); c = torch.empty_like(a)
t = timeit(lambda: b.copy_(a), n=50)
print("torch copy 33.5MB->33.5MB: %.1f us %.0f GB/s" % (t * 1e6, 2 * a.numel() * 2 / t / 1e9))
