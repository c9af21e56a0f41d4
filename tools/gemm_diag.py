"""GEMM diagnostics: where does the time of the short-K FFN GEMM go?  (epilogue-only K=64, plain store, each epilogue feature)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip
from bench_kernels import timeit, dev

def case(name, M, N, K, out_dtype=torch.bfloat16, **kw):
    A = torch.randn((M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K), device=dev).to(torch.bfloat16)
    out = torch.zeros((M, N), dtype=out_dtype, device=dev)
    t = timeit(lambda: hip.gemm(A, B, M=M, N=N, K=K, lda=K, ldb=K, out=out, **kw), n=30)
    byts = 2.0 * (M * K + N * K) + out.element_size() * M * N * (2 if "preact" in kw else 1) + (2.0 * M * N if ("resid" in kw or "aux" in kw) else 0)
    print("%-44s M=%6d N=%5d K=%5d %8.1f us %7.1f TF/s %7.1f GB/s" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12, byts / t / 1e9))

M = 16384
bias = torch.zeros(4096, device=dev)
pre = torch.empty((M, 2048), dtype=torch.bfloat16, device=dev)
res = torch.randn((M, 2048), device=dev).to(torch.bfloat16)
for N in (2048, 512):
    case("epilogue only (K=64), plain", M, N, 64)
    case("K=512 plain", M, N, 512)
    case("K=512 +bias", M, N, 512, bias=bias[:N])
    case("K=512 +bias+swish", M, N, 512, bias=bias[:N], act=2)
    case("K=512 +bias+swish+preact", M, N, 512, bias=bias[:N], act=2, preact=pre[:, :N].contiguous())
    case("K=512 +bias+swish+preact+dropout", M, N, 512, bias=bias[:N], act=2, preact=pre[:, :N].contiguous(), p_drop=0.1, seed=5)
    case("K=512 +resid", M, N, 512, resid=res[:, :N].contiguous(), ldr=N)
    case("K=2048 plain", M, N, 2048)
    case("K=512 f32 out", M, N, 512, out_dtype=torch.float32)
a = torch.randn((M * 2048,), device=dev).to(torch.bfloat16); b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a), n=30)
print("torch copy 67MB->67MB: %.1f us %.1f GB/s" % (t * 1e6, 2 * a.numel() * 2 / t / 1e9))
