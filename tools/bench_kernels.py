"""Micro-benchmarks of the main kernels at BASELINE config-2 shapes (B=64): prints TFLOP/s or GB/s per kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip

dev = torch.device("cuda:0")


def timeit(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def gemm_case(name, M, N, K, a_kc=True, b_kc=True, out_dtype=torch.bfloat16, split=0, **kw):
    A = torch.randn((M, K) if a_kc else (K, M), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev).to(torch.bfloat16)
    out = torch.zeros((M, N), dtype=out_dtype, device=dev)
    t = timeit(lambda: hip.gemm(A, B, a_kc=a_kc, b_kc=b_kc, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], out=out, split_k=split, **kw))
    print("%-34s M=%6d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))


def fp8_case(name, M, N, K, **kw):
    A = torch.randn((M, K), device=dev).to(torch.bfloat16)
    B = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    Bq, sb = hip.fp8_quantize(B)
    Aq, sa = hip.fp8_quantize(A)
    t_mm = timeit(lambda: hip.gemm_fp8(Aq, sa, Bq, sb, M=M, N=N, K=K, out=out, **kw))
    t_q = timeit(lambda: hip.fp8_quantize(A))
    t_bf = timeit(lambda: hip.gemm(A, B, M=M, N=N, K=K, lda=K, ldb=K, out=out, **kw))
    print("%-34s M=%6d N=%5d K=%6d  fp8 gemm %7.1f us (%6.1f TFLOP/s) + activation quantise %6.1f us | bf16 gemm %7.1f us | "
          "speed-up gemm only %.2fx, incl. quantise %.2fx" % (name, M, N, K, t_mm * 1e6, 2.0 * M * N * K / t_mm / 1e12, t_q * 1e6, t_bf * 1e6,
                                                              t_bf / t_mm, t_bf / (t_mm + t_q)))


if __name__ == "__main__":
    if "--fp8" in sys.argv:
        bias = torch.zeros(4096, device=dev)
        for d in (512, 256):
            fp8_case("fp8 ffn1 NT d=%d (+bias,swish)" % d, 16384, 4 * d, d, bias=bias[:4 * d], act=2)
            fp8_case("fp8 ffn2 NT d=%d" % d, 16384, d, 4 * d, bias=bias[:d])
            fp8_case("fp8 qkv NT d=%d" % d, 16384, 3 * d, d, bias=bias[:3 * d])
        fp8_case("fp8 decoder1 NT", 16384, 3072, 768, bias=bias[:3072], act=1)
        fp8_case("fp8 decoder2 NT", 16384, 1024, 3072, bias=bias[:1024])
        fp8_case("fp8 big square", 8192, 8192, 8192)
        sys.exit(0)
    Mr = 16384
    bias = torch.zeros(4096, device=dev)
    for d in (512, 256):
        gemm_case("ffn1 NT d=%d (+bias,swish,preact)" % d, Mr, 4 * d, d, bias=bias[:4 * d], act=2)
        gemm_case("ffn1 NT d=%d (... + dropout 0.1)" % d, Mr, 4 * d, d, bias=bias[:4 * d], act=2, p_drop=0.1, seed=7)
        gemm_case("ffn2 NT d=%d" % d, Mr, d, 4 * d, bias=bias[:d])
        gemm_case("ffn2 NT d=%d (+ dropout 0.1)" % d, Mr, d, 4 * d, bias=bias[:d], p_drop=0.1, seed=7)
        gemm_case("qkv NT d=%d" % d, Mr, d, d, bias=bias[:d])
        gemm_case("ffn1 dX NN d=%d" % d, Mr, d, 4 * d, b_kc=False)
        gemm_case("ffn2 dX NN d=%d" % d, Mr, 4 * d, d, b_kc=False)
        gemm_case("ffn1 dW TN d=%d" % d, 4 * d, d, Mr, a_kc=False, b_kc=False, out_dtype=torch.float32, split=12)
        gemm_case("qkv  dW TN d=%d" % d, d, d, Mr, a_kc=False, b_kc=False, out_dtype=torch.float32, split=48)
    gemm_case("decoder1 NT", Mr, 3072, 768, bias=bias[:3072], act=1)
    gemm_case("decoder2 NT", Mr, 1024, 3072, bias=bias[:1024])
    gemm_case("patch NT d=512", Mr, 512, 1024)
    gemm_case("big square", 8192, 8192, 8192)
    # conv
    x = torch.randn((64, 256, 256, 64), device=dev).to(torch.bfloat16)
    w = (torch.randn((9, 64, 64), device=dev) * 0.05).to(torch.bfloat16)
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    fl = 2.0 * 64 * 65536 * 64 * 576
    t = timeit(lambda: hip.conv3x3_fwd(x, w, sc, sh), n=10)
    print("conv3x3_fwd (prologue)   %8.1f us  %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    t = timeit(lambda: hip.conv3x3_fwd(x, w), n=10)
    print("conv3x3_fwd (identity)   %8.1f us  %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    t = timeit(lambda: hip.conv3x3_wgrad(x, x, sc, sh), n=10)
    print("conv3x3_wgrad            %8.1f us  %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    aff = torch.stack([sc, sh, sh, sc]).contiguous()
    y2 = torch.randn((64, 256, 256, 64), device=dev).to(torch.bfloat16)
    red = hip.cl_bn_bwd_reduce(x, y2, 64, aff, 1)
    t0 = timeit(lambda: hip.cl_bn_bwd_apply(x, y2, 64, aff, 1, False, True, red), n=10)
    print("cl_bn_bwd_apply %6.1f us" % (t0 * 1e6))
