"""The d = 256 FFN products with their step epilogues, timed over a ROTATION of buffer sets larger than the 256 MB memory-side cache
(what a launch sees inside the step: operands written long ago or by the previous kernel, outputs going to lines nobody has touched)
next to the usual same-buffers loop.    python tools/gemm_diag_cold.py [nsets]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip
dev = torch.device("cuda:0")
M = 16384
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timeit(fns, n=48):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def case(name, N, K, dt_=torch.float16, preact=False, resid=False, **kw):
    B = torch.randn((N, K), device=dev).to(dt_)
    sets = []
    for _ in range(NS):
        A = torch.randn((M, K), device=dev).to(dt_)
        out = torch.zeros((M, N), dtype=dt_, device=dev)
        extra = dict(kw)
        if preact:
            extra["preact"] = torch.empty((M, N), dtype=dt_, device=dev)
        if resid:
            extra.update(resid=torch.randn((M, N), device=dev).to(dt_), ldr=N, res_scale=1.0)
        sets.append((A, out, extra))
    mk = lambda s: (lambda: hip.gemm(s[0], B, M=M, N=N, K=K, lda=K, ldb=K, out=s[1], **s[2]))
    warm = timeit([mk(sets[0])])
    cold = timeit([mk(s) for s in sets])
    print("%-40s N=%5d K=%5d   same buffers %6.1f us   rotating %d sets %6.1f us" % (name, N, K, warm, NS, cold), flush=True)


bias = torch.zeros(4096, device=dev)
case("ffn1 plain", 1024, 256)
case("ffn1 as in the step", 1024, 256, preact=True, bias=bias[:1024], act=2, p_drop=0.1, seed=5)
case("ffn2 plain", 256, 1024)
case("ffn2 as in the step", 256, 1024, resid=True, bias=bias[:256], p_drop=0.1, seed=5, out_scale=0.5)
case("ffn1 d=512 as in the step", 2048, 512, preact=True, bias=bias[:2048], act=2, p_drop=0.1, seed=5)
case("ffn2 d=512 as in the step", 512, 2048, resid=True, bias=bias[:512], p_drop=0.1, seed=5, out_scale=0.5)
case("decoder1 (N=3072, K=768) relu", 3072, 768, bias=bias[:3072], act=1)
case("decoder2 (N=1024, K=3072)", 1024, 3072, bias=bias[:1024])
case("projection N=256 K=256", 256, 256, bias=bias[:256])
# yardstick: torch copy over the same rotation
xs = [torch.randn((M * 1024,), device=dev).to(torch.float16) for _ in range(NS)]
ys = [torch.empty_like(x) for x in xs]
print("torch copy 33.5 MB -> 33.5 MB: same %.1f us, rotating %.1f us" % (timeit([lambda: ys[0].copy_(xs[0])]), timeit([(lambda i=i: ys[i].copy_(xs[i])) for i in range(NS)])))
