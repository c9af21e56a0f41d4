"""Per-phase cycle stamps of the pipelined NT GEMM's K loop (tools/gemm_nt/gemm_nt.hip, round-3 experiment; workgroup (0,0), all waves, first 24 K-tiles): builds a probe
copy of the library with -DGEMM_STAMPS into tmp_ab/ and prints where a K-tile's cycles go.   SARSSL_GEMM_NT_CFG=<n> python tools/gemm_nt/gemm_nt_stamps.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
out = os.path.join(ROOT, "tmp_ab", "libgemmntstamps.so")
if "--build-only" in sys.argv or not os.path.exists(out):
    sys.path.insert(0, HERE)
    import build_probe
    build_probe.build(out, stamps=True)
    if "--build-only" in sys.argv:
        sys.exit(0)
cfg = os.environ.get("SARSSL_GEMM_NT_CFG", "auto")
os.environ["SARSSL_HIP_LIB"] = out
sys.path.insert(0, ROOT)
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip, _lib
dev = torch.device("cuda:0")
buf = torch.zeros((8, 24, 8), dtype=torch.int64, device=dev)
names = ["wait vmcnt", "barrier", "issue DMA", "ds_read+mfma", "loop"]
for label, M, N, K in (("decoder2", 16384, 1024, 3072), ("ffn1 d512", 16384, 2048, 512), ("ffn2 d512", 16384, 512, 2048)):
    A = torch.randn((M, K), device=dev).bfloat16()
    B = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    run = lambda: hip.gemm(A, B, M=M, N=N, K=K, lda=K, ldb=K, out=o)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    buf.zero_()
    _lib.call("sarssl_gemm_stamp_buffer", _lib.c_void_p(buf.data_ptr()))
    run(); torch.cuda.synchronize()
    _lib.call("sarssl_gemm_stamp_buffer", _lib.c_void_p(0))
    t = buf.cpu().numpy().astype("int64")
    nk = int((t[0, :23, 0] != 0).sum())
    print("== cfg %s  %s  M=%d N=%d K=%d: %.1f us; cycles per phase, K-tiles 3..%d averaged" % (cfg, label, M, N, K, e0.elapsed_time(e1) * 1e3, nk - 2))
    w = t[:, 23, :4]
    for wv in range(8):
        if w[wv, 0]:
            print("  wave %d whole kernel: prologue (entry -> first tile visible) %6d   K loop %7d   epilogue %6d cycles" % (wv, w[wv, 1] - w[wv, 0], w[wv, 2] - w[wv, 1], w[wv, 3] - w[wv, 2]))
            break
    if nk < 6:
        print("   (too few stamped tiles)"); continue
    for wv in range(8):
        if t[wv, 3, 0] == 0:
            continue
        d = (t[wv, 3:nk - 1, 1:5] - t[wv, 3:nk - 1, 0:4]).mean(axis=0)
        tot = (t[wv, 4:nk - 1, 0] - t[wv, 3:nk - 2, 0]).mean()
        print("  wave %d: " % wv + "  ".join("%s %6.0f" % (n, v) for n, v in zip(names, d)) + "  | K-tile %6.0f" % tot)
