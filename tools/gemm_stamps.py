"""Per-phase cycle stamps of the GEMM K loop (workgroup (0,0,0), its 4 waves, first 16 K-tiles): builds a probe copy of the library with
-DGEMM_STAMPS into gpurun_out/ and prints where a K-tile's cycles go for the hot shapes.   python tools/gemm_stamps.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "sar-ssl_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "libgemmprobe.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics", "-Wno-unused-result"]
objs = [os.path.join(C, f) for f in os.listdir(C) if f.endswith(".o") and f != "gemm.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DGEMM_STAMPS", "-c", os.path.join(C, "gemm.hip"), "-o", "/tmp/gemm_probe.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, "/tmp/gemm_probe.o"] + objs + ["-lpthread"])
os.environ["SARSSL_HIP_LIB"] = out
sys.path.insert(0, ROOT)
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip, _lib
dev = torch.device("cuda:0")
M = 16384
buf = torch.zeros((4, 16, 8), dtype=torch.int64, device=dev)
names = ["lds store", "barrier1", "issue loads", "mfma phase", "barrier2", "loop"]
for label, N, K, b_kc in (("decoder2 NT", 1024, 3072, True), ("decoder1 NT", 3072, 768, True), ("ffn1 d512 NT", 2048, 512, True),
                          ("ffn2 d512 NT", 512, 2048, True), ("ffn1 dX NN d512", 512, 2048, False), ("ffn2 d256 NT", 256, 1024, True)):
    A = torch.randn((M, K), device=dev).bfloat16()
    B = (torch.randn((N, K) if b_kc else (K, N), device=dev) * 0.05).bfloat16()
    o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    run = lambda: hip.gemm(A, B, a_kc=True, b_kc=b_kc, M=M, N=N, K=K, lda=K, ldb=B.shape[1], out=o)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    _lib.call("sarssl_gemm_stamp_buffer", _lib.c_void_p(buf.data_ptr()))
    run(); torch.cuda.synchronize()
    _lib.call("sarssl_gemm_stamp_buffer", _lib.c_void_p(0))
    t = buf.cpu().numpy().astype("int64")
    nk = min(16, K // 64)
    print("== %s  M=%d N=%d K=%d: %.1f us; cycles per phase (s_memtime ticks = 100 MHz? see loop total), K-tiles 2..%d averaged" % (label, M, N, K, e0.elapsed_time(e1) * 1e3, nk - 2))
    for wv in range(4):
        d = (t[wv, 2:nk - 1, 1:6] - t[wv, 2:nk - 1, 0:5]).mean(axis=0)
        tot = (t[wv, 3:nk - 1, 0] - t[wv, 2:nk - 2, 0]).mean()
        print("  wave %d: " % wv + "  ".join("%s %6.0f" % (n, v) for n, v in zip(names, d)) + "  | K-tile %6.0f" % tot)
