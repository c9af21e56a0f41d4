"""Per-phase cycle stamps of the ping-pong convolution kernel (workgroup 0, 8 waves, first 8 tiles): builds a probe copy of the library
with -DCONV_STAMPS into gpurun_out/ and prints where a half-tile's cycles go.   python tools/conv_stamps.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "sar-ssl_amd", "csrc")
out = os.path.join(ROOT, "tmp_ab", "libconvprobe.so")        # (tmp_ab/ travels to the GPU box, gpurun_out/ does not)
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics", "-Wno-unused-result"]
objs = [os.path.join(C, f) for f in os.listdir(C) if f.endswith(".o") and f != "conv3x3.o"]
REBUILD = os.environ.get("PROBE_REBUILD", "1") != "0" or not os.path.exists(out)
if REBUILD: subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DCONV_STAMPS", "-c", os.path.join(C, "conv3x3.hip"), "-o", "/tmp/conv_probe.o"])
if REBUILD: subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, "/tmp/conv_probe.o"] + objs + ["-lpthread"])
os.environ["SARSSL_HIP_LIB"] = out
sys.path.insert(0, ROOT)
import sarssl_boot  # noqa
import torch
from sar_ssl_amd import hip, _lib
dev = torch.device("cuda:0")
B = 64
x = torch.randn((B, 256, 256, 64), device=dev).to(torch.bfloat16)
w = (torch.randn((9, 64, 64), device=dev) * 0.05).to(torch.bfloat16)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
buf = torch.zeros((8, 8, 12), dtype=torch.int64, device=dev)
for mode in ("prologue+stats", "identity"):
    for _ in range(2):
        hip.conv3x3_fwd(x, w, sc, sh, want_stats=True) if mode != "identity" else hip.conv3x3_fwd(x, w)
    torch.cuda.synchronize()
    _lib.call("sarssl_conv_stamp_buffer", _lib.c_void_p(buf.data_ptr()))
    hip.conv3x3_fwd(x, w, sc, sh, want_stats=True) if mode != "identity" else hip.conv3x3_fwd(x, w)
    torch.cuda.synchronize()
    _lib.call("sarssl_conv_stamp_buffer", _lib.c_void_p(0))
    t = buf.cpu().numpy().astype("int64")
    names = ["write_tile", "barrier1", "issue_loads", "mfma", "vmcnt0", "barrier2", "acc->lds", "drain+stats", "reduce", "barrier3"]
    print("== %s: cycles per phase (s_memtime ticks), waves 0-3 = half 0, 4-7 = half 1; tiles 2..6 averaged" % mode)
    for wv in range(8):
        d = (t[wv, 2:7, 1:11] - t[wv, 2:7, 0:10]).mean(axis=0)
        tot = (t[wv, 3:7, 0] - t[wv, 2:6, 0]).mean()
        print("wave %d: " % wv + " ".join("%s %5d" % (n, v) for n, v in zip(names, d)) + " | iteration %6d" % tot)
    h0, h1 = t[0, 2:7, 3], t[4, 2:7, 3]
    print("mfma-phase start of half 0:", (h0 - h0[0]).tolist(), " half 1:", (h1 - h0[0]).tolist())
