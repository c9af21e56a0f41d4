"""Isolated timings of the HBM-bound stem passes at the timed shape (B = 64: 64 x 256 x 256 pixels, 64 channels, bf16), with the bytes
each one has to move: 64->4 forward (+ BatchNorm(4) sums), its two-phase backward, BatchNorm-backward apply / reduce.

    python tools/bench_stem.py            # shipped library
    PROBE_ENV=1 python tools/bench_stem.py    # probe build in tmp_ab/ whose workgroup caps / read-stream counts come from SARSSL_GRID_* / *_STREAMS
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_ENV") and "SARSSL_HIP_LIB" not in os.environ:
    sys.path.insert(0, os.path.join(ROOT, "sar-ssl_amd", "csrc"))
    import build as B
    out = os.path.join(ROOT, "tmp_ab", "libstemprobe.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.environ.get("PROBE_REBUILD") or not os.path.exists(out):
        objs = []
        for s in B.SOURCES:
            o = os.path.join(ROOT, "tmp_ab", "stemprobe_" + s.replace(".hip", ".o"))
            subprocess.check_call([B._hipcc()] + B.FLAGS + (["-DSARSSL_PROBE_ENV"] if s == "stem.hip" else []) + ["-c", os.path.join(B.HERE, s), "-o", o])
            objs.append(o)
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"])
    os.environ["SARSSL_HIP_LIB"] = out
    if os.environ.get("PROBE_BUILD_ONLY"):
        sys.exit(0)

import torch  # noqa: E402
from sar_ssl_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
Bn, F, T = 64, 256, 256
U = Bn * F * T * 64 * 2 / 1e6          # MB of one 64-channel tensor


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(1)
y3 = torch.randn((Bn, F, T, 64), device=dev, generator=g).to(torch.bfloat16)
dz = torch.randn((Bn, F, T, 64), device=dev, generator=g).to(torch.bfloat16)
dy4 = torch.randn((Bn, T, F, 4), device=dev, generator=g).to(torch.bfloat16)
W4 = torch.randn((4, 64), device=dev, generator=g) * 0.1
sc, sh = torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.1
aff = torch.stack([sc, sh, sh * 0.5, sc]).contiguous()
lines = []
t = timeit(lambda: hip.stem_c4_fwd(y3, W4, sc, sh, want_stats=True))
lines.append(("stem_c4_fwd (+sums)", t, U + U / 16))
t = timeit(lambda: hip.stem_c4_bwd_two_phase(y3, dy4, W4, aff, True))
lines.append(("stem_c4_bwd two-phase (2 launches)", t, 3 * U + 2 * U / 16))
red = hip.cl_bn_bwd_reduce(dz, y3, 64, aff, 1)
t = timeit(lambda: hip.cl_bn_bwd_reduce(dz, y3, 64, aff, 1))
lines.append(("cl_bn_bwd_reduce C=64", t, 2 * U))
out = torch.empty_like(dz)
t = timeit(lambda: hip.cl_bn_bwd_apply(dz, y3, 64, aff, 1, False, True, red, out=out))
lines.append(("cl_bn_bwd_apply C=64", t, 3 * U))
for name, us, mb in lines:
    print("%-38s %8.1f us   %7.1f MB   %5.2f TB/s" % (name, us, mb, mb / us), flush=True)
