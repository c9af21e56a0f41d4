"""Probe copy of the library with the direct-to-LDS pipelined NT GEMM (tools/gemm_nt/gemm_nt.hip) linked in and dispatched from sarssl_gemm
(-DSARSSL_WITH_GEMM_NT; SARSSL_GEMM_NT=0 switches it off again, SARSSL_GEMM_NT_CFG=<n> forces one tile configuration) - round-3 experiment,
not part of the product library (NOTES.md 4.2).  hipcc 7.2 note: simplifycfg segfaults when it sinks "common" instructions out of branches
that hold the LDS-DMA intrinsic (llvm.amdgcn.raw.ptr.buffer.load.lds has immediate operands): -mllvm -simplifycfg-sink-common=false."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
C = os.path.join(ROOT, "sar-ssl_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics", "-Wno-unused-result",
         "-DSARSSL_WITH_GEMM_NT", "-I" + C]


def build(out, stamps=False):
    """-> path of the probe .so (kept inside the repository tree so that it travels to the GPU box)."""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flags = FLAGS + (["-DGEMM_STAMPS"] if stamps else [])
    tmp = os.path.join(os.path.dirname(out), "_obj_stamps" if stamps else "_obj")
    os.makedirs(tmp, exist_ok=True)
    objs = [os.path.join(C, f) for f in os.listdir(C) if f.endswith(".o") and f not in ("gemm.o", "gemm_fp8.o")]
    mine = []
    for src, extra in ((os.path.join(C, "gemm.hip"), []), (os.path.join(C, "gemm_fp8.hip"), []),
                       (os.path.join(HERE, "gemm_nt.hip"), ["-mllvm", "-simplifycfg-sink-common=false"])):
        o = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + extra + ["-c", src, "-o", o])
        mine.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + mine + objs + ["-lpthread"])
    return out
