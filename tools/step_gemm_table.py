"""Per-shape table of the GEMM launches of one training step (B = 64, single stream, events around every launch):
    SARSSL_PROF_SHAPES=1 python tools/step_gemm_table.py [--precision fp16]
label = [M, N, K x batch, A layout (k = K contiguous) B layout, output dtype, split-K / activation / fused activation backward]."""
import argparse
import os
import re
import sys

os.environ["SARSSL_PROF_SHAPES"] = "1"
os.environ["SARSSL_TWO_STREAMS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa: E402,F401
import torch  # noqa: E402
from sar_ssl_amd import hip, model, runtime, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--workload", default="config2", choices=["config2", "config5"], help="config5: 4 microphones x 10 s, T = 624, batch 16 x 6 pairs")
a = ap.parse_args()
dev = torch.device("cuda:0")
runtime.set_precision(a.precision)
torch.manual_seed(1)
T_, nsample_, nmic_, batch_ = (256, 65792, 2, 64) if a.workload == "config2" else (624, 160000, 4, 16)
net = model.SARSSL(sig_shape=(256, T_, 2, 2), pretrain=True, device=dev).to(dev).train()
flat = runtime.FlatParams(net)
opt = runtime.FusedAdam(flat, lr=1e-3)
opt.zero_grad()
pcm = torch.from_numpy(synth.to_pcm16(synth.make_batch(0, batch_, nsample=nsample_, nch=nmic_))).to(dev)


def step():
    loss, _, _ = net(hip.stft_frontend(pcm, ch_mode="M" if nmic_ == 2 else "MM"))
    loss.backward()
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
import gc  # noqa: E402
gc.collect()
gc.disable()
hip.profile_start(all_calls=True)
for _ in range(a.steps):
    hip.gpu_runway(40.0)             # the profiled step is host-bound: enqueue it behind a spinning kernel so no interval contains a host gap
    step()
    torch.cuda.synchronize()
p = hip.profile_stop()
rows = []
for k, (n, ms) in p.items():
    m = re.match(r"gemm(?:_split)?\[(\d+),(\d+),(\d+) x(\d+)", k)       # gemm_split: x = K segments (FLOPs executed = segments x 2 M N K)
    if m:
        M, N, K, nb = (int(v) for v in m.groups())
        flop = 2.0 * M * N * K * nb
        rows.append((ms / a.steps, n / a.steps, 1e3 * ms / n, flop / (ms / n * 1e-3) / 1e12, k))
rows.sort(reverse=True)
print("%9s %6s %9s %8s  %s" % ("ms/step", "n/step", "us/launch", "TFLOP/s", "shape"))
for r in rows:
    print("%9.4f %6.1f %9.1f %8.1f  %s" % r)
print("total gemm ms/step: %.3f" % sum(r[0] for r in rows))
other = sorted(((ms / a.steps, n / a.steps, k) for k, (n, ms) in p.items() if k.startswith("call:")), reverse=True)
print("--- every other entry point, ms/step (launches)")
for ms, n, k in other[:40]:
    print("%9.4f %6.1f  %s" % (ms, n, k))
for k in ("call:sarssl_gemm_group_tn", "call:sarssl_splitk_reduce_multi", "call:sarssl_colsum_multi_partials", "call:sarssl_colsum_store"):
    if k in p:
        print("%-40s %.4f ms/step (%d launches)" % (k, p[k][1] / a.steps, p[k][0] / a.steps))
