"""Host-side enqueue time per step (no GPU sync inside the loop) vs GPU wall time."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import numpy as np, torch
from sar_ssl_amd import hip, model, runtime, synth
dev = torch.device("cuda:0")
net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device=dev).to(dev).train()
flat = runtime.FlatParams(net); opt = runtime.FusedAdam(flat, lr=1e-3); opt.zero_grad()
pcm = torch.from_numpy(synth.to_pcm16(np.repeat(synth.make_batch(0, 4), 16, axis=0))).to(dev)
def step():
    x = hip.stft_frontend(pcm); loss, _, _ = net(x); loss.backward(); opt.step(); opt.zero_grad()
for _ in range(3): step()
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N): step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.2f ms/step, total wall %.2f ms/step" % (1e3 * t_host / N, 1e3 * t_all / N))
# the captured step: host time of one replay with an idle GPU (hipGraphLaunch + mask draw/upload), and back to back
from sar_ssl_amd.graph import PretrainStepGraph
from sar_ssl_amd import _lib
g = PretrainStepGraph(net, flat, lr=1e-3)
for _ in range(3): g.step(pcm=pcm, static=True)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); g.step(pcm=pcm, static=True); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
print("graph replay: host %.3f ms per step with an idle GPU (min %.3f)" % (1e3 * sum(ts) / len(ts), 1e3 * min(ts)))
t0 = time.perf_counter()
for _ in range(N): g.step(pcm=pcm, static=True)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("graph replay back to back: host %.2f ms/step (includes waiting for the previous replay), total wall %.2f ms/step" % (1e3 * t_host / N, 1e3 * t_all / N))
n0 = _lib.ncalls; step(); print("eager step: %d C-ABI calls" % (_lib.ncalls - n0))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
