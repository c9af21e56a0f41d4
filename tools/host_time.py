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
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
