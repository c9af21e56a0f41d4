"""Debug: run-to-run noise of the eager gradient vs graph-replay gradient (same masks, same dropout seeds)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import numpy as np, torch
from sar_ssl_amd import hip, model, runtime, synth
from sar_ssl_amd.graph import PretrainStepGraph
RT = runtime.RT
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16")
runtime.set_precision(prec)
T, B = 16, 4
pdrop = float(os.environ.get("PDROP", "0.1"))
nsample = 512 + 256 * (T - 1)
sig = torch.from_numpy(synth.make_batch(3, B, nsample=nsample)).cuda()
x = hip.stft_frontend(sig)
idx = np.tile(np.arange(T // 2)[None, :] * 2, (B, 1)); ch = np.array([1, 0, 1, 0])
torch.manual_seed(9)
net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
for m in net.modules():
    if isinstance(m, torch.nn.Dropout): m.p = pdrop
net.to(dev).train()
flat = runtime.FlatParams(net)
g = PretrainStepGraph(net, flat, lr=0.0)
g.zero_grad_in_adam = False

def spans(d):
    out = {}
    for name, (a, b) in flat.group_spans.items():
        out[name] = float(d[a:b].abs().max())
    return out

def per_param(d, ref):
    worst = []
    for (n, p), o in zip([(n, p) for n, p in net.named_parameters()], [None] * 10000):
        pass
    res = []
    name_of = {id(p): n for n, p in net.named_parameters()}
    for p, o in zip(flat.params, flat.offsets):
        k = p.numel()
        e = float(d[o:o + k].abs().max()); r = float(ref[o:o + k].abs().max())
        res.append((e / (r + 1e-30), e, r, name_of[id(p)]))
    res.sort(reverse=True)
    return res[:8]

net.set_masks(idx, ch)
l_rep = float(g.step(x=x)[0]); g1 = flat.grad.clone(); flat.grad.zero_()
net.set_masks(idx, ch)
l_rep2 = float(g.step(x=x)[0]); g1b = flat.grad.clone(); flat.grad.zero_()      # another replay: different salt
def eager(salted):
    keep = RT._ctr
    if salted: hip.step_state_attach(g.state)
    try:
        RT._ctr = g._seed_ctr0
        g._body(None, g.src, g.idx, g.ch, g.mp, False, with_adam=False)
    finally:
        hip.step_state_attach(None); RT._ctr = keep
    torch.cuda.synchronize()
    r = flat.grad.clone(); flat.grad.zero_(); return float(g.out[0]), r
l2, g2 = eager(True)
l3, g3 = eager(True)
print("loss replay2 %.7f eager-salted %.7f %.7f" % (l_rep2, l2, l3))
gm = float(g2.abs().max())
print("eager vs eager (same salt):   max|d|/max|g| = %.3e" % (float((g2 - g3).abs().max()) / gm))
print("replay vs eager (same salt):  max|d|/max|g| = %.3e" % (float((g1b - g2).abs().max()) / gm))
print("replay1 vs replay2 (other salt): %.3e" % (float((g1 - g1b).abs().max()) / gm))
for r in per_param(g1b - g2, g2): print("   replay-vs-eager", "%.3e %.3e %.3e %s" % r)
for r in per_param(g2 - g3, g2): print("   eager-vs-eager ", "%.3e %.3e %.3e %s" % r)
