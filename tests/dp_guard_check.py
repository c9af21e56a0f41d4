"""Data-parallel skip-on-overflow must be a COLLECTIVE decision (advisor finding, round 5).  Run under torch.distributed.run with 2 ranks
(gloo when both share one GPU).  Three fp16-forward training steps through graph.PretrainStepGraph (captured, or launch by launch with
DPGUARD_FORM=eager); in the second one ONLY RANK 1 gets an input that does not fit fp16 (its loss is NaN, rank 0's is finite).  Every
rank must skip that step - parameters and Adam moments untouched on both - and the replicas must stay bit-identical throughout; a
guard on the local loss would let rank 0 apply an update averaged over rank 1's garbage gradient.  Prints one JSON line on rank 0."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sarssl_boot  # noqa
import numpy as np
import torch
import torch.distributed as dist
from sar_ssl_amd import dist as sdist, hip, model, runtime, synth
from sar_ssl_amd.graph import PretrainStepGraph


def main():
    rank, world, local = sdist.init_from_env()
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    runtime.set_precision(os.environ.get("DPGUARD_PRECISION", "fp16"))
    eager = os.environ.get("DPGUARD_FORM", "captured") == "eager"
    T, B = 16, 4
    torch.manual_seed(5)
    net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev).to(dev).train()
    flat = runtime.FlatParams(net)
    sdist.broadcast_parameters(flat)
    red = sdist.FlatGradAllReduce(net, flat)
    g = PretrainStepGraph(net, flat, red, lr=1e-3)
    random.seed(100 + rank)
    runtime.RT.manual_seed(100 + rank)
    sig = torch.from_numpy(synth.make_batch(40 + rank, 3 * B, nsample=512 + 256 * (T - 1))).to(dev)
    xs = [hip.stft_frontend(sig[i * B:(i + 1) * B]) for i in range(3)]
    if rank == 1:
        xs[1] = xs[1].clone()
        xs[1][0, 0, 3, 2, 0] = 1.0e6                       # does not fit fp16: the input-encoding launch raises the context's overflow word
    losses, snaps = [], []
    for x in xs:
        out = g.step_eager(x=x) if eager else g.step(x=x)
        torch.cuda.synchronize()
        losses.append(float(out[0]))
        snaps.append((flat.flat.clone(), g.m.clone(), g.v.clone()))
    skipped = g.skipped_steps()
    # replicas bit-identical after every step?
    same = []
    for p, m, v in snaps:
        ok = True
        for t in (p, m, v):
            other = t.clone()
            dist.broadcast(other, src=0)
            ok = ok and bool(torch.equal(other, t))
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same.append(bool(flag.item() > 0))
    untouched = all(bool(torch.equal(a, b)) for a, b in zip(snaps[0], snaps[1]))
    moved = not torch.equal(snaps[1][0], snaps[2][0])
    sk = torch.tensor([float(skipped)], device=dev)
    lst = [torch.zeros_like(sk) for _ in range(world)]
    dist.all_gather(lst, sk)
    ls = torch.tensor(losses, dtype=torch.float64, device=dev)
    lall = [torch.zeros_like(ls) for _ in range(world)]
    dist.all_gather(lall, ls)
    if rank == 0:
        print(json.dumps({"world": world, "form": "eager" if eager else "captured", "replicas_identical_after_each_step": same,
                          "step2_left_parameters_and_moments_untouched_on_rank0": untouched, "step3_moved_parameters": moved,
                          "skipped_per_rank": [int(t.item()) for t in lst],
                          "losses_per_rank": [[None if not np.isfinite(v) else float(v) for v in t.cpu().tolist()] for t in lall]}), flush=True)
    red.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
