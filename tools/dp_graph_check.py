"""Data-parallel check of the captured training step (run under torchrun, 2 ranks; gloo when both ranks share one GPU, RCCL when
each has its own): three Adam steps through graph.PretrainStepGraph - graphs cut at the gradient-bucket boundaries, the bucket
all-reduces issued eagerly between them - against the same three steps through the launch-by-launch data-parallel step
(dist.FlatGradAllReduce hooks + runtime.FusedAdam).  Dropout off, same masks.  Prints one JSON line on rank 0."""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import numpy as np
import torch
from sar_ssl_amd import dist as sdist, hip, model, runtime, synth
from sar_ssl_amd.graph import PretrainStepGraph


def main():
    rank, world, local = sdist.init_from_env()
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    runtime.set_precision(os.environ.get("DPCHECK_PRECISION", "bf16"))
    T, B, nstep, lr = 16, 4, 3, 1e-3
    nsample = 512 + 256 * (T - 1)
    sig = torch.from_numpy(synth.make_batch(100 * rank, nstep * B, nsample=nsample)).to(dev)
    xs = [hip.stft_frontend(sig[i * B:(i + 1) * B]) for i in range(nstep)]

    def build():
        torch.manual_seed(21)
        net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        net.to(dev).train()
        flat = runtime.FlatParams(net)
        sdist.broadcast_parameters(flat)
        return net, flat, sdist.FlatGradAllReduce(net, flat)

    # launch-by-launch
    net, flat, red = build()
    opt = runtime.FusedAdam(flat, lr=lr)
    opt.zero_grad()
    random.seed(5 + rank)
    ref = []
    for x in xs:
        loss, _, _ = net(x)
        loss.backward()
        opt.step(grad_scale=red.finish())
        opt.zero_grad()
        ref.append(float(loss))
    p_ref = flat.flat.clone()
    # captured
    net2, flat2, red2 = build()
    g = PretrainStepGraph(net2, flat2, red2, lr=lr)
    random.seed(5 + rank)
    got = [float(g.step(x=x)[0]) for x in xs]
    plan = [k if k != "reduce" else "reduce:" + str(it) for k, it in g._plan]
    loss_rel = max(abs(a - b) / abs(b) for a, b in zip(got, ref))
    dpar = float((flat2.flat - p_ref).abs().max() / lr)
    # all ranks hold the same parameters after the averaged update
    mine = flat2.flat.clone()
    other = mine.clone()
    if world > 1:
        torch.distributed.broadcast(other, src=0)
    stats = torch.tensor([loss_rel, dpar, float((mine - other).abs().max())], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"world": world, "backend": torch.distributed.get_backend() if world > 1 else None, "plan": plan,
                          "loss_rel": float(stats[0]), "param_lr_units": float(stats[1]), "rank_param_diff": float(stats[2]),
                          "losses": got}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
