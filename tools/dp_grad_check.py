"""Data-parallel gradient check on the REAL model (run under torchrun, 2 ranks; gloo when both ranks share one GPU, RCCL when
each has its own): every rank runs forward/backward on its half of a batch through the overlapped bucketed all-reduce
(dist.FlatGradAllReduce, backward-stage hooks, optional side stream), then rank 0 recomputes the gradient of the whole batch in
one process and compares.  BatchNorm is put in eval mode so the per-rank statistics do not enter (the model is then per-sample
independent and the mean of the per-rank gradients equals the full-batch gradient).  Prints one JSON line on rank 0."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import numpy as np
import torch
from sar_ssl_amd import dist as sdist, hip, model, runtime


def main():
    rank, world, local = sdist.init_from_env()
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    runtime.set_precision(os.environ.get("DPCHECK_PRECISION", "fp32"))
    T, Bper = 16, 2
    torch.manual_seed(7)
    net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev).to(dev)
    net.eval()
    flat = runtime.FlatParams(net)
    sdist.broadcast_parameters(flat)
    red = sdist.FlatGradAllReduce(net, flat)
    g = np.random.default_rng(11)
    sig = torch.from_numpy(g.standard_normal((world * Bper, 512 + 256 * (T - 1), 2)).astype(np.float32)).to(dev)
    idx = np.stack([np.sort(g.choice(T, T // 2, replace=False)) for _ in range(world * Bper)])
    ch = g.integers(0, 2, size=world * Bper)

    def grad_of(rows):
        flat.zero_grad()
        x = hip.stft_frontend(sig[rows])
        net.set_masks(idx[rows], ch[rows])
        loss, _, _ = net(x)
        loss.backward()
        return loss

    rows = slice(rank * Bper, (rank + 1) * Bper)
    grad_of(rows)
    scale = red.finish()
    got = (flat.grad * scale).clone()
    order = list(red.order)
    # single-process reference on the whole batch (hooks still fire; world > 1 all-reduces are avoided by a detached reducer)
    net.set_backward_stage_hook(None)
    grad_of(slice(0, world * Bper))
    ref = flat.grad.clone()
    err = float((got - ref).abs().max() / ref.abs().max())
    spans = {k: [int(a), int(b)] for k, (a, b) in red.spans.items()}
    worst = {k: float((got[a:b] - ref[a:b]).abs().max() / ref.abs().max()) for k, (a, b) in spans.items()}
    out = torch.tensor([err], dtype=torch.float64)
    if world > 1:
        gathered = [torch.zeros_like(out) for _ in range(world)]
        torch.distributed.all_gather(gathered, out.to(dev) if torch.distributed.get_backend() == "nccl" else out)
        err = max(float(t[0]) for t in gathered)
    if rank == 0:
        print(json.dumps({"world": world, "backend": torch.distributed.get_backend() if world > 1 else None, "max_rel_err": err,
                          "per_bucket": worst, "hook_order": order, "two_streams": os.environ.get("SARSSL_TWO_STREAMS", "1")}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
