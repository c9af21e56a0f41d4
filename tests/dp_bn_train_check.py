"""Data-parallel step with TRAINING-mode BatchNorm against an N-replica emulation of the CPU oracle (round-3 verdict, item 6a;
SURVEY.md section 7 "DP vs single-device BN"; reference: code/learner.py:25-31 - every DataParallel replica normalises with the
statistics of ITS part of the batch).  TEST INFRASTRUCTURE (imports oracle/): run under torch.distributed.run with 2 ranks (gloo when both share one GPU,
RCCL when each has its own).  Every rank runs forward / backward on its half of the batch through the overlapped bucketed all-reduce
(dist.FlatGradAllReduce, backward-stage hooks); rank 0 then runs the oracle (fp32, CPU) once per half batch - train mode, dropout 0,
BatchNorm batch statistics of that half - and compares the AVERAGE of the per-half gradients with the all-reduced gradient x 1/world,
per parameter.  A bucket exchanged before its last gradient kernel has finished (e.g. the patch-GEMM weight gradient that is folded
from split-K partials after the Conformer blocks) shows up here; the eval-mode check of tools/dp_grad_check.py cannot see per-rank
statistics at all.  Prints one JSON line on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sarssl_boot  # noqa
import numpy as np
import torch
import recipes
import sarssl_oracle as orc
from sar_ssl_amd import dist as sdist, hip, model, runtime


def main():
    rank, world, local = sdist.init_from_env()
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    runtime.set_precision(os.environ.get("DPCHECK_PRECISION", "fp32"))
    T, Bper = 16, 4
    man = {k: v for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")))["pretrain"].items()}
    net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
    sd = {k: v for k, v in recipes.recipe_state_dict(man, 3).items()}
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.to(dev).train()
    flat = runtime.FlatParams(net)
    sdist.broadcast_parameters(flat)
    red = sdist.FlatGradAllReduce(net, flat)
    g = np.random.default_rng(11)
    sig = torch.from_numpy(g.standard_normal((world * Bper, 512 + 256 * (T - 1), 2)).astype(np.float32))
    idx = np.stack([np.sort(g.choice(T, T // 2, replace=False)) for _ in range(world * Bper)])
    ch = g.integers(0, 2, size=world * Bper)
    rows = slice(rank * Bper, (rank + 1) * Bper)
    flat.zero_grad()
    net.set_masks(idx[rows], ch[rows])
    loss, _, _ = net(hip.stft_frontend(sig[rows].to(dev)))
    loss.backward()
    scale = red.finish()
    torch.cuda.synchronize()
    out = None
    if rank == 0:
        grads = {k: torch.zeros_like(v) for k, v in sd.items() if orc.is_param(k)}
        losses = []
        for r in range(world):
            rr = slice(r * Bper, (r + 1) * Bper)
            osd = {k: v.clone() for k, v in sd.items()}
            params = {k: t.requires_grad_(True) for k, t in osd.items() if orc.is_param(k)}
            l, _, _ = orc.sarssl_pretrain_forward(orc.data_preprocess(sig[rr]), osd, torch.from_numpy(idx[rr]), torch.from_numpy(ch[rr]),
                                                  train=True, p_drop=0.0, return_pred=False)
            l.backward()
            losses.append(float(l))
            for k, p in params.items():
                grads[k] += p.grad / world
        top = max(float(v.abs().max()) for v in grads.values())
        worst, worst_key, num, den = 0.0, None, 0.0, 0.0
        for k, p in net.named_parameters():
            got = (p.grad.detach().float().cpu() * scale).double()
            ref = grads[k].double()
            e = float((got - ref).abs().max()) / top
            num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
            if e > worst:
                worst, worst_key = e, k
        out = {"world": world, "backend": torch.distributed.get_backend() if world > 1 else None, "max_err_over_max_grad": worst,
               "worst_param": worst_key, "rel_l2": (num / den) ** 0.5, "loss_rank0": float(loss), "oracle_loss_rank0": losses[0],
               "hook_order": list(red.order), "precision": runtime.get_precision()}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
