"""Data-parallel pretraining over the GPUs of one node: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process ``torch.nn.DataParallel`` (code/learner.py:25-31).  Gradients live in ONE flat
f32 buffer (runtime.FlatParams) laid out in module order (spec encoder | spat encoder | decoder).  The hand-written
backward finishes those three regions in reverse order and calls a stage hook after each, which immediately issues the
all-reduce of that contiguous slice (22 / 20 / 28 MB) so communication overlaps the remaining backward - above all the long,
almost parameter-free CNN-stem backward that runs last.  No per-step parameter broadcast, ``pe`` buffers are never
communicated, BatchNorm statistics stay per rank exactly like the reference's per-replica behaviour.  The 1/world scaling is
folded into the fused Adam kernel.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style env vars.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SARSSL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def stage_slices(net, flat):
    """[start, end) of each backward stage's parameters inside the flat buffers (decoder / spat_encoder / spec_encoder)."""
    owner = {}
    for name in ("spec_encoder", "spat_encoder", "decoder"):
        mod = getattr(net, name, None)
        if mod is not None:
            for p in mod.parameters():
                owner[id(p)] = name
    spans = {}
    for p, o in zip(flat.params, flat.offsets):
        name = owner.get(id(p), "other")
        end = o + (p.numel() + 7) // 8 * 8
        s = spans.get(name)
        spans[name] = (min(s[0], o), max(s[1], end)) if s else (o, end)
    return spans


class FlatGradAllReduce:
    """Bucketed, overlapped gradient all-reduce driven by the model's backward-stage hooks."""

    def __init__(self, net, flat, process_group=None):
        self.net, self.flat, self.pg = net, flat, process_group
        self.spans = stage_slices(net, flat)
        covered = sorted(self.spans.values())
        assert covered[0][0] == 0 and covered[-1][1] == flat.numel, "stage spans must tile the flat buffer"
        for (a0, a1), (b0, b1) in zip(covered[:-1], covered[1:]):
            assert a1 == b0, "stage spans must be contiguous"
        self.handles = []
        self.world = world_size()
        net.set_backward_stage_hook(self._on_stage)

    def _on_stage(self, name):
        if self.world <= 1 or name not in self.spans:
            return
        s, e = self.spans[name]
        self.handles.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """Wait for all outstanding buckets (also reduces 'other' parameters not owned by a stage).  Returns the gradient
        scale (1/world) to fold into the optimizer step."""
        if self.world > 1 and "other" in self.spans:
            s, e = self.spans["other"]
            self.handles.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles = []
        return 1.0 / self.world


def broadcast_parameters(flat, src=0):
    """One-time parameter sync at start-up (replaces DataParallel's per-step replicate)."""
    if world_size() > 1:
        dist.broadcast(flat.flat, src=src)
        flat._synced = None
        flat.ensure_shadow()
