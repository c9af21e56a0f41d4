"""Data-parallel pretraining over the GPUs of one node: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process ``torch.nn.DataParallel`` (code/learner.py:25-31).  Gradients live in ONE flat
f32 buffer (runtime.FlatParams) laid out by ``SARSSL.flat_param_groups``: stems (0.6 MB) | spec block + patch-GEMM weight (28 MB) |
spat blocks + patch-GEMM weight (20 MB) | decoder (22 MB).  The hand-written backward calls a stage hook as soon as a group's
gradients are final - decoder, then the spat blocks, then the spec block, all BEFORE the long, almost parameter-free CNN-stem
backward starts - and the hook immediately issues the all-reduce of that contiguous slice, so 70 of 70.6 MB are exchanged
underneath the stem backward; only the 0.6 MB stem bucket is reduced after it.  No per-step parameter broadcast, ``pe`` buffers
are never communicated, BatchNorm batch statistics stay per rank exactly like the reference's per-replica behaviour.  The 1/world
scaling is folded into the fused Adam kernel.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style env vars.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        # select this rank's GPU before the process group exists: RCCL binds a communicator to the current device at its first collective
        torch.cuda.set_device(local % torch.cuda.device_count())
    # SARSSL_DIST_FORCE=1: a process group (and every exchange of the data-parallel step) also at world size 1 - how the RCCL code path
    # is exercised on a one-GPU box: communicator set-up, the bucket all-reduce kernels on RCCL's stream under the stem backward, the
    # segmented graph replay with collectives between the graphs; each all-reduce is then a copy
    if (world > 1 or os.environ.get("SARSSL_DIST_FORCE", "0") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SARSSL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def exchanging():
    """True when gradients are exchanged between ranks: more than one rank, or SARSSL_DIST_FORCE=1 with a process group of one."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SARSSL_DIST_FORCE", "0") == "1"


STAGES = ("decoder", "spat_encoder", "spec_encoder", "stems")          # order in which backward completes them


def stage_slices(net, flat):
    """[start, end) of each backward stage's parameters inside the flat buffers.  Uses the layout groups of the flat buffer when
    the model defines them (SARSSL.flat_param_groups); otherwise spans by owning sub-module (spec_encoder / spat_encoder /
    decoder), anything else under 'other'."""
    if getattr(flat, "group_spans", None):
        return dict(flat.group_spans)
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


class _NativeExchange:
    """The bucket all-reduces through the library's own RCCL entry points (sarssl_allreduce_bucket, csrc/comm.hip) instead of
    torch.distributed's NCCL process group: one communicator per rank, created from an id that rank 0 generates and the existing
    process group (any backend - gloo is enough) carries to the others; every bucket is enqueued on ONE dedicated communication stream
    behind an event of the stream that produced its gradients, and finish() makes the compute stream wait for that stream - the same
    overlap torch's async_op gives, with plain HIP events, capturable into the step graph.  Opt-in (SARSSL_NATIVE_RCCL=1)."""

    def __init__(self, device, world, rank, group=None):
        from . import hip
        self.hip = hip
        ids = [hip.comm_unique_id() if rank == 0 else None]
        if world > 1 and dist.is_initialized():
            dist.broadcast_object_list(ids, src=0, group=group)
        self.comm = hip.comm_create(world, rank, ids[0])
        self.stream = torch.cuda.Stream(device=device)
        self.pending = False

    def all_reduce(self, t):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        self.hip.allreduce_bucket(self.comm, t, stream=self.stream)
        self.pending = True

    def wait(self):
        if self.pending:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.pending = False

    def close(self):
        if self.comm:
            self.hip.comm_destroy(self.comm)
            self.comm = None


class FlatGradAllReduce:
    """Bucketed, overlapped gradient all-reduce driven by the model's backward-stage hooks."""

    def __init__(self, net, flat, process_group=None, strict=None, native=None):
        self.net, self.flat, self.pg = net, flat, process_group
        self.spans = stage_slices(net, flat)
        covered = sorted(self.spans.values())
        assert covered[0][0] == 0 and covered[-1][1] == flat.numel, "stage spans must tile the flat buffer"
        for (a0, a1), (b0, b1) in zip(covered[:-1], covered[1:]):
            assert a1 == b0, "stage spans must be contiguous"
        self.handles = []
        self._fired = set()
        self._calls = {}                                  # stage name -> hook calls since the last finish()
        self._closed = True                               # finish() ran: the next hook call starts a new step's record
        self.order = []                                   # stage names in the order their all-reduce was issued in the LAST step (tests)
        self.world = world_size()
        self.exchange = exchanging()                      # (world > 1, or the forced single-rank process group of the readiness runs)
        # strict: the pretraining backward (model._PretrainFn) reports every stage exactly once per step; anything else means a bucket
        # was exchanged before its gradients were final (or twice) - fail loudly instead of training on a wrong average
        self.strict = bool(getattr(net, "pretrain", False)) if strict is None else bool(strict)
        self.nsteps = 0
        # native: the exchange through sarssl_allreduce_bucket (None: SARSSL_NATIVE_RCCL=1 and a GPU buffer; world 1 included, where RCCL's
        # all-reduce is a copy - that is how the path is exercised on a one-GPU box)
        if native is None:
            native = os.environ.get("SARSSL_NATIVE_RCCL", "0") == "1"
        self.native = None
        if native and flat.grad.is_cuda:
            rank = dist.get_rank(process_group) if self.exchange else 0
            self.native = _NativeExchange(flat.grad.device, self.world, rank, process_group)
        net.set_backward_stage_hook(self._on_stage)

    def _on_stage(self, name):
        if name not in self.spans:
            return
        if self._closed:
            self._closed = False
            self.order = []
            self._calls = {}
        self._calls[name] = self._calls.get(name, 0) + 1
        if name in self._fired:
            return
        self.order.append(name)
        if not self.exchange and self.native is None:
            return
        self._fired.add(name)
        s, e = self.spans[name]
        self._reduce(self.flat.grad[s:e])

    def _reduce(self, t):
        if self.native is not None:
            self.native.all_reduce(t)
        else:
            self.handles.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self, guard=None):
        """Wait for all outstanding buckets (also reduces every span no stage hook fired for in this step, e.g. 'other' parameters
        not owned by a stage).  Returns the gradient scale (1/world) to fold into the optimizer step.

        guard (optional, f32 device tensor of one element holding this rank's loss): summed over the ranks IN PLACE next to the buckets,
        so that every rank's guarded optimizer launch sees the same value - a non-finite loss on ANY rank (its fp16 forward overflowed)
        makes the sum non-finite everywhere and all replicas skip the step together.  Deciding on the local loss would let the ranks
        disagree: the one that overflowed skips, the others apply an update averaged over a gradient that contains its garbage, and the
        replicas drift apart for good (parameters are broadcast only once).  The reference's GradScaler decides once, on the gradients
        of its single process (code/learner.py:105-108)."""
        bad = {}
        if self.strict and not self._closed:
            bad = {n: self._calls.get(n, 0) for n in self.spans if n in STAGES and self._calls.get(n, 0) != 1}
        # drain first, complain afterwards: a failed check must not leave collectives in flight or stale per-step counters behind
        try:
            if self.exchange or self.native is not None:
                for name, (s, e) in sorted(self.spans.items(), key=lambda kv: kv[1]):
                    if name not in self._fired:
                        self._reduce(self.flat.grad[s:e])
                if guard is not None and self.world > 1:
                    assert guard.numel() == 1 and guard.dtype == torch.float32
                    self._reduce(guard)
            for h in self.handles:
                h.wait()
            if self.native is not None:
                self.native.wait()
        finally:
            self.handles = []
            self._fired = set()
            self._calls = {}
            self._closed = True
            self.nsteps += 1
        if bad:
            msg = "gradient buckets must be reported exactly once per step, got %r" % (bad,)
            if self.world > 1:                            # nothing is exchanged at world 1 (e.g. a manual extra backward): only say so
                raise AssertionError(msg)
            import warnings
            warnings.warn(msg + " (world size 1: nothing was exchanged)")
        return 1.0 / self.world

    def close(self):
        """Teardown: detach from the model and destroy the library communicator of the native exchange (ncclCommDestroy)."""
        if self.native is not None:
            self.native.wait()
            self.native.close()
            self.native = None
        if getattr(self.net, "_stage_hook", None) == self._on_stage:
            self.net.set_backward_stage_hook(None)

    def __del__(self):
        try:
            if self.native is not None:
                self.native.close()
        except Exception:
            pass

    def describe(self):
        """What a benchmark line needs to prove which exchange ran: backend, library version, ranks, bucket sizes in issue order."""
        info = {"world": self.world, "backend": ("sarssl_allreduce_bucket (RCCL %d)" % self.native.hip.comm_rccl_version() if self.native is not None
                                                 else dist.get_backend(self.pg) if self.exchange else None),
                "buckets": [{"name": n, "bytes": 4 * (self.spans[n][1] - self.spans[n][0])} for n in STAGES if n in self.spans]}
        if self.exchange and info["backend"] == "nccl":
            try:
                info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:                        # (version query is informational only)
                info["rccl_version"] = "unknown (%s)" % type(e).__name__
        return info

    def time_buckets(self, iters=5):
        """Event-timed all-reduce of every bucket's slice on its own (ms per bucket, max over ranks is the caller's business):
        the exchange a step issues, without the backward it normally hides under.  Scratch copies: gradients are not touched."""
        out = {}
        if not self.exchange:
            return out
        for n in STAGES:
            if n not in self.spans:
                continue
            s, e = self.spans[n]
            buf = torch.zeros(e - s, dtype=torch.float32, device=self.flat.grad.device)
            dist.all_reduce(buf, group=self.pg)                       # warm-up (communicator / channel setup)
            if buf.is_cuda:
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(iters):
                    dist.all_reduce(buf, group=self.pg)
                b.record()
                torch.cuda.synchronize()
                out[n] = a.elapsed_time(b) / iters
            else:
                import time
                t0 = time.perf_counter()
                for _ in range(iters):
                    dist.all_reduce(buf, group=self.pg)
                out[n] = 1e3 * (time.perf_counter() - t0) / iters
        return out


def broadcast_parameters(flat, src=0):
    """One-time parameter sync at start-up (replaces DataParallel's per-step replicate)."""
    if world_size() > 1:
        dist.broadcast(flat.flat, src=src)
        flat._synced = None
        flat.ensure_shadow()


def broadcast_buffers(module, src=0):
    """Before validation / checkpointing: every rank takes rank ``src``'s BatchNorm running statistics.  Under the reference's
    ``nn.DataParallel`` the replicas' buffer updates are discarded and only device 0's survive (torch broadcasts module buffers
    from device 0 at every forward), so "rank 0's statistics" IS the reference behaviour; without it the ranks would validate
    different models and could disagree on early stopping."""
    if world_size() <= 1:
        return
    bufs = [b for n, b in module.named_buffers() if n.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    for dtype in (torch.float32, torch.int64):
        sel = [b for b in bufs if b.dtype == dtype]
        if not sel:
            continue
        flat = torch.cat([b.reshape(-1) for b in sel])
        dist.broadcast(flat, src=src)
        o = 0
        for b in sel:
            b.copy_(flat[o:o + b.numel()].view_as(b))
            o += b.numel()


def agree(values, src=0, device=None):
    """Rank ``src``'s python floats on every rank (validation loss -> identical early-stopping / best-epoch decisions)."""
    if world_size() <= 1:
        return list(values)
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.broadcast(t, src=src)
    return [float(v) for v in t.cpu()]
