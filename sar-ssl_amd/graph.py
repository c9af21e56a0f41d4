"""One pretraining step (code/learner.py:93-115: data -> model -> loss.backward() -> optimizer.step() -> optimizer.zero_grad())
captured into HIP graphs and replayed.

Why: the eager step enqueues ~450 launches from Python (9 ms of host time against 14 ms of GPU time at batch 64); every kernel-side
gain beyond that would be hidden by the host.  A captured step costs the host one graph launch.

What makes a step replayable although launch arguments are frozen at capture:
  * masks: drawn on the host exactly as before (Python's MT19937, bit-identical to the reference), written into ONE pinned staging
    buffer and copied into fixed device buffers before the replay;
  * dropout: every launch keeps its static seed and adds a device-resident salt that the first node of the graph advances
    (csrc/api.hip, SarsslStepState) - forward and backward of one replay see the same salt, two replays never do;
  * Adam: step count and bias corrections live in the same device state (``sarssl_adam_step_dev``); learning rate / "optimizer
    re-created per epoch" (learner.py:83) are a one-thread reset kernel; the gradient buffer is cleared in the Adam pass;
  * re-laid-out weights (conv taps, patch-GEMM weight, fp8 copies) are rebuilt inside the graph from the shadow weights the Adam
    kernel of the previous replay wrote.

Data parallel (world > 1): the collectives stay OUTSIDE the graphs.  The step is cut into segments at the points where a gradient
bucket becomes final (model._PretrainFn.backward: decoder | Conformer blocks + patch GEMMs | stems); between two segments the
bucket's all-reduce is issued eagerly (RCCL, its own stream) and runs underneath the next segment - four graph launches and four
collectives per step instead of ~450 launches.  With the library's own exchange (SARSSL_NATIVE_RCCL=1: sarssl_allreduce_bucket on a
dedicated communication stream forked and joined by events) the collectives are ordinary stream operations and are captured with
everything else: one graph per step - at world size 1 (the form that has run on hardware) or with SARSSL_NATIVE_RCCL_CAPTURE=1; with
more ranks the native exchange is issued between the segment graphs like torch.distributed's until a captured RCCL kernel has run on a
multi-GPU node.  The segmented replay is the data-parallel default since round 5 (bit-equal to the eager step at world 2 and 4).
"""
import numpy as np
import torch

from . import hip, runtime, _lib
from .runtime import RT


class _Ctx:
    """Stand-in for the autograd ctx: the hand-written forward / backward pair is called directly."""

    def mark_non_differentiable(self, *a):
        pass


class PretrainStepGraph:
    """``g = PretrainStepGraph(net, flat, lr=...)``; ``g.step(x)`` or ``g.step(pcm=...)`` runs one training step and returns the
    device tensor ``[loss, diff]`` (overwritten by the next step).  Captured lazily for the first input shape; inputs of another
    shape raise (the caller falls back to the eager step for ragged tail batches)."""

    def __init__(self, net, flat, reducer=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, two_streams=None):
        assert net.pretrain and flat.on_gpu
        self.net, self.flat, self.reducer = net, flat, reducer
        self.lr, self.betas, self.eps = float(lr), betas, eps
        self.dev = flat.flat.device
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.state = hip.step_state_new(self.dev, RT._seed ^ 0x6A09E667F3BCC909, self.lr, betas)
        self.acc = torch.zeros(2, dtype=torch.float64, device=self.dev)        # running sum of (loss, diff) since reset_epoch()
        self.nsteps = 0
        self._plan = None         # the first captured plan (kept under this name for tests / tools)
        self._plans = {}          # full_pred (bool) -> (plan, pred, xin, vis_masks, ecat): the compact step and, captured on first use, the
                                  # step with the decoder / block tails on EVERY frame (the batch whose vis an epoch returns)
        self._key = None
        self._pool = None
        self._stage = []          # ring of (pinned staging buffer, event) for the masks
        self._stage_i = 0
        self.guard = torch.zeros(1, dtype=torch.float32, device=self.dev)      # data parallel: the ranks' summed loss = the optimizer launch's guard
        self.zero_grad_in_adam = True     # optimizer.zero_grad() folded into the Adam pass (tests switch it off to read the gradient)

    # ------------------------------------------------------------------------------------------------ optimizer interface
    def reset_epoch(self, lr=None):
        """The reference constructs a new Adam at the start of every epoch (learner.py:83): moments and step count restart."""
        if lr is not None:
            self.lr = float(lr)
        self.m.zero_()
        self.v.zero_()
        hip.step_state_reset(self.state, self.lr, self.betas)
        self.acc.zero_()
        self.nsteps = 0

    # ------------------------------------------------------------------------------------------------ masks
    def _mask_layout(self, B, T, nm):
        o_idx, o_ch, o_mp = 0, B * nm * 4, B * nm * 4 + B * 4
        return o_idx, o_ch, o_mp, o_mp + B * T

    def _draw_masks(self, B, T, consume_rng=True):
        net = self.net
        if consume_rng and net._forced_masks is not None:
            idx, ch = net._forced_masks
            net._forced_masks = None
        elif consume_rng:
            idx, ch = net.patch_mask.sample(B, 2)
        else:                                               # capture warm-up: must not advance the python RNG the reference shares
            nm = net.patch_mask.nmasked_patch
            idx = np.tile((np.arange(nm, dtype=np.int64) * max(T // max(nm, 1), 1))[None, :] % T, (B, 1))       # distinct frames (nm <= T)
            ch = np.zeros((B,), dtype=np.int64)
        idx = np.sort(np.asarray(idx), axis=1)                                   # ascending per item (row order of the compact decoder path)
        if idx.shape[1] > 1 and not (np.diff(idx, axis=1) > 0).all():
            raise ValueError("masked-frame indices must be distinct per item (got a repeated frame index)")
        return idx, np.asarray(ch).reshape(-1)

    def _upload_masks(self, idx, ch, B, T):
        nm = idx.shape[1]
        o_idx, o_ch, o_mp, total = self._mask_layout(B, T, nm)
        if not self._stage:
            self._stage = [(torch.empty(total, dtype=torch.uint8).pin_memory(), torch.cuda.Event()) for _ in range(4)]
        buf, ev = self._stage[self._stage_i]
        self._stage_i = (self._stage_i + 1) % len(self._stage)
        ev.synchronize()                                    # the copy issued four steps ago has long been consumed
        h = buf.numpy()
        h[o_idx:o_ch].view(np.int32)[:] = idx.astype(np.int32).reshape(-1)
        h[o_ch:o_mp].view(np.int32)[:] = ch.astype(np.int32)
        mp = h[o_mp:total].reshape(B, T)
        mp[:] = 1
        np.put_along_axis(mp, idx.astype(np.int64), 0, axis=1)
        self.mbuf.copy_(buf, non_blocking=True)
        ev.record()

    # ------------------------------------------------------------------------------------------------ capture
    def _body(self, seg, src, idx, ch, mp, from_pcm, with_adam=True):
        """Enqueues one full step on the current stream(s); ``seg`` receives the cut points (None: plain eager enqueue)."""
        from .model import _PretrainFn
        net = self.net
        if with_adam:
            hip.step_tick(self.state)
        # weight-only launches of the step (taps, patch matrices, feed-forward packs, positional projections): on the side stream, under
        # the front-end launch (the callers have validated the 16-bit shadows)
        F_, T_ = (256, (src.shape[1] - 512) // 256 + 1) if from_pcm else (src.shape[2], src.shape[3])
        net.__dict__["_prep_event"] = net.prepare_weights(F_, T_)
        if from_pcm:        # front-end and input masks in one pass over the spectrum; the forward below picks the masked inputs up
            x, spec_in, spat_in = hip.stft_frontend(src, masks=(mp, ch), dtype=RT.dtype)
            net.__dict__["_premasked"] = (x, spec_in, spat_in)
        else:
            x = src
        ctx = _Ctx()
        net.__dict__["_loss_sink"] = (self.out, self.acc)     # (loss, diff) -> self.out, += self.acc inside the loss's finalize launch
        import os
        if os.environ.get("SARSSL_LOSS_GRAD_FUSED", "1") != "0":     # backward below starts from d(loss) = 1: the loss launch writes dpred too
            net.__dict__["_loss_grad_with_forward"] = True
        try:
            loss, out, pred = _PretrainFn.forward(ctx, net, x, idx, ch, mp)
        finally:
            net.__dict__.pop("_loss_sink", None)
            net.__dict__.pop("_loss_grad_with_forward", None)
            net.__dict__.pop("_premasked", None)
            net.__dict__.pop("_prep_event", None)
        if with_adam and self.reducer is not None and self.reducer.world > 1:
            # data parallel: this rank's loss goes into the guard word that dist.FlatGradAllReduce.finish sums over the ranks in front of
            # the optimizer launch (copied HERE, in the step's first graph segment - the exchange runs between the later segments)
            self.guard.copy_(self.out[:1])
        self.pred, self.xin, self.vis_masks = pred, x, (mp, ch)
        self.ecat = net.__dict__.pop("_last_ecat", None)      # compact decoder path: the decoder's input of every frame (for vis)
        _PretrainFn.backward(ctx, self.one, None, None)
        if not with_adam:
            return
        world = self.reducer.world if self.reducer is not None else 1
        in_graph = self._exchange_in_graph()
        # more than one rank: the guard of the optimizer launch is the SUM of the ranks' losses (exchanged next to the buckets), so that all
        # replicas skip a step together when any rank's forward overflowed (dist.FlatGradAllReduce.finish)
        guard = self.guard if world > 1 else self.out      # (self.guard: copied from the loss right behind the forward pass, see above)
        if seg is not None and not in_graph:
            if self.reducer is not None and (self.reducer.exchange or getattr(self.reducer, "native", None) is not None):
                seg.cut(("finish", None))
        elif self.reducer is not None:                     # (native exchange under capture: the join with the communication stream becomes a graph edge)
            self.reducer.finish(guard=self.guard if world > 1 else None)      # (world 1: closes the step's hook record, see FlatGradAllReduce.strict)
        # guard = the step's loss: a forward that overflowed fp16 (non-finite loss) leaves parameters and moments alone - the reference's
        # GradScaler skips such a step too (code/learner.py:105-108); decided on the device, counted in the step state.  Deliberately
        # stricter than the reference in the modes without a GradScaler counterpart (fp32 / bf16, where the reference would let a NaN
        # propagate into the parameters): the captured step never applies an update computed from a non-finite loss, in any mode; the
        # launch-by-launch learner path guards under --use-amp only, like the reference (advisor, round 5: documented, not aligned)
        hip.adam_step_dev(self.flat.flat, self.flat.grad, self.m, self.v, self.flat.w16, self.state, gscale=1.0 / world, eps=self.eps,
                          zero_grad=self.zero_grad_in_adam, ph16=self.flat.wh16, guard=guard,
                          pl16=self.flat.wl16)      # hybrid mode: the weights' fp16 lo shadow is rewritten by the same pass (None otherwise)
        self.flat._lo_synced = self.flat._synced

    def _exchange_in_graph(self):
        """True when the bucket all-reduces are captured INSIDE the step graph: the library's own exchange (sarssl_allreduce_bucket,
        SARSSL_NATIVE_RCCL=1) - and only at world size 1 unless SARSSL_NATIVE_RCCL_CAPTURE=1 says otherwise.  With more than one rank
        a captured RCCL kernel has never run (no multi-GPU node in five rounds): there the native exchange is issued BETWEEN the
        segment graphs like torch.distributed's, which is the form the 2- and 4-rank tests pin (advisor, round 4)."""
        import os
        red = self.reducer
        if red is None or getattr(red, "native", None) is None:
            return False
        return red.world == 1 or os.environ.get("SARSSL_NATIVE_RCCL_CAPTURE", "0") == "1"

    def step_eager(self, x=None, pcm=None):
        """The same step enqueued launch by launch (inputs whose shape differs from the captured one, e.g. a ragged last batch):
        shares the Adam moments / step count / accumulators with the captured step."""
        assert (x is None) != (pcm is None)
        src = pcm if pcm is not None else x
        if pcm is not None:
            B, T = pcm.shape[0], (pcm.shape[1] - 512) // 256 + 1
        else:
            B, T = x.shape[0], x.shape[3]
        if not hasattr(self, "out"):
            self.out = torch.zeros(2, dtype=torch.float32, device=self.dev)
            self.one = torch.ones((), dtype=torch.float32, device=self.dev)
        self.flat._fresh = False
        self.flat.ensure_shadow()
        idx, ch, mp = self.net._masks(B, T, self.dev)
        self._body(None, src, idx, ch, mp, pcm is not None, with_adam=True)
        runtime.bump_version()
        self.nsteps += 1
        return self.out

    def _capture(self, x, pcm, static=False, full=False):
        """Captures the step for this input shape.  full: the variant whose decoder and block tails run on EVERY frame (vis of the batch an
        epoch returns, code/learner.py:131) - a second set of graphs sharing every buffer of the first (input, masks, loss word, moments,
        step state), captured the first time it is asked for."""
        net, dev = self.net, self.dev
        src = pcm if pcm is not None else x
        if pcm is not None:
            nb, nsample, nch = pcm.shape
            assert nch == 2, "graph step: 2-channel segments ('M' pairing of two mics)"
            B, T = nb, (nsample - 512) // 256 + 1
        else:
            B, T = x.shape[0], x.shape[3]
        if not self._plans:                                 # buffers shared by every captured variant
            self.src = src if static else src.clone()      # input buffer of the graph (static: the caller's tensor is persistent and refilled in place)
            nm = net.patch_mask.nmasked_patch
            o_idx, o_ch, o_mp, total = self._mask_layout(B, T, nm)
            self.mbuf = torch.zeros(total, dtype=torch.uint8, device=dev)
            self.idx = self.mbuf[o_idx:o_ch].view(torch.int32).view(B, nm)
            self.ch = self.mbuf[o_ch:o_mp].view(torch.int32)
            self.mp = self.mbuf[o_mp:total].view(B, T)
            self.out = torch.zeros(2, dtype=torch.float32, device=dev)
            self.one = torch.ones((), dtype=torch.float32, device=dev)
            self.B, self.T = B, T
        elif self.src.data_ptr() != src.data_ptr():
            self.src.copy_(src, non_blocking=True)

        def body(seg, with_adam):
            if full:
                net.__dict__["_full_pred_once"] = True      # (popped by the forward pass: decoder and block tails on every frame)
            self._body(seg, self.src, self.idx, self.ch, self.mp, pcm is not None, with_adam=with_adam)

        # ---- warm-up: one eager pass without side effects (lazy kernel loading, workspaces, allocator) - buffers restored, no Adam
        bufs = [b for b in net.buffers()]
        keep = [b.clone() for b in bufs]
        acc_keep = self.acc.clone()
        hook, net._stage_hook = net._stage_hook, None
        idx, ch = self._draw_masks(B, T, consume_rng=False)
        self._upload_masks(idx, ch, B, T)
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream()
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(cur)
        try:
            with torch.cuda.stream(cap):
                body(None, False)
                cap.synchronize()
                self.flat.grad.zero_()
                for b, k in zip(bufs, keep):
                    b.copy_(k)
                self.acc.copy_(acc_keep)
                runtime.bump_version()            # every re-laid-out weight cache misses during capture: its rebuild becomes graph nodes
                seg = _Segments(self, cap)
                net._stage_hook = seg.on_stage
                # cut mode: the spat / spec hooks fire on the ORIGIN stream after the side stream has joined it.  Needed where the graph is
                # cut at a bucket boundary, and also when the exchange is captured: its communication stream must fork off the origin
                # stream, not off the already-forked side stream (a fork of a fork crashes hipStreamEndCapture on ROCm 7.2,
                # tools/capture_nested_fork_repro.py)
                in_graph = self._exchange_in_graph()
                net._cut_mode = in_graph or (self.reducer is not None and (self.reducer.exchange or getattr(self.reducer, "native", None) is not None))
                seg.in_graph = in_graph
                hip.step_state_attach(self.state)
                try:
                    self._seed_ctr0 = RT._ctr               # (tests: the static dropout seeds of the captured launches)
                    seg.begin()
                    body(seg, True)
                    seg.end()
                finally:
                    hip.step_state_attach(None)
                    net._cut_mode = False
        finally:
            net._stage_hook = hook
        cur.wait_stream(cap)
        torch.cuda.synchronize()
        self._plans[bool(full)] = (seg.plan, self.pred, self.xin, self.vis_masks, self.ecat)
        if self._plan is None:
            self._plan = seg.plan
            self._key = (tuple(src.shape), src.dtype, RT.dtype, RT.fp8, RT.hybrid, net.training)

    # ------------------------------------------------------------------------------------------------ replay
    def matches(self, x=None, pcm=None):
        src = pcm if pcm is not None else x
        return self._key is None or self._key == (tuple(src.shape), src.dtype, RT.dtype, RT.fp8, RT.hybrid, self.net.training)

    def step(self, x=None, pcm=None, static=False, full_pred=False):
        """full_pred: run the decoder and the encoders' last row-wise layers on every frame in this step (the full-prediction variant of
        the captured step) - vis() then returns this very step's prediction at every frame, as the reference does for the batch an epoch
        returns; the default (compact) step computes what the loss reads and forms the rest of vis["pred"] on request."""
        assert (x is None) != (pcm is None)
        assert RT.replay is None, "replayed dropout masks (parity tests) need the eager step"
        src = pcm if pcm is not None else x
        full_pred = bool(full_pred)
        if self._plan is not None and not self.matches(x, pcm):
            raise ValueError("PretrainStepGraph was captured for %r, got %r" % (self._key[0], tuple(src.shape)))
        if full_pred not in self._plans:
            self._capture(x, pcm, static, full=full_pred)
        elif self.src.data_ptr() != src.data_ptr():
            self.src.copy_(src, non_blocking=True)
        plan, self.pred, self.xin, self.vis_masks, self.ecat = self._plans[full_pred]
        self.flat._fresh = False
        self.flat.ensure_shadow()                          # parameters changed through torch since the last step (load_state_dict, ...)
        idx, ch = self._draw_masks(self.B, self.T)
        self._upload_masks(idx, ch, self.B, self.T)
        for kind, item in plan:
            if kind == "graph":
                item.replay()
            elif kind == "reduce":
                self.reducer._on_stage(item)
            else:                                           # "finish": wait for the buckets (stream-side for RCCL) + the collective guard
                self.reducer.finish(guard=self.guard if self.reducer.world > 1 else None)
        runtime.bump_version()                             # weights moved: eager users of the re-laid-out caches must rebuild
        self.nsteps += 1
        return self.out

    def skipped_steps(self):
        """Steps whose update was skipped because the loss was not finite (synchronises)."""
        return hip.step_state_skipped(self.state)

    def vis(self):
        """vis dict of the last step (same keys as SARSSL.forward's third result).  Copies: the graph's pool tensors are overwritten
        by the next replay, a vis dict of the eager path keeps its contents."""
        from .model import LazyVis
        pred = self.pred.clone()
        if getattr(self, "ecat", None) is not None:
            # the step ran its decoder on the masked frames only: the full prediction is formed on request from the step's decoder input
            # (with the decoder as it is NOW - after the step's update; the learner runs the batch whose vis it returns with the full decoder)
            from .model import _full_pred_fn
            e = self.ecat
            e = tuple(t.clone() if torch.is_tensor(t) else t for t in e) if isinstance(e, tuple) else e.clone()
            pred = _full_pred_fn(e, self.net.decoder, self.net, step_rows=(pred, self.idx.clone(), self.B, self.T))     # the step's own rows at the masked frames
        return LazyVis(pred, self.xin.clone(), self.vis_masks[0].clone(), self.vis_masks[1].clone())


class _Segments:
    """Capture as a sequence of graphs sharing one memory pool, cut where an eager action (a collective) has to run in between."""

    def __init__(self, owner, stream):
        self.owner, self.stream = owner, stream
        self.plan = []
        self.g = None
        self.pool = None

    def begin(self):
        self.g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        self.g.capture_begin(pool=self.pool, capture_error_mode="thread_local")      # (loader threads may be making HIP calls)
        self.mark = _lib.ncalls

    def end(self):
        self.g.capture_end()
        self.plan.append(("graph", self.g))
        self.g = None

    def cut(self, action):
        if _lib.ncalls == self.mark and self.plan and self.plan[-1][0] != "graph":
            self.plan.append(action)                       # nothing was enqueued since the previous cut: no empty graph in between
            return
        self.end()
        self.plan.append(action)
        self.begin()

    in_graph = False

    def on_stage(self, name):
        red = self.owner.reducer
        if red is not None and self.in_graph:
            # the library's own RCCL entry point is an ordinary stream operation: the bucket's all-reduce is captured where the hook
            # fires, on the communication stream forked off by an event - ONE graph per step also when data-parallel
            red._on_stage(name)
        elif red is not None and (red.exchange or getattr(red, "native", None) is not None) and name in red.spans:
            self.cut(("reduce", name))
