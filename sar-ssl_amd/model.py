"""MC-Conformer models with the reference's constructors, forward signatures and state_dict keys
(code/model.py: EmbedEncoder :18, EmbedDecoder :264, SARSSL :350, SARSSL_MultiCH :793, MCConformer :824).

Only the architectures on the pretraining path are built (model=['cnn','conformer'] encoders, ['','fc'] decoder,
frame patches); the ablation backbones of the reference are out of scope and rejected loudly.
All device work runs through the HIP kernels (engine.py); the whole pretrain forward is one autograd node whose
backward is the hand-written backward pass.
"""
import contextlib

import numpy as np
import torch
import torch.nn as nn

from . import engine, hip
from .autograd import tape_apply, pool_mean, head_apply
from .runtime import RT, begin_forward as runtime_begin_forward
from .common import utils_module as at_module
from .common.Conformer import ConformerEncoder


def _check_cnn_conformer(model):
    if list(model) != ["cnn", "conformer"]:
        raise NotImplementedError("only model=['cnn','conformer'] (the MC-Conformer) is implemented, got %r" % (model,))


class EmbedEncoder(nn.Module):
    def __init__(self, sig_shape, patch_shape, dembed, model=["cnn", "conformer"], mode="spat", use_cls=False, device="cpu"):
        super().__init__()
        _check_cnn_conformer(model)
        if use_cls:
            raise NotImplementedError("use_cls is never enabled on the reference's default path")
        self.sig_shape, self.patch_shape, self.dembed, self.device = sig_shape, patch_shape, dembed, device
        self.model, self.use_cls = model, use_cls
        nf, nt, nreim, nmic = sig_shape
        nch = nreim * nmic
        assert nch == 4 and tuple(patch_shape) == (nf, 1), "frame patches of 2 mics x (re, im) only"
        mhsa_nlayer = 1 if mode == "spec" else 3                                              # code/model.py:38-43
        conv_chs = 64
        self.patch_recover = at_module.PatchRecover(output_shape=(nf, nt), patch_shape=self.patch_shape)
        self.patch_embed = nn.Sequential(
            nn.Conv2d(nch, conv_chs, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=False),
            nn.BatchNorm2d(conv_chs),
            nn.ReLU(inplace=True),
            nn.Conv2d(conv_chs, conv_chs, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False),
            nn.BatchNorm2d(conv_chs),
            nn.ReLU(inplace=True),
            nn.Conv2d(conv_chs, conv_chs, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False),
            nn.BatchNorm2d(conv_chs),
            nn.ReLU(inplace=True),
            nn.Conv2d(conv_chs, nch, kernel_size=(1, 1), stride=(1, 1), padding=0, bias=False),
            nn.BatchNorm2d(nch),
            nn.ReLU(inplace=True),
            nn.Conv2d(int(nch), dembed, kernel_size=patch_shape, stride=patch_shape, padding=0, bias=False),
        )
        self.embed = ConformerEncoder(encoder_dim=self.dembed, num_layers=mhsa_nlayer, num_attention_heads=4,
                                      feed_forward_expansion_factor=4)

    # ---- channels-last fused path used by SARSSL / MCConformer
    def _fwd_cl(self, a0, B, T, saved, out=None):
        e = engine.stem_fwd(a0, self.patch_embed, self.training, saved)
        return engine.encoder_fwd(e, self.embed, B, T, self.training, saved, out=out)

    def _bwd_cl(self, dy, saved):
        de = engine.encoder_bwd(dy, self.embed, saved)
        return engine.stem_bwd(engine.patch_bwd(de, self.patch_embed, saved), self.patch_embed, saved)

    def forward(self, embed, add_same_one=False):
        """embed: (nbatch, npatch, dpatch*nch) -> (nbatch, npatch, dembed)."""
        assert not add_same_one
        B, T, dim = embed.shape
        F = self.patch_shape[0]

        def fwd(x, saved):
            a0 = x.view(B, T, F, 4).permute(0, 2, 1, 3).contiguous()                          # (B,F,T,4), model.py:204-206
            return self._fwd_cl(a0, B, T, saved).view(B, T, self.dembed)

        def bwd(dy, saved):
            self._bwd_cl(dy.reshape(B * T, self.dembed), saved)
            return None                                                                       # inputs are data
        return tape_apply(self, fwd, bwd, embed)


class EmbedDecoder(nn.Module):
    def __init__(self, sig_shape, patch_shape, dembed, model=["", "fc"], use_cls=False):
        super().__init__()
        if list(model) != ["", "fc"]:
            raise NotImplementedError("only the ['', 'fc'] decoder of the pretraining path is implemented")
        self.model, self.use_cls = model, use_cls
        nf, nt, nreim, nmic = sig_shape
        self.dpatch = patch_shape[0] * patch_shape[1]
        dembed_out = self.dpatch * nreim * nmic
        dff = dembed_out * 3
        self.proj = nn.Sequential(nn.Linear(dembed, dff), nn.ReLU(), nn.Linear(dff, dembed_out))

    def forward(self, embed, add_same_one=False):
        B, T, d = embed.shape
        return tape_apply(
            self,
            lambda x, saved: engine.decoder_fwd(x.view(B * T, d), self, saved).view(B, T, -1),
            lambda dy, saved: engine.decoder_bwd(dy.reshape(B * T, -1), self, saved).view(B, T, d),
            embed)


class LazyVis(dict):
    """The reference folds mask / prediction / target to (B,F,T,..) every step (code/model.py:595-599, 776-790) although
    they are only consumed on plotting epochs.  Same keys, materialised on first access."""

    def __init__(self, pred, x, mp_u8, mch):
        super().__init__()
        self._src = (pred, x, mp_u8, mch)
        for k in ("mask", "pred", "tar"):
            dict.__setitem__(self, k, None)

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is None:
            pred, x, mp, mch = self._src
            B, _, F, T, _ = x.shape
            if key == "pred":                                  # (B,T,F,2,2) -> (B,F,T,2,2)
                if callable(pred):                             # training step: the decoder ran on the masked frames only (_PretrainFn) -
                    pred = pred()                              # the full prediction is formed now, from the step's decoder input
                v = pred.detach().float().view(B, T, F, 2, 2).permute(0, 2, 1, 3, 4)
            elif key == "tar":                                 # x is (B,mic,F,T,reim) -> (B,F,T,reim,mic)
                v = x.detach().permute(0, 2, 3, 4, 1)
            else:                                              # mask (B,F,T,mic): 0 where (masked frame, masked mic)
                mc = torch.ones((B, 2), device=x.device).scatter_(1, mch.long().view(B, 1), 0.0)
                m = 1.0 - (1.0 - mp.float()).view(B, 1, T, 1) * (1.0 - mc).view(B, 1, 1, 2)
                v = m.expand(B, F, T, 2)
            dict.__setitem__(self, key, v)
        return v


class _PretrainFn(torch.autograd.Function):
    """Whole pretrain forward (masking -> 2 encoders -> decoder -> masked MSE) as one autograd node."""

    @staticmethod
    def forward(ctx, net, x, idx_i32, ch_i32, mp_u8, *params):
        B, _, F, T, _ = x.shape
        saved = []
        runtime_begin_forward(net.parameters() if not params else params)
        # weight-only launches (taps, patch matrices, feed-forward packs, positional projections): already under way on the side stream
        # when the caller issued them in front of its front-end launches (graph.py), otherwise started here, next to the masking pass
        prep = net.__dict__.pop("_prep_event", None) or net.prepare_weights(F, T)
        pre = net.__dict__.pop("_premasked", None)          # (graph.py: the front-end launch already applied the masks to this very x)
        if pre is not None and pre[0] is x and pre[1].dtype == RT.dtype:
            spec_in, spat_in = pre[1], pre[2]
        else:
            spec_in, spat_in = hip.mask_inputs(x, mp_u8, ch_i32, 0, RT.dtype)
        ds, dt_ = net.spec_encoder.dembed, net.spat_encoder.dembed
        # Decoder - and the row-wise tail of each encoder's last block - on the masked frames only: the loss reads the prediction at the
        # masked frames (code/model.py:585-592, 721-747) and these layers treat every frame separately, so a TRAINING step (a backward pass
        # follows) runs them on the gathered rows - half of them - and scatters the input gradient back: exact.  Forward-only calls (eval,
        # no_grad) and `_full_pred_once` (the learner sets it for the batch whose vis it returns) keep every row; otherwise vis["pred"] is
        # formed on request from the tensors the skipped part starts from.
        full_once = net.__dict__.pop("_full_pred_once", False)
        compact = engine._DEC_MASKED and not full_once and not RT.inference and RT.dtype in engine._16 and RT.replay is None
        tail = compact and engine._TAIL_MASKED and net._side_stream(x.device) is not None
        nrow = B * idx_i32.shape[1] if tail else B * T
        # (hybrid: the f32 stream - or, with the compact tails, directly the fp16 pair the decoder's first product contracts: hi | lo halves)
        pair_out = RT.hybrid and tail
        ecat = torch.empty((2 * nrow if pair_out else nrow, ds + dt_), dtype=(torch.float16 if pair_out else torch.float32) if RT.hybrid else RT.dtype,
                           device=x.device)
        ecat_lo = ecat[nrow:] if pair_out else None
        if pair_out:
            ecat = ecat[:nrow]
        osl = (lambda a, b: (ecat[:, a:b], ecat_lo[:, a:b])) if pair_out else (lambda a, b: ecat[:, a:b])
        # The two encoders are independent until the decoder: run them on two HIP streams so one encoder's HBM-bound passes
        # (BatchNorm statistics / backward, LayerNorm, ...) overlap the other's MFMA-bound convolutions and GEMMs.
        saved_spat = []
        side = net._side_stream(x.device)
        main = torch.cuda.current_stream()
        if prep is not True:
            main.wait_event(prep)                          # (the side stream runs the weight-only launches in order anyway)
        if side is not None:
            pass                                           # (the lazy weight refresh already happened in begin_forward, on the main stream)
            side.wait_stream(main)
            # tensors allocated on the main stream but consumed on the side stream: tell the caching allocator, otherwise their
            # memory can be recycled by main-stream allocations while side-stream kernels that read them are still queued
            spat_in.record_stream(side)
            ecat.record_stream(side)
            # the host enqueues the two encoders in alternating chunks so that neither HIP queue sits empty while the other
            # one is being filled (enqueueing ~300 launches takes the host a few ms)
            train = net.training
            spe, spa = net.spec_encoder, net.spat_encoder
            with torch.cuda.stream(side):
                hip.stamp("fwd.spat.stem.begin")
                e_spat = engine.stem_fwd(spat_in, spa.patch_embed, train, saved_spat)
                hip.stamp("fwd.spat.stem.end")
            hip.stamp("fwd.spec.stem.begin")
            e_spec = engine.stem_fwd(spec_in, spe.patch_embed, train, saved)
            hip.stamp("fwd.spec.stem.end")
            nl = len(spa.embed.layers)
            with torch.cuda.stream(side):
                e_spat = engine.block_fwd(e_spat, spa.embed.layers[0], B, T, train, saved_spat, out=osl(ds, ds + dt_) if nl == 1 else None,
                                          next_blk=spa.embed.layers[1] if nl > 1 else None, rows=idx_i32 if (tail and nl == 1) else None)
            assert len(spe.embed.layers) == 1
            engine.block_fwd(e_spec, spe.embed.layers[0], B, T, train, saved, out=osl(0, ds), rows=idx_i32 if tail else None)
            hip.stamp("fwd.spec.block.end")
            with torch.cuda.stream(side):
                for li in range(1, nl):
                    e_spat = engine.block_fwd(e_spat, spa.embed.layers[li], B, T, train, saved_spat,
                                              out=osl(ds, ds + dt_) if li == nl - 1 else None,
                                              next_blk=spa.embed.layers[li + 1] if li + 1 < nl else None,
                                              rows=idx_i32 if (tail and li == nl - 1) else None)
                hip.stamp("fwd.spat.blocks.end")
            main.wait_stream(side)
            hip.stamp("fwd.join")
        else:
            net.spec_encoder._fwd_cl(spec_in, B, T, saved, out=ecat[:, :ds])
            net.spat_encoder._fwd_cl(spat_in, B, T, saved_spat, out=ecat[:, ds:])
        tails_x = (saved[-1][3], saved_spat[-1][3]) if tail else None      # full-row inputs of the two compact tails (vis on request)
        saved.append(saved_spat)
        sink = net.__dict__.get("_loss_sink")            # graph.py: (persistent f32[2], running f64[2] sums) filled by the finalize launch
        ctx.dpred = None
        net.__dict__["_last_ecat"] = None
        if compact:
            ecat_c = (hip.Pair(ecat, ecat_lo) if pair_out else ecat) if tail else hip.gather_rows(ecat, idx_i32, B, T)
            pred = engine.decoder_fwd(ecat_c, net.decoder, saved)                                   # [B * nm, F * 4]
            if net.__dict__.get("_loss_grad_with_forward"):
                out, ctx.dpred = hip.masked_mse_compact(pred, x, idx_i32, ch_i32, sink=sink, with_grad=True)
            else:
                out = hip.masked_mse_compact(pred, x, idx_i32, ch_i32, sink=sink)
            # what a full prediction would start from: the decoder input of every frame - or (compact tails) the inputs of the two tails
            net.__dict__["_last_ecat"] = ecat if not tail else ("tails", tails_x[0], tails_x[1], ds, dt_, net.training)
        else:
            pred = engine.decoder_fwd(ecat, net.decoder, saved)
            if net.__dict__.get("_loss_grad_with_forward"):  # graph.py: backward follows at once with an incoming gradient of exactly 1
                out, ctx.dpred = hip.masked_mse_fwd(pred, x, idx_i32, ch_i32, sink=sink, with_grad=True)
            else:
                out = hip.masked_mse_fwd(pred, x, idx_i32, ch_i32, sink=sink)
        ctx.net, ctx.saved, ctx.aux = net, saved, (pred, x, mp_u8, ch_i32, idx_i32.shape[1], ds, (idx_i32, tail) if compact else None)
        ctx.nparams = len(params)
        ctx.mark_non_differentiable(out, pred)
        return (out[0] if sink is not None else out[0].clone()), out, pred

    @staticmethod
    def backward(ctx, dloss, _dout, _dpred):
        net, saved = ctx.net, ctx.saved
        pred, x, mp_u8, ch_i32, nm, ds, idx_c = ctx.aux
        hip.sums_arena_reset(x.device)
        dpred = getattr(ctx, "dpred", None)
        hip.stamp("bwd.decoder.begin")
        if idx_c is not None:           # compact decoder (forward): gradient rows of the masked frames, scattered back behind the decoder
            idx_c, tail = idx_c
            if dpred is None:
                dpred = hip.masked_mse_bwd_compact(pred, x, idx_c, ch_i32, 1.0, dloss.contiguous().float())
            decat = engine.decoder_bwd(dpred, net.decoder, saved)
            if not tail:            # (compact tails: the encoders' last blocks take the compact gradient and scatter behind their own tails)
                decat = hip.scatter_rows(decat, idx_c, x.shape[0], x.shape[3])
        else:
            if dpred is None:
                dpred = hip.masked_mse_bwd(pred, x, mp_u8, ch_i32, nm, 1.0, dloss.contiguous().float())
            decat = engine.decoder_bwd(dpred, net.decoder, saved)
        hip.stamp("bwd.decoder.end")
        net._after_backward_stage("decoder")
        saved_spat = saved.pop()
        # encoder backward takes column slices of the concatenated decoder-input gradient (row stride 768).  Schedule: all
        # Conformer blocks and the two patch GEMMs first - that completes every sizeable gradient bucket, whose all-reduce then
        # overlaps the long, nearly parameter-free CNN-stem backward (dist.py) - then the two stems, then the 0.6 MB stem bucket.
        side = net._side_stream(x.device)
        main = torch.cuda.current_stream() if side is not None else None
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        if side is not None:
            side.wait_stream(main)
            decat.record_stream(side)
        spe, spa = net.spec_encoder, net.spat_encoder
        nl = len(spa.embed.layers)
        with on_side():                                         # the host enqueues the two streams in alternating chunks, see forward
            hip.stamp("bwd.spat.blocks.begin")
            d_spat = decat[:, ds:]
            for li in range(nl - 1, 0, -1):
                d_spat = engine.block_bwd(d_spat, spa.embed.layers[li], saved_spat)
        d_spec = decat[:, :ds]
        hip.stamp("bwd.spec.block.begin")
        for blk in reversed(spe.embed.layers):
            d_spec = engine.block_bwd(d_spec, blk, saved, first=blk is spe.embed.layers[0])
        hip.stamp("bwd.spec.block.end")
        cut = side is not None and getattr(net, "_cut_mode", False)   # graph capture cut at the bucket boundaries (graph.py): a cut
        with on_side():                                               # needs both streams joined, so the two hooks fire after the join
            d_spat = engine.block_bwd(d_spat, spa.embed.layers[0], saved_spat, first=True)
            dz_spat = engine.patch_bwd(d_spat, spa.patch_embed, saved_spat)
            hip.stamp("bwd.spat.blocks.end")
            if not cut:
                net._after_backward_stage("spat_encoder")      # its bucket is reduced behind the side stream
        dz_spec = engine.patch_bwd(d_spec, spe.patch_embed, saved)
        if cut:
            main.wait_stream(side)
            net._after_backward_stage("spat_encoder")
        net._after_backward_stage("spec_encoder")
        if cut:
            side.wait_stream(main)
        net._after_backward_stage("stem_bwd_begin")            # (not a bucket: a marker for tests / tracing)
        # CUs of the persistent 3x3 gradient launches: the library's default leaves 1/8 of the chip to the other encoder's stream
        # (csrc/conv3x3.hip: conv_cus).  That pays for the spec stem, which starts while the spat stream is still in its three
        # Conformer blocks - and costs 1/8 of the chip for the spat stem, which runs last and alone (same-box A/B, three interleaved
        # rounds: 10.97 -> 10.84 ms with spec 7/8 + spat all, 11.08 ms the other way round).  One stream: every launch runs alone.
        # Data-parallel runs keep 7/8 for both stems: the bucket all-reduces overlap exactly this part of the backward pass, and a
        # collective kernel holding a few CUs would make the persistent launch's last workgroups start late with their full share of
        # tiles (a convolution workgroup owns its CU's whole register file and LDS - nothing co-resides).
        from . import dist as _dist
        all_cus = 1 << 16 if (engine._STEM_LAST_ALL_CUS and not _dist.exchanging()) else 0
        try:        # the override is per host thread (here: autograd's worker) and must not outlive this pass if a launch raises
            hip.conv_cus_override(all_cus if side is None else 0)
            hip.stamp("bwd.spec.stem.begin")
            engine.stem_bwd(dz_spec, spe.patch_embed, saved)
            hip.stamp("bwd.spec.stem.end")
            hip.conv_cus_override(all_cus)
            with on_side():
                hip.stamp("bwd.spat.stem.begin")
                engine.stem_bwd(dz_spat, spa.patch_embed, saved_spat)
                hip.stamp("bwd.spat.stem.end")
        finally:
            hip.conv_cus_override(0)
        if side is not None:
            main.wait_stream(side)
        net._after_backward_stage("stems")
        return (None,) * (5 + ctx.nparams)


class SARSSL(nn.Module):
    def __init__(self, sig_shape=[256, 256, 2, 2], patch_shape=(256, 1), patch_mode="T", nmasked_patch=128 * 1, pretrain=True,
                 use_cls=False, downstream_token="all", downstream_head="mlp", downstream_embed="spec_spat",
                 downstream_dlabel=1, device="cpu", pretrain_frozen_encoder=False):
        super().__init__()
        nf, nt, nreim, nmic = sig_shape
        if tuple(patch_shape) != (nf, 1):
            patch_shape = (nf, 1)                                                             # frame patches only
        npatch_shape = [int(nf / patch_shape[0]), int(nt / patch_shape[1])]
        if nmasked_patch != (npatch_shape[0] * npatch_shape[1] // 2):                         # code/model.py:361-364
            nmasked_patch = npatch_shape[0] * npatch_shape[1] // 2
        if use_cls or pretrain_frozen_encoder:
            raise NotImplementedError("use_cls / pretrain_frozen_encoder are outside the pretraining hot path")
        self.pretrain, self.pretrain_frozen_encoder, self.device, self.use_cls = pretrain, pretrain_frozen_encoder, device, use_cls
        self.sig_shape = list(sig_shape)
        self.patch_split = at_module.PatchSplit(patch_shape=patch_shape, f_first=False)
        self.patch_recover = at_module.PatchRecover(output_shape=(nf, nt), patch_shape=patch_shape, f_first=False)
        spec_dembed, spat_dembed = 256 * 2, 256                                               # code/model.py:378-379
        self.in_ver = "separate"
        self.embed_use4ds = downstream_embed
        self.spec_encoder = EmbedEncoder(sig_shape=sig_shape, patch_shape=patch_shape, dembed=spec_dembed,
                                         model=["cnn", "conformer"], mode="spec", use_cls=use_cls, device=device)
        self.spat_encoder = EmbedEncoder(sig_shape=sig_shape, patch_shape=patch_shape, dembed=spat_dembed,
                                         model=["cnn", "conformer"], mode="spat", use_cls=use_cls, device=device)
        if self.pretrain:
            self.patch_mask = at_module.PatchMask(patch_mode=patch_mode, nmasked_patch=nmasked_patch,
                                                  npatch_shape=npatch_shape, device=device)
            self.decoder = EmbedDecoder(sig_shape=sig_shape, patch_shape=patch_shape, dembed=spec_dembed + spat_dembed,
                                        model=["", "fc"], use_cls=False)
        else:
            dembed_ds = {"spec_spat": spec_dembed + spat_dembed, "spec": spec_dembed, "spat": spat_dembed,
                         "noinfo": spec_dembed}[downstream_embed]
            if downstream_head == "mlp":
                if downstream_dlabel == 1:
                    self.mlp_head = nn.Sequential(nn.LayerNorm(dembed_ds), nn.Linear(dembed_ds, downstream_dlabel))
                else:
                    self.joint_head = nn.Sequential(nn.LayerNorm(dembed_ds), nn.Linear(dembed_ds, dembed_ds), nn.ReLU(),
                                                    nn.Linear(dembed_ds, downstream_dlabel))
            self.downstream_head, self.downstream_dlabel, self.ds_token = downstream_head, downstream_dlabel, downstream_token
        self._stage_hook = None
        self._cut_mode = False
        self._param_list = None
        self._forced_masks = None

    def flat_param_groups(self):
        """Layout of runtime.FlatParams for this model = the data-parallel gradient buckets in the order backward completes them
        last-to-first (dist.py): stems | spec block(s) + spec patch GEMM | spat blocks + spat patch GEMM | decoder."""
        def stem(enc):
            return [p for i, m in enumerate(enc.patch_embed) if i != 12 for p in m.parameters()]

        def body(enc):
            return list(enc.patch_embed[12].parameters()) + list(enc.embed.parameters())
        groups = [("stems", stem(self.spec_encoder) + stem(self.spat_encoder)), ("spec_encoder", body(self.spec_encoder)),
                  ("spat_encoder", body(self.spat_encoder))]
        if self.pretrain:
            groups.append(("decoder", list(self.decoder.parameters())))
        return groups

    def prepare_weights(self, F, T):
        """Issues engine.prepare_step_weights: on the side stream (-> the event the main stream has to wait for before it uses any of the
        results) or, single-stream, in place (-> True).  graph.py calls it in front of the front-end launch and leaves the event in
        ``_prep_event`` for the forward pass."""
        dev = next(self.parameters()).device
        side = self._side_stream(dev) if (engine._PREP_ASYNC and dev.type == "cuda") else None
        if side is None:
            engine.prepare_step_weights(self, F, T, need_bwd=not RT.inference)
            return True
        main = torch.cuda.current_stream()
        side.wait_stream(main)                             # (the previous step's optimizer launch wrote the 16-bit shadows on the main stream)
        with torch.cuda.stream(side):
            engine.prepare_step_weights(self, F, T, need_bwd=not RT.inference)
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def _side_stream(self, device):
        """Second HIP stream for the spat encoder (None disables the two-stream schedule: SARSSL_TWO_STREAMS=0)."""
        import os
        if os.environ.get("SARSSL_TWO_STREAMS", "1") == "0" or RT.replay is not None:       # replayed dropout masks are drawn in the
            return None                                                                    # reference's order: spec encoder first
        s = self.__dict__.get("_side")
        if s is None:
            s = self.__dict__["_side"] = torch.cuda.Stream(device=device)
        return s

    # ---- hooks used by the data-parallel wrapper to start gradient all-reduce while backward continues
    def set_backward_stage_hook(self, fn):
        self._stage_hook = fn

    def _after_backward_stage(self, name):
        if self._stage_hook is not None:
            self._stage_hook(name)

    def set_masks(self, mask_patch_idx, mask_ch_idx):
        """Test hook: use explicit masks for the next forward instead of drawing them from python ``random``."""
        self._forced_masks = (np.asarray(mask_patch_idx, dtype=np.int64), np.asarray(mask_ch_idx, dtype=np.int64).reshape(-1))

    def _masks(self, B, T, dev):
        if self._forced_masks is not None:
            idx, ch = self._forced_masks
            self._forced_masks = None
        else:
            idx, ch = self.patch_mask.sample(B, 2)
        idx = np.sort(np.asarray(idx), axis=1)            # ascending per item: the row order of the compact decoder path (the mask is a set)
        if idx.shape[1] > 1 and not (np.diff(idx, axis=1) > 0).all():
            # the compact loss maps a frame to its row by counting the masked frames below it: a repeated index would silently shift the rows
            # (advisor, round 5) - the reference's PatchMask draws without replacement (utils_module.py:265-267), so this is a caller error
            raise ValueError("masked-frame indices must be distinct per item (got a repeated frame index)")
        mp = np.ones((B, T), dtype=np.uint8)
        np.put_along_axis(mp, idx, 0, axis=1)
        # pinned staging: an H2D copy from pageable memory first drains the stream (the host could never run ahead of the GPU
        # by more than one step); the pinned blocks come from torch's caching host allocator, which keeps them alive until
        # the asynchronous copy has been consumed
        def to(a, dt_):
            h = torch.from_numpy(np.ascontiguousarray(a))
            if dev.type == "cuda":
                h = h.pin_memory()
            return h.to(dev, non_blocking=True).to(dt_)
        return to(idx.astype(np.int32), torch.int32), to(ch.astype(np.int32), torch.int32), to(mp, torch.uint8)

    def forward(self, x):
        if not x.is_cuda:
            raise hip._lib.SarsslHipError("SARSSL runs on the GPU only (no CPU fallback); got a CPU tensor")
        nbatch, nmic, nf, nt, nreim = x.shape
        x = x.contiguous().float()
        if self.pretrain:
            idx, ch, mp = self._masks(nbatch, nt, x.device)
            if self._param_list is None:
                self._param_list = [p for p in self.parameters() if p.requires_grad]
            if torch.is_grad_enabled() and self._param_list:
                loss, out, pred = _PretrainFn.apply(self, x, idx, ch, mp, *self._param_list)
            else:
                RT.inference = True
                try:
                    loss, out, pred = _PretrainFn.forward(_NoCtx(), self, x, idx, ch, mp)
                finally:
                    RT.inference = False
            ecat = self.__dict__.pop("_last_ecat", None)
            if ecat is not None:        # compact training step: vis["pred"] = the decoder on every frame of this step's decoder input, on request,
                pred = _full_pred_fn(ecat, self.decoder, self, step_rows=(pred.detach(), idx, nbatch, nt))      # with the step's own rows at the masked frames
            return loss, out[1], LazyVis(pred, x, mp, ch)
        # ---- downstream branch (code/model.py:667-719): both encoders on the unmasked input, mean over frames, MLP head
        B, T, F = nbatch, nt, nf
        v = x.permute(0, 3, 2, 4, 1).reshape(B, T, F * 4)                                      # (B, npatch, dpatch*nreim*nmic)
        embed_spec = self.spec_encoder(v)
        embed_spat = self.spat_encoder(v)
        if self.embed_use4ds == "spec_spat":
            embed = torch.cat([embed_spec, embed_spat], dim=2)
        elif self.embed_use4ds == "spec":
            embed = embed_spec
        elif self.embed_use4ds == "spat":
            embed = embed_spat
        else:
            embed = torch.zeros_like(embed_spec)
        # pooling + head (a "next" row): mean over the frames, LayerNorm, one or two small Linear layers - through the
        # library like everything else (csrc/head.hip; round 6)
        pooled = pool_mean(embed)
        head = self.mlp_head if self.downstream_dlabel == 1 else self.joint_head
        return head_apply(head, pooled), pooled


def _full_pred_fn(ecat, dec, net=None, step_rows=None):
    """The decoder on every frame of a step's decoder input, in the numeric mode of that step (the caller may have switched modes since).
    ecat = ("tails", x_spec, x_spat, ds, dt, train): the step also ran the tails of the encoders' last blocks on the masked frames only -
    they are run on every row first.  step_rows = (pred_c [B * nm, F * 4], idx [B, nm], B, T): the step's OWN prediction at the masked
    frames (the rows its decoder did run on) is written over the re-formed rows, so vis["pred"] at every frame that entered the loss is
    the step's prediction bit for bit - with dropout on as well (code/model.py:595-599 clones the step's pred); the unmasked frames,
    which no loss ever reads, are a second draw of the row-wise layers' dropout masks (the compact step never computed them)."""
    from . import runtime
    prec = runtime.get_precision()

    def full_pred():
        now, keep = runtime.get_precision(), RT.inference
        runtime.set_precision(prec)
        RT.inference = True
        try:
            with torch.no_grad():
                if isinstance(ecat, tuple):
                    _, xs, xt, ds, dt_, train = ecat
                    full = torch.empty((xs.shape[0], ds + dt_), dtype=xs.dtype, device=xs.device)
                    engine.block_tail_full(xs, net.spec_encoder.embed.layers[-1], train, out=full[:, :ds])
                    engine.block_tail_full(xt, net.spat_encoder.embed.layers[-1], train, out=full[:, ds:])
                    pred = engine.decoder_fwd(full, dec, [])
                else:
                    pred = engine.decoder_fwd(ecat, dec, [])
                if step_rows is not None:          # (pure data movement on a plotting path: torch indexing)
                    pc, idx, B, T = step_rows
                    pred = pred.clone()
                    rows = (torch.arange(B, device=idx.device, dtype=torch.long)[:, None] * T + idx.long()).reshape(-1)
                    pred.view(B * T, -1).index_copy_(0, rows, pc.reshape(rows.numel(), -1).to(pred.dtype))
                return pred
        finally:
            RT.inference = keep
            runtime.set_precision(now)
    return full_pred


class _NoCtx:
    """Stand-in ctx for running _PretrainFn.forward without autograd (eval / no_grad)."""

    def mark_non_differentiable(self, *a):
        pass


class MCConformer(nn.Module):
    """Plain encoder/decoder (code/model.py:824-912): both encoders on the unmasked input, decoder, fold."""

    def __init__(self, sig_shape=[256, 256, 2, 2], patch_shape=(256, 1), spec_model=["cnn", "conformer"],
                 spat_model=["cnn", "conformer"], dembed={"spec": 512, "spat": 256}, dec_model=["", "fc"], device="cpu"):
        super().__init__()
        nf, nt, nreim, nmic = sig_shape
        self.dembed = dembed
        self.patch_split = at_module.PatchSplit(patch_shape=patch_shape, f_first=False)
        self.patch_recover = at_module.PatchRecover(output_shape=(nf, nt), patch_shape=patch_shape, f_first=False)
        assert dembed["spec"] > 0 and dembed["spat"] > 0
        self.spec_encoder = EmbedEncoder(sig_shape=sig_shape, patch_shape=patch_shape, dembed=dembed["spec"], model=spec_model,
                                         mode="spec", use_cls=False, device=device)
        self.spat_encoder = EmbedEncoder(sig_shape=sig_shape, patch_shape=patch_shape, dembed=dembed["spat"], model=spat_model,
                                         mode="spat", use_cls=False, device=device)
        self.decoder = EmbedDecoder(sig_shape=sig_shape, patch_shape=patch_shape, dembed=dembed["spec"] + dembed["spat"],
                                    model=dec_model, use_cls=False)

    def forward(self, x):
        B, nmic, F, T, _ = x.shape
        v = x.permute(0, 3, 2, 4, 1).reshape(B, T, F * 4)
        embed = torch.cat([self.spec_encoder(v), self.spat_encoder(v)], dim=2)
        pred = self.decoder(embed).view(B, T, F, 2, nmic)
        return pred.permute(0, 2, 1, 3, 4)                                                    # (nbatch, nf, nt, nreim, nmic)


class SARSSL_MultiCH(nn.Module):
    """Multi-pair head on top of the single-pair encoder (code/model.py:793-821)."""

    def __init__(self, sig_shape, nmic_pair, task, device):
        super().__init__()
        self.model_sch = SARSSL(sig_shape=sig_shape, pretrain=False, device=device, downstream_token="all", downstream_head="",
                                downstream_embed="spat", downstream_dlabel=1)
        dembed_ds = 256
        factor = nmic_pair if task == "TDOA" else 1
        self.head_mch = nn.Sequential(nn.LayerNorm(dembed_ds * nmic_pair), nn.Linear(dembed_ds * nmic_pair, dembed_ds * nmic_pair),
                                      nn.ReLU(), nn.Linear(dembed_ds * nmic_pair, factor))
        self.nmic_pair = nmic_pair

    def forward(self, x):
        v = x.permute(0, 3, 2, 4, 1).reshape(x.shape[0], x.shape[3], -1)
        embed_sch = pool_mean(self.model_sch.spat_encoder(v))
        embed_sch = embed_sch.reshape(-1, self.nmic_pair * embed_sch.shape[-1])
        return head_apply(self.head_mch, embed_sch), embed_sch
