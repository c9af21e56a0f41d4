"""Training loop ("learner") with the reference's interface (code/learner.py: Learner :13, STFTLearner :488).

What changed underneath: the STFT front-end is one fused HIP kernel pair, the model step is the hand-written HIP
forward/backward, Adam is one fused kernel over the flat parameter buffer, multi-GPU is one process per GPU with bucketed
RCCL all-reduce (dist.py) instead of DataParallel, and ``--use-amp`` selects the 'hybrid' numeric mode (fp16 CNN stem + f32 residual
stream on fp16-pair matrix-core products, bf16 gradients: no GradScaler needed; runtime.set_precision) instead of fp16 autocast.
"""
import os
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import hip, runtime, dist as sdist
from .common import utils_module as at_module


def _flag_last(iterable):
    """(item, is_last) for every item of ``iterable`` (one item of look-ahead)."""
    it = iter(iterable)
    try:
        prev = next(it)
    except StopIteration:
        return
    for cur in it:
        yield prev, False
        prev = cur
    yield prev, True


class Learner(ABC):
    def __init__(self, model):
        self.model = model
        self.max_score = -np.inf
        self.early_stop_counter = 0
        self.use_amp = False
        self.start_epoch = 1
        self.device = "cpu"
        self._flat = None
        self._reducer = None
        runtime.set_precision("fp32")            # reference default is fp32 (code/learner.py:100-103); .amp() switches to the 16-bit modes
        super().__init__()

    # ---- device / precision plumbing -------------------------------------------------------------------------------
    def mul_gpu(self):
        """Data parallel: one process per GPU.  Requires a torchrun-style launch (RANK / WORLD_SIZE / LOCAL_RANK)."""
        sdist.init_from_env()
        self._want_dp = True

    def cuda(self):
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if sdist.world_size() > 1:
            torch.cuda.set_device(local)
        self.model.cuda()
        self.device = "cuda"
        self._flat = runtime.FlatParams(self.model)
        self.__dict__.pop("_step_graph", None)           # a captured step holds the OLD flat buffers' addresses: never reuse it
        if sdist.world_size() > 1:
            sdist.broadcast_parameters(self._flat)
        if hasattr(self.model, "set_backward_stage_hook"):
            self._reducer = sdist.FlatGradAllReduce(self.model, self._flat)

    def cpu(self):
        raise hip._lib.SarsslHipError("the sar_ssl_amd learner runs on the GPU only: there is no CPU fallback "
                                      "(use the oracle in oracle/ for CPU reference numbers)")

    def amp(self, dtype=None):
        """Mixed precision (code/learner.py:46-50: the reference autocasts to fp16 with a GradScaler).  Here, by ``dtype`` or
        SARSSL_AMP_DTYPE: 'hybrid' (default since round 6) = fp16 CNN stem and module-internal tensors, f32 residual stream through the
        Conformer blocks / decoder with f32 activations and weights contracted as fp16 pairs - the mode that meets the 1e-3 per-bin
        tolerance against the reference's f32 path; 'fp16' = fp16 forward (storage and MFMA operands, the reference's autocast dtype;
        ~12 % faster, 1.2e-3 per bin); 'bf16' = bf16 throughout.  All with a bf16 backward pass (gradients keep their range: no loss
        scaling) and f32 accumulation."""
        self.use_amp = True
        runtime.set_precision(dtype or os.environ.get("SARSSL_AMP_DTYPE", "hybrid"))

    @abstractmethod
    def data_preprocess(self, mic_sig_batch=None, gt_batch=None):
        pass

    # ---- pretraining (code/learner.py:76-167) ------------------------------------------------------------------------
    def pretrain_epoch(self, dataset, lr=0.0001, epoch=None, return_diff=True):
        self.model.train()
        if self._use_step_graph():
            return self._pretrain_epoch_graph(dataset, lr, return_diff)
        optimizer = runtime.FusedAdam(self._flat, lr=float(lr), betas=(0.9, 0.999))        # re-created every epoch (learner.py:83)
        optimizer.zero_grad()
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        n = 0
        vis_batch = None
        for batch, last in _flag_last(dataset):
            mic_sig_batch = batch[0] if isinstance(batch, (list, tuple)) else batch
            in_batch, = self.data_preprocess(mic_sig_batch, None)
            if last:        # the batch whose vis is returned (code/learner.py:131): full decoder, so vis["pred"] is this very step's prediction
                self.model.__dict__["_full_pred_once"] = True
            loss_batch, diff_batch, vis_batch = self.model(in_batch)
            loss_batch.backward()
            # (16-bit modes: a non-finite loss - an fp16 forward that overflowed - skips the update on the device, like GradScaler.step;
            #  data parallel: the guard is the sum of the ranks' losses, exchanged with the buckets - every replica takes the same decision)
            guard = loss_batch.detach().clone().reshape(1) if (self.use_amp and loss_batch.dtype == torch.float32) else None
            gscale = self._reducer.finish(guard=guard) if self._reducer is not None else 1.0
            optimizer.step(grad_scale=gscale, guard=guard)
            optimizer.zero_grad()
            if self.use_amp:                                                                # a skipped step is not part of the epoch mean
                ok = torch.isfinite(guard[0]) if guard is not None else torch.isfinite(loss_batch.detach())
                acc[0] += torch.where(ok, loss_batch.detach().double(), acc.new_zeros(()))
                acc[1] += torch.where(ok, diff_batch.detach().double(), acc.new_zeros(()))
            else:
                acc[0] += loss_batch.detach().double()                                       # no per-step .item() sync
                acc[1] += diff_batch.detach().double()
            n += 1
        nskip = int(optimizer.nskipped.item()) if optimizer.nskipped is not None else 0
        self._report_skipped(nskip, n)
        return self._epoch_means(acc, n - nskip, vis_batch, return_diff)

    def _report_skipped(self, nskip, n):
        self.skipped_steps_last_epoch = nskip
        if nskip:
            import warnings
            warnings.warn("%d of %d training steps had a non-finite loss (fp16 forward overflow) and were skipped - parameters and Adam "
                          "moments untouched, as torch.cuda.amp.GradScaler does for the reference (code/learner.py:105-108); "
                          "`--use-amp` with SARSSL_AMP_DTYPE=bf16, or no `--use-amp`, has the range for such input" % (nskip, n))

    def _epoch_means(self, acc, n, vis_batch, return_diff):
        acc = acc / max(n, 1)
        if sdist.world_size() > 1:
            torch.distributed.all_reduce(acc)
            acc /= sdist.world_size()
        loss, diff = float(acc[0]), float(acc[1])
        return (loss, diff, vis_batch) if return_diff else loss

    def _use_step_graph(self):
        """The captured step (graph.py) is the default training step on one GPU; SARSSL_GRAPH=0, replayed dropout masks (parity
        fixtures that need the reference's draw order), models without the pretraining node and models with frozen parameters (the
        captured Adam pass updates the whole flat buffer) fall back to the launch-by-launch step.  Data parallel (world > 1): the
        segmented replay - four graph launches with the bucket all-reduces between them - is the default too since round 5 (bit-equal
        to the launch-by-launch step at world 2 and 4, tests/test_gpu_graph.py; the eager step leaves the host only 25 % of margin per
        rank with eight ranks on one host); SARSSL_GRAPH=0 opts out."""
        return (os.environ.get("SARSSL_GRAPH", "1") != "0" and runtime.RT.replay is None and getattr(self.model, "pretrain", False)
                and self._flat is not None and self._flat.on_gpu and all(p.requires_grad for p in self._flat.params))

    def _graph_takes_raw_batch(self, sig):
        """The captured step can run the STFT front-end itself (as bench.py's does) when the batch is the plain 2-microphone case
        of the default front-end: no per-step STFT launch + 67 MB copy into the graph's input buffer, only the 17 MB int16 /
        34 MB f32 batch."""
        # a subclass that overrides data_preprocess (augmentation, another eps / normalisation) must get ITS front-end in the captured
        # step too: the raw-batch path is only the default front-end's (advisor, round 3)
        if getattr(type(self), "data_preprocess", None) is not STFTLearner.data_preprocess:
            return False
        return (torch.is_tensor(sig) and sig.dim() == 3 and sig.shape[2] == 2 and sig.dtype in (torch.float32, torch.int16)
                and getattr(self, "ch_mode", None) == "M" and getattr(self, "win_len", None) == 512 and getattr(self, "nfft", None) == 512
                and getattr(self, "win_shift_ratio", None) == 0.5)

    def _pretrain_epoch_graph(self, dataset, lr, return_diff):
        """Same epoch with every full-size batch replayed from the captured HIP graph(s); batches of another shape (a ragged last
        batch) take the same step eagerly and share the optimizer state."""
        from .graph import PretrainStepGraph
        g = self.__dict__.get("_step_graph")
        if g is not None and g.flat is not self._flat:                                      # (flat buffers rebuilt since the capture)
            g = None
        if g is None:
            g = self.__dict__["_step_graph"] = PretrainStepGraph(self.model, self._flat, self._reducer, lr=float(lr), betas=(0.9, 0.999))
        g.reset_epoch(float(lr))                                                            # "Adam re-created every epoch" (learner.py:83)
        skipped0 = g.skipped_steps()
        self._flat.grad.zero_()
        for batch, last in _flag_last(dataset):
            mic_sig_batch = batch[0] if isinstance(batch, (list, tuple)) else batch
            if not torch.is_tensor(mic_sig_batch):
                mic_sig_batch = torch.as_tensor(mic_sig_batch)
            # the batch whose vis is returned (code/learner.py:131) runs the FULL-prediction variant of the captured step (decoder and block
            # tails on every frame: vis["pred"] = this very step's prediction) - a second set of graphs captured at its first use (round 6;
            # round 5 took this batch launch by launch: ~450 host launches once per epoch); a batch of another shape takes the step eagerly
            if self._graph_takes_raw_batch(mic_sig_batch):                                  # STFT front-end inside the replay
                sig = mic_sig_batch.to(self.device, non_blocking=True).contiguous()
                if g.matches(pcm=sig):
                    g.step(pcm=sig, full_pred=last)
                else:
                    if last:
                        self.model.__dict__["_full_pred_once"] = True
                    g.step_eager(pcm=sig)
                continue
            in_batch, = self.data_preprocess(mic_sig_batch, None)
            in_batch = in_batch.contiguous().float()
            if g.matches(x=in_batch):
                g.step(x=in_batch, full_pred=last)
            else:
                if last:
                    self.model.__dict__["_full_pred_once"] = True
                g.step_eager(x=in_batch)
        vis_batch = g.vis() if g.nsteps else None
        nskip = g.skipped_steps() - skipped0
        self._report_skipped(nskip, g.nsteps)
        return self._epoch_means(g.acc.clone(), g.nsteps - nskip, vis_batch, return_diff)

    def pretest_epoch(self, dataset, return_diff=True, return_eval=False):
        self.model.eval()
        with torch.no_grad():
            acc = torch.zeros(2, dtype=torch.float64, device=self.device)
            n = 0
            vis_batch = None
            for data in dataset:
                in_batch, = self.data_preprocess(data[0], None)
                loss_batch, diff_batch, vis_batch = self.model(in_batch)                     # random masking stays active in eval
                acc[0] += loss_batch.double()
                acc[1] += diff_batch.double()
                n += 1
            acc = acc / max(n, 1)
            loss, diff = float(acc[0]), float(acc[1])
        if return_diff and return_eval:                                                          # last batch only, as in the reference
            result_batch = self.pretrain_evaluate(pred_batch=vis_batch["pred"], gt_batch=vis_batch["tar"], mask_batch=vis_batch["mask"])
            return loss, diff, vis_batch, result_batch
        return (loss, diff, vis_batch) if return_diff else loss

    # ---- supervised fine-tuning / evaluation (code/learner.py:168-269) ------------------------------------------------
    def train_epoch(self, dataset, lr=0.0001, epoch=None, return_metric=False):
        """One epoch of downstream training: Adam re-created per epoch (learner.py:176), per batch preprocess -> model ->
        ``self.loss`` -> backward -> step.  Loss / metric are accumulated on the device (one sync per epoch, not per step)."""
        self.model.train()
        optimizer = runtime.FusedAdam(self._flat, lr=float(lr), betas=(0.9, 0.999))
        optimizer.zero_grad()
        frozen = self._flat.frozen_ranges()
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        n = 0
        world = sdist.world_size()
        for mic_sig_batch, gt_batch in dataset:
            in_batch, gt_batch = self.data_preprocess(mic_sig_batch, gt_batch)
            pred_batch, embed_batch = self.model(in_batch)
            loss_batch = self.loss(pred_batch=pred_batch, gt_batch=gt_batch)
            loss_batch.backward()
            if world > 1:
                torch.distributed.all_reduce(self._flat.grad)
            self._flat.zero_frozen_grads(frozen)
            optimizer.step(grad_scale=1.0 / world)
            optimizer.zero_grad()
            acc[0] += loss_batch.detach().double()
            if return_metric:
                acc[1] += self.evaluate(pred_batch=pred_batch, gt_batch=gt_batch).double()
            n += 1
        acc = acc / max(n, 1)
        if world > 1:
            torch.distributed.all_reduce(acc)
            acc /= world
        loss, metric = float(acc[0]), acc[1].float().cpu()
        if self.use_amp and runtime.RT.dtype == torch.float16 and hip.fp16_overflow(clear=True):
            # (the downstream path has no loss launch that reads the context's fp16-overflow word: report it once per epoch)
            import warnings
            warnings.warn("network input outside fp16's range in this epoch (spectrum / (mean|X_0| + eps) > 65 504: a near-silent reference "
                          "microphone?) - the fp16 forward clipped it; SARSSL_AMP_DTYPE=bf16 or no --use-amp has the range")
        return (loss, metric) if return_metric else loss

    def test_epoch(self, dataset, return_metric=False, return_vis=False):
        self.model.eval()
        with torch.no_grad():
            acc = torch.zeros(2, dtype=torch.float64, device=self.device)
            embed, gt = [], []
            n = 0
            for mic_sig_batch, gt_batch in dataset:
                in_batch, gt_batch = self.data_preprocess(mic_sig_batch, gt_batch)
                pred_batch, embed_batch = self.model(in_batch)
                acc[0] += self.loss(pred_batch=pred_batch, gt_batch=gt_batch).double()
                if return_metric:
                    acc[1] += self.evaluate(pred_batch=pred_batch, gt_batch=gt_batch).double()
                if return_vis:
                    embed += [embed_batch]
                    gt += [gt_batch]
                n += 1
            acc = acc / max(n, 1)
            loss, metric = float(acc[0]), acc[1].float().cpu()
            out = (loss,) + ((metric,) if return_metric else ())
            if return_vis:
                out += ({"embed": torch.cat(embed, dim=0), "label": torch.cat(gt, dim=0)},)
            return out if len(out) > 1 else loss

    def smooth_data(self, data_list, alpha=0.8):
        """current_smooth = alpha * previous_smooth + (1 - alpha) * current (code/learner.py:271-281)."""
        out, cur = [data_list[0]], data_list[0]
        for v in data_list[1:]:
            cur = alpha * cur + (1 - alpha) * v
            out.append(cur)
        return out

    def ensembling(self, checkpoints_dir, epochs):
        """Average the saved models of ``epochs`` into the live model and write ensemble_model.tar (code/learner.py:302-331)."""
        avg = {}
        for i, epoch in enumerate(epochs):
            path = checkpoints_dir + "/model" + str(epoch) + ".tar"
            assert os.path.exists(path), f"{path} does not exist, can not load best model."
            sd = torch.load(path, map_location="cpu", weights_only=False)["model"]
            for k, v in sd.items():
                avg[k] = v * 1 / len(epochs) if i == 0 else avg[k] + v * 1 / len(epochs)
        own = self.model.state_dict()
        self.model.load_state_dict({k: v.to(own[k].dtype) for k, v in avg.items()})
        if self._flat is not None:
            self._flat._synced = None
            self._flat.ensure_shadow()
        if int(os.environ.get("RANK", "0")) == 0:
            torch.save({"epoch": epochs, "model": self._state_dict_cpu()}, checkpoints_dir + "/ensemble_model.tar")

    def remove_checkpoint_epochs(self, checkpoints_dir, epochs):
        """Remove the per-epoch checkpoints of ``epochs`` (code/learner.py:481-486)."""
        if int(os.environ.get("RANK", "0")) != 0:
            return
        for epoch in epochs:
            os.remove(checkpoints_dir + "/model" + str(epoch) + ".tar")

    # ---- early stopping / checkpoints (code/learner.py:283-300, 333-448) ------------------------------------------------
    def early_stopping(self, current_score, patience=5):
        if current_score >= self.max_score:
            self.max_score = current_score
            self.early_stop_counter = 0
            return False, True
        self.early_stop_counter += 1
        return self.early_stop_counter >= patience, False

    def is_best_epoch(self, current_score):
        if current_score >= self.max_score:
            self.max_score = current_score
            return True
        return False

    def _state_dict_cpu(self):
        return {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}

    def save_checkpoint(self, epoch, checkpoints_dir, is_best_epoch=False, save_extra_hist=False):
        """Same file names and dict layout as the reference: {'epoch', 'max_score', 'model'} (no optimizer state)."""
        if int(os.environ.get("RANK", "0")) != 0:
            return
        state_dict = {"epoch": epoch, "max_score": self.max_score, "model": self._state_dict_cpu()}
        torch.save(state_dict, checkpoints_dir + "/latest_model.tar")
        if save_extra_hist:
            torch.save(state_dict, checkpoints_dir + "/model" + str(epoch) + ".tar")
        if is_best_epoch:
            torch.save(state_dict, checkpoints_dir + "/best_model.tar")

    def _load_model_state(self, sd, as_all_state=True, ex_key=""):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}   # DataParallel prefix (Q13)
        if as_all_state:
            self.model.load_state_dict(sd)
        else:
            all_state_dict = self.model.state_dict()
            match_key_cnt = 0
            for key in sd:
                if ex_key + key in all_state_dict:
                    all_state_dict[ex_key + key] = sd[key]
                    match_key_cnt += 1
            assert match_key_cnt > 1, "loaded model parameters and original parameters unmatched~"
            self.model.load_state_dict(all_state_dict)
        if self._flat is not None:
            self._flat.ensure_shadow()
        self.__dict__.pop("_step_graph", None)           # (requires_grad / buffers may change with a checkpoint: capture again)
        return sd

    def resume_checkpoint(self, checkpoints_dir, from_latest=True, as_all_state=True, ex_key=""):
        model_path = checkpoints_dir + ("/latest_model.tar" if from_latest else "/best_model.tar")
        assert os.path.exists(model_path), f"{model_path} does not exist, can not load latest checkpoint."
        checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
        self.start_epoch = checkpoint["epoch"] + 1
        self.max_score = checkpoint["max_score"]
        self._load_model_state(checkpoint["model"], as_all_state, ex_key)

    def load_checkpoint_best(self, checkpoints_dir, as_all_state=True, param_frozen=False, ex_key=""):
        best_model_path = checkpoints_dir + "/best_model.tar"
        assert os.path.exists(best_model_path), f"{best_model_path} does not exist, can not load best model."
        checkpoint = torch.load(best_model_path, map_location="cpu", weights_only=False)
        sd = self._load_model_state(checkpoint["model"], as_all_state, ex_key)
        if param_frozen:
            for key, value in self.model.named_parameters():
                if key in sd:
                    value.requires_grad = False
        return checkpoint["epoch"]


class STFTLearner(Learner):
    """Learner for models fed with STFTs of the microphone signals (code/learner.py:488-553)."""

    def __init__(self, model, win_len, win_shift_ratio, nfft, fre_used_ratio, fs, mel_scale=False, task=None, ch_mode="M"):
        super().__init__(model)
        if mel_scale or fre_used_ratio != 1 or ch_mode not in ("M", "MM"):
            raise NotImplementedError("only the pretraining front-end (linear frequency, bins 1..nfft/2, ch_mode 'M' | 'MM') is implemented")
        self.ch_mode = ch_mode
        self.win_len, self.win_shift_ratio, self.nfft = win_len, win_shift_ratio, nfft
        self.stft = at_module.STFT(win_len=win_len, win_shift_ratio=win_shift_ratio, nfft=nfft)
        self.istft = at_module.ISTFT(win_len=win_len, win_shift_ratio=win_shift_ratio, nfft=nfft, inv=False)
        self.fre_range_used = range(1, int(nfft / 2 * fre_used_ratio) + 1, 1)
        self.addbatch = at_module.AddChToBatch(ch_mode=self.ch_mode)
        self.task = task

    def data_preprocess(self, mic_sig_batch=None, gt_batch=None, eps=1e-6):
        """mic_sig_batch (nbatch, nsample, nch) f32 or int16 PCM -> [reim (nb*(nch-1), 2, nf, nt, 2)] on the GPU: STFT,
        normalisation by the mean magnitude of mic 0, pairing with mic 0 and DC removal fused in csrc/stft.hip."""
        data = []
        if mic_sig_batch is not None:
            if not torch.is_tensor(mic_sig_batch):
                mic_sig_batch = torch.as_tensor(mic_sig_batch)
            sig = mic_sig_batch.to(self.device, non_blocking=True)
            if sig.dtype not in (torch.float32, torch.int16):
                sig = sig.float()
            data += [hip.stft_frontend(sig, eps=eps, win_len=self.win_len, hop=int(self.win_len * self.win_shift_ratio),
                                       nfft=self.nfft, ch_mode=self.ch_mode)]
        if gt_batch is not None:
            gt = gt_batch[self.task].to(self.device)
            data += [self.get_tar_batch(gt)]
        return data

    def pretrain_evaluate(self, pred_batch, gt_batch, mask_batch):
        """Reconstruction quality of the pretext task (code/learner.py:574-618).  pred/gt (nb,nf,nt,nreim,nch), mask (nb,nf,nt,nch)
        -> {'sig_pred','sig_tar','mse','mse_mask','mse_mask_ch','pesq','pesq_mask_ch'}.  The waveforms come from the HIP inverse
        STFT; PESQ is the third-party ``torchmetrics``/``pesq`` package and is NaN when that package is not installed."""
        def to_sig(v):
            st = torch.view_as_complex(v.permute(0, 1, 2, 4, 3).contiguous().float())          # (nb,nf,nt,nch)
            st = torch.cat((torch.zeros_like(st[:, 0:1]), st), dim=1)                           # DC row back
            sig = self.istft(st)
            return sig / torch.max(sig)
        sig_pred, sig_gt = to_sig(pred_batch), to_sig(gt_batch)
        mask_dense = mask_batch[:, :, :, np.newaxis, :].tile(1, 1, 1, 2, 1)
        diff = (pred_batch - gt_batch) ** 2
        diff_mask = diff * (1 - mask_dense)
        mse_mask = torch.sum(diff_mask) / torch.sum(1 - mask_dense)
        mse_mask_ch = torch.mean(torch.sum(diff_mask, dim=4))
        mse = torch.mean(diff)
        nb, _, _, nch = mask_batch.shape
        pesq = torch.full((nb, nch), float("nan"))
        pesq_mask_ch = torch.full((nb,), float("nan"))
        try:
            from torchmetrics.functional.audio.pesq import perceptual_evaluation_speech_quality as _pesq
        except Exception:
            _pesq = None
        if _pesq is not None:
            for b_idx in range(nb):
                mask_ch_idx = 0 if mask_batch[b_idx, :, :, 1].sum() > mask_batch[b_idx, :, :, 0].sum() else 1
                for ch_idx in range(nch):
                    pesq[b_idx, ch_idx] = _pesq(sig_pred[b_idx, :, ch_idx].cpu(), sig_gt[b_idx, :, ch_idx].cpu(), 16000, "wb")
                pesq_mask_ch = pesq[:, mask_ch_idx]                                            # (sic) last item's channel, learner.py:614
        return {"sig_pred": sig_pred, "sig_tar": sig_gt, "mse": mse, "mse_mask": mse_mask, "mse_mask_ch": mse_mask_ch,
                "pesq": pesq, "pesq_mask_ch": pesq_mask_ch}

    def get_tar_batch(self, gt_batch):
        if self.task == "TDOA":
            return gt_batch[:, np.newaxis] * 16000
        if self.task in ("DRR", "C50", "T60", "ABS"):
            return gt_batch[:, np.newaxis]
        raise Exception("Task mode unrecognized")

    def loss(self, pred_batch, gt_batch):
        return torch.nn.functional.mse_loss(pred_batch.contiguous(), gt_batch.contiguous().detach())

    def evaluate(self, pred_batch, gt_batch):
        return torch.mean(torch.abs(pred_batch.contiguous().detach() - gt_batch.contiguous().detach()))
