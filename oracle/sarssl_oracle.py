"""CPU oracle for the SAR-SSL cross-channel-reconstruction pretraining path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU, fp32) *restatement* of the
reference algorithm, written functionally over a state dict that uses the reference's own
``state_dict`` key names.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / reported
baseline - never as the thing shipped.  The product path (``sar-ssl_amd/``) must not import
anything from ``oracle/`` and has no CPU fallback.

Parity pinning: ``oracle/make_golden.py`` (run in the build container, where
``/root/reference`` exists) executes the REAL reference modules on seeded inputs and stores
inputs/outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file
against those vectors, ``tests/test_oracle_vs_reference.py`` checks it live against the
reference when the tree is present.  The reference itself ships no tests or golden vectors
(SURVEY.md section 4), so those fixtures are the pin.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

EPS_BN = 1e-5
EPS_LN = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# Front-end: STFT + normalisation + mic pairing      (code/common/utils_module.py:49-72,
#                                                     code/learner.py:525-553,
#                                                     code/common/utils_module.py:124-148)
# --------------------------------------------------------------------------------------
def stft(signal, win_len=512, win_shift_ratio=0.5, nfft=512):
    """(B, nsample, nch) float32 -> (B, nf=nfft/2+1, nt, nch) complex64.

    Restates ``STFT.forward`` (utils_module.py:49-72): per channel ``torch.stft`` with a
    periodic Hann window, ``center=False``, one-sided, unnormalised.  That is exactly
    framing (hop = win_len*ratio) * window -> rfft.
    """
    hop = int(win_len * win_shift_ratio)
    k = torch.arange(win_len, dtype=torch.float64)
    window = (0.5 - 0.5 * torch.cos(2.0 * math.pi * k / win_len)).to(torch.float32)  # periodic Hann
    frames = signal.permute(0, 2, 1).unfold(-1, win_len, hop)          # (B, nch, nt, win)
    spec = torch.fft.rfft(frames * window, n=nfft, dim=-1)               # (B, nch, nt, nf)
    return spec.permute(0, 3, 2, 1).contiguous()                         # (B, nf, nt, nch)


def data_preprocess(mic_sig, win_len=512, win_shift_ratio=0.5, nfft=512, eps=1e-6, ch_mode="M"):
    """(B, nsample, nch) -> (B*(nch-1), 2, nfft/2, nt, 2) float32.

    Restates ``STFTLearner.data_preprocess`` (learner.py:525-553) with
    ``fre_used_ratio=1`` (bins 1..nfft/2, learner.py:515-516) and ``AddChToBatch('M')``
    (utils_module.py:128-134): scale = mean |X[ch 0]| over all nfft/2+1 bins and frames.
    """
    X = stft(mic_sig, win_len, win_shift_ratio, nfft).permute(0, 3, 1, 2)   # (B, nch, nf, nt)
    mag = X[:, 0:1].abs()
    mean_value = mag.reshape(mag.shape[0], -1).mean(dim=1)
    X = X / (mean_value[:, None, None, None] + eps)
    nb, nch = X.shape[:2]
    if ch_mode == "M":
        ref = X[:, 0:1].expand(nb, nch - 1, *X.shape[2:])
        pairs = torch.stack([ref, X[:, 1:]], dim=2)                       # (B, nch-1, 2, nf, nt)
        X = pairs.reshape(nb * (nch - 1), 2, *X.shape[2:])
    elif ch_mode == "MM":                                                # utils_module.py:136-143
        out = []
        for b in range(nb):
            for c0 in range(nch - 1):
                for c1 in range(c0 + 1, nch):
                    out.append(torch.stack([X[b, c0], X[b, c1]], dim=0))
        X = torch.stack(out, dim=0)
    reim = torch.view_as_real(X.contiguous())                             # (B', 2, nf, nt, 2)
    return reim[:, :, 1:nfft // 2 + 1].contiguous()


# --------------------------------------------------------------------------------------
# Masks                                   (code/common/utils_module.py:255-273, 305-308)
# --------------------------------------------------------------------------------------
def gen_masks(nbatch, npatch, nmasked_patch, nmic=2, rng=random):
    """Host RNG call order of ``PatchMask.forward`` for patch_mode 'T': per batch item
    ``random.sample(range(npatch), nmasked)`` then ``random.randint(0, nmic-1)``.
    Returns (mask_patch_idx int64 (B, nmasked), mask_ch_idx int64 (B,))."""
    idx = np.empty((nbatch, nmasked_patch), dtype=np.int64)
    ch = np.empty((nbatch,), dtype=np.int64)
    for b in range(nbatch):
        idx[b] = rng.sample(range(0, npatch), nmasked_patch)
        ch[b] = rng.randint(0, nmic - 1)
    return torch.from_numpy(idx), torch.from_numpy(ch)


def dense_masks(mask_patch_idx, mask_ch_idx, npatch, nmic=2):
    """(mp (B, npatch), mc (B, nmic)) float32, 0 = masked (utils_module.py:268-270)."""
    nb = mask_patch_idx.shape[0]
    mp = torch.ones(nb, npatch)
    mp.scatter_(1, mask_patch_idx, 0.0)
    mc = torch.ones(nb, nmic)
    mc.scatter_(1, mask_ch_idx.view(nb, 1), 0.0)
    return mp, mc


# --------------------------------------------------------------------------------------
# Conformer pieces
# --------------------------------------------------------------------------------------
def _drop(x, p, train):
    return F.dropout(x, p=p, training=train) if (train and p > 0.0) else x


def feed_forward(x, sd, pre, p_drop, train):
    """FeedForwardModule (conformer/feed_forward.py:47-57): LN -> Linear(d,4d) -> Swish ->
    Dropout -> Linear(4d,d) -> Dropout."""
    h = F.layer_norm(x, (x.shape[-1],), sd[pre + "0.weight"], sd[pre + "0.bias"], EPS_LN)
    h = F.linear(h, sd[pre + "1.linear.weight"], sd[pre + "1.linear.bias"])
    h = h * torch.sigmoid(h)                                             # activation.py:27
    h = _drop(h, p_drop, train)
    h = F.linear(h, sd[pre + "4.linear.weight"], sd[pre + "4.linear.bias"])
    return _drop(h, p_drop, train)


def relative_shift(pos_score):
    """``_relative_shift`` (conformer/attention.py:105-113), written as the index map it is:
    out[i, j] = pos[i, T-1-(i-j)] for j <= i;  0 for j == i+1;  pos[i+1, j-i-2] for j >= i+2."""
    T1, T2 = pos_score.shape[-2:]
    assert T1 == T2
    T = T1
    i = torch.arange(T).view(T, 1)
    j = torch.arange(T).view(1, T)
    lower = j <= i
    upper = j >= i + 2
    src_row = torch.where(lower, i, (i + 1).clamp(max=T - 1)).expand(T, T)
    src_col = torch.where(lower, T - 1 - (i - j), (j - i - 2).clamp(min=0))
    gathered = pos_score[..., src_row, src_col]
    return gathered * (lower | upper).to(pos_score.dtype)


def mhsa(x, sd, pre, num_heads, p_drop, train):
    """MultiHeadedSelfAttentionModule + RelativeMultiHeadAttention
    (conformer/attention.py:143-151, 72-103; embedding.py:31-42).  Note the 1/sqrt(d_model)
    scaling (attention.py:57,91) and that the positional projection is batch-invariant."""
    B, T, d = x.shape
    dh = d // num_heads
    a = pre + "attention."
    pe = sd[pre + "positional_encoding.pe"][0, :T]                        # (T, d)
    h = F.layer_norm(x, (d,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], EPS_LN)
    q = F.linear(h, sd[a + "query_proj.linear.weight"], sd[a + "query_proj.linear.bias"]).view(B, T, num_heads, dh)
    k = F.linear(h, sd[a + "key_proj.linear.weight"], sd[a + "key_proj.linear.bias"]).view(B, T, num_heads, dh)
    v = F.linear(h, sd[a + "value_proj.linear.weight"], sd[a + "value_proj.linear.bias"]).view(B, T, num_heads, dh)
    pos = F.linear(pe, sd[a + "pos_proj.linear.weight"]).view(T, num_heads, dh)
    content = torch.einsum("bihd,bjhd->bhij", q + sd[a + "u_bias"], k)
    pos_score = torch.einsum("bihd,jhd->bhij", q + sd[a + "v_bias"], pos)
    score = (content + relative_shift(pos_score)) / math.sqrt(d)
    attn = _drop(torch.softmax(score, dim=-1), p_drop, train)
    ctx = torch.einsum("bhij,bjhd->bihd", attn, v).reshape(B, T, d)
    out = F.linear(ctx, sd[a + "out_proj.linear.weight"], sd[a + "out_proj.linear.bias"])
    return _drop(out, p_drop, train)


def batch_norm(x, sd, pre, train, channel_dim=1):
    """nn.BatchNorm{1,2}d (channel dim 1): batch statistics (biased var) in train mode with the running-stat update
    (momentum 0.1, unbiased var), running statistics in eval mode."""
    assert channel_dim == 1
    if train:
        with torch.no_grad():
            sd[pre + "num_batches_tracked"] += 1
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        training=train, momentum=BN_MOMENTUM, eps=EPS_BN)


def conv_module(x, sd, pre, p_drop, train):
    """ConformerConvModule (conformer/convolution.py:136-149): LN -> PW conv d->2d -> GLU ->
    depthwise conv k=31 (no bias) -> BatchNorm1d -> Swish -> PW conv d->d -> Dropout."""
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[pre + "0.weight"], sd[pre + "0.bias"], EPS_LN)
    h = F.linear(h, sd[pre + "2.conv.weight"][:, :, 0], sd[pre + "2.conv.bias"])    # (B, T, 2d)
    h = h[..., :d] * torch.sigmoid(h[..., d:])                                      # activation.py:40-42
    wd = sd[pre + "4.conv.weight"]                                                  # (d, 1, K)
    K = wd.shape[-1]
    h = F.conv1d(h.transpose(1, 2), wd, None, padding=(K - 1) // 2, groups=d)       # (B, d, T)
    h = batch_norm(h, sd, pre + "5.", train, channel_dim=1)
    h = h * torch.sigmoid(h)
    # the reference applies this Dropout to the (B, d, T) output of the pointwise Conv1d, before the final transpose
    # (convolution.py:144-149): same values either way, but the mask is DRAWN in (B, d, T) order - kept here so that, under the
    # same torch seed, the oracle consumes the generator exactly like the reference (28 draws per step)
    h = F.conv1d(h, sd[pre + "7.conv.weight"], sd[pre + "7.conv.bias"])             # (B, d, T)
    return _drop(h, p_drop, train).transpose(1, 2)


def conformer_block(x, sd, pre, num_heads, p_drop, train):
    """ConformerBlock (code/common/Conformer.py:59-91), half-step FFN residuals."""
    s = pre + "sequential."
    x = x + 0.5 * feed_forward(x, sd, s + "0.module.sequential.", p_drop, train)
    x = x + mhsa(x, sd, s + "1.module.", num_heads, p_drop, train)
    x = x + conv_module(x, sd, s + "2.module.sequential.", p_drop, train)
    x = x + 0.5 * feed_forward(x, sd, s + "3.module.sequential.", p_drop, train)
    return F.layer_norm(x, (x.shape[-1],), sd[s + "4.weight"], sd[s + "4.bias"], EPS_LN)


def conformer_encoder(x, sd, pre, num_layers, num_heads=4, p_drop=0.1, train=False):
    """ConformerEncoder.forward (code/common/Conformer.py:165-195), add_same_one=False."""
    for L in range(num_layers):
        x = conformer_block(x, sd, pre + "layers.%d." % L, num_heads, p_drop, train)
    return x


# --------------------------------------------------------------------------------------
# Encoder / decoder / full model
# --------------------------------------------------------------------------------------
def stem(img, sd, pre, train):
    """``patch_embed`` (code/model.py:50-64): four bias-free convs each followed by
    BatchNorm2d + ReLU, then the (nf,1)/(nf,1) patch conv.  img: (B, 4, F, T)."""
    h = img
    for conv_i, bn_i, pad in ((0, 1, 0), (3, 4, 1), (6, 7, 1), (9, 10, 0)):
        h = F.conv2d(h, sd[pre + "%d.weight" % conv_i], None, padding=pad)
        h = torch.relu(batch_norm(h, sd, pre + "%d." % bn_i, train, channel_dim=1))
    w = sd[pre + "12.weight"]                                            # (d, 4, F, 1)
    return F.conv2d(h, w, None, stride=(w.shape[2], w.shape[3]))          # (B, d, 1, T)


def embed_encoder(vec, sd, pre, num_layers, train, p_drop=0.1):
    """EmbedEncoder.forward for model=['cnn','conformer'] (code/model.py:194-220).
    vec: (B, T, F, reim, mic).  PatchRecover with patch (F,1) is the pure permutation
    img[b, reim*2+mic, f, t] = vec[b, t, f, reim, mic] (utils_module.py:222-231)."""
    B, T, Fq = vec.shape[:3]
    img = vec.permute(0, 3, 4, 2, 1).reshape(B, -1, Fq, T)
    e = stem(img, sd, pre + "patch_embed.", train)
    e = e.reshape(B, e.shape[1], T).permute(0, 2, 1)
    return conformer_encoder(e, sd, pre + "embed.", num_layers, 4, p_drop, train)


def decoder(embed, sd, pre="decoder.proj."):
    """EmbedDecoder.forward for model=['','fc'] (code/model.py:321-334, ctor :295-301)."""
    h = torch.relu(F.linear(embed, sd[pre + "0.weight"], sd[pre + "0.bias"]))
    return F.linear(h, sd[pre + "2.weight"], sd[pre + "2.bias"])


def sarssl_pretrain_forward(x, sd, mask_patch_idx, mask_ch_idx, train=False, p_drop=0.1,
                            spec_layers=1, spat_layers=3, return_pred=True):
    """``SARSSL.forward`` pretrain branch (code/model.py:519-601) with explicit masks.

    x: (B, nmic=2, F, T, 2).  Returns (loss, diff, aux) where aux holds 'pred'
    (B, T, F, 2, 2) = vec_patch_pred and 'tar' = vec_patch (patch-domain, before the fold of
    ``vis_results``)."""
    B, nmic, Fq, T, _ = x.shape
    v = x.permute(0, 3, 2, 4, 1)                                          # (B, T, F, reim, mic)  model.py:524-525
    mp, mc = dense_masks(mask_patch_idx, mask_ch_idx, T, nmic)
    mp5 = mp.view(B, T, 1, 1, 1)
    mc5 = mc.view(B, 1, 1, 1, nmic)
    spec_in = v * (1 - mp5) * mc5 + v * mp5 * (1 - mc5)                   # model.py:541
    spat_in = v * mp5                                                     # model.py:563
    e_spec = embed_encoder(spec_in, sd, "spec_encoder.", spec_layers, train, p_drop)
    e_spat = embed_encoder(spat_in, sd, "spat_encoder.", spat_layers, train, p_drop)
    pred = decoder(torch.cat([e_spec, e_spat], dim=2), sd).reshape(B, T, Fq, 2, nmic)   # model.py:581-589
    tar = (v * (1 - mc5)).sum(-1)                                         # model.py:585
    other = (v * mc5).sum(-1)                                             # model.py:586
    pred_sel = (pred * (1 - mc5)).sum(-1)                                 # model.py:590
    gi = mask_patch_idx.view(B, -1, 1, 1).expand(-1, -1, Fq, 2)
    p_m, t_m, o_m = pred_sel.gather(1, gi), tar.gather(1, gi), other.gather(1, gi)      # model.py:736-740
    loss = torch.mean((p_m - t_m) ** 2)                                   # model.py:743
    diff = torch.mean((t_m - o_m) ** 2)                                   # model.py:745
    aux = {"pred": pred, "tar": v} if return_pred else {}
    return loss, diff, aux


def sarssl_downstream_forward(x, sd, embed_use="spat", train=False, p_drop=0.1,
                              spec_layers=1, spat_layers=3):
    """``SARSSL.forward`` downstream branch (code/model.py:667-719) for
    downstream_head='mlp', downstream_dlabel=1."""
    B, nmic, Fq, T, _ = x.shape
    v = x.permute(0, 3, 2, 4, 1)
    e_spec = embed_encoder(v, sd, "spec_encoder.", spec_layers, train, p_drop)
    e_spat = embed_encoder(v, sd, "spat_encoder.", spat_layers, train, p_drop)
    if embed_use == "spec_spat":
        e = torch.cat([e_spec, e_spat], dim=2)
    elif embed_use == "spec":
        e = e_spec
    else:
        e = e_spat
    m = e.mean(dim=1)
    h = F.layer_norm(m, (m.shape[-1],), sd["mlp_head.0.weight"], sd["mlp_head.0.bias"], EPS_LN)
    return F.linear(h, sd["mlp_head.1.weight"], sd["mlp_head.1.bias"]), m


# --------------------------------------------------------------------------------------
# Optimiser / schedule / training step       (code/learner.py:83, 95-113;
#                                             code/common/utils.py:108-139)
# --------------------------------------------------------------------------------------
BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "positional_encoding.pe")


def is_param(key):
    return not key.endswith(BUFFER_SUFFIXES)


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults as used at learner.py:83 (weight_decay 0, no amsgrad)."""
    b1, b2 = betas
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def cosine_lr(epoch, total_steps, base, warmup_steps=1):
    """create_learning_rate_schedule(decay_type='cosine') (code/common/utils.py:108-139)."""
    progress = float(np.clip((epoch - warmup_steps) / float(total_steps - warmup_steps), 0.0, 1.0))
    lr = base * 0.5 * (1.0 + math.cos(math.pi * progress))
    if warmup_steps:
        lr = lr * min(1.0, epoch / warmup_steps)
    return float(np.float32(lr))


def train_step(mic_sig, sd, opt_state, lr, mask_patch_idx=None, mask_ch_idx=None, p_drop=0.1,
               rng=random):
    """One iteration of ``Learner.pretrain_epoch`` (code/learner.py:95-113): preprocess ->
    forward (train mode) -> backward -> Adam.  ``sd`` is updated in place.  Returns
    (loss, diff) floats."""
    x = data_preprocess(mic_sig)
    B, _, Fq, T, _ = x.shape
    if mask_patch_idx is None:
        mask_patch_idx, mask_ch_idx = gen_masks(B, T, T // 2, 2, rng)
    params = {k: t for k, t in sd.items() if is_param(k)}
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    loss, diff, _ = sarssl_pretrain_forward(x, sd, mask_patch_idx, mask_ch_idx, train=True,
                                            p_drop=p_drop, return_pred=False)
    loss.backward()
    with torch.no_grad():
        grads = {k: p.grad for k, p in params.items()}
        for p in params.values():
            p.requires_grad_(False)
        adam_step(params, grads, opt_state, lr)
        for p in params.values():
            p.grad = None
    return float(loss.detach()), float(diff.detach())


def pretrain_epoch(batches, sd, lr, p_drop=0.1, rng=random):
    """``Learner.pretrain_epoch`` (code/learner.py:76-131): train mode, a NEW Adam (zero moments, step count 0) per call, one
    ``train_step`` per batch, returns (mean loss, mean diff, pred of the LAST batch as computed before its optimiser step - the
    ``vis_batch`` the reference returns, folded to (B, F, T, reim, mic) like ``vis_results``, code/model.py:776-790)."""
    opt_state = {}
    loss_sum = diff_sum = 0.0
    last = None
    for mic_sig in batches:
        x = data_preprocess(mic_sig)
        B, _, Fq, T, _ = x.shape
        idx, ch = gen_masks(B, T, T // 2, 2, rng)
        with torch.no_grad():                                            # the returned vis uses the pre-step weights (train-mode BN)
            last = (x, idx, ch, {k: v.clone() for k, v in sd.items()})
        l, d = train_step(mic_sig, sd, opt_state, lr, idx, ch, p_drop=p_drop)
        loss_sum += l
        diff_sum += d
    x, idx, ch, sd_before = last
    with torch.no_grad():
        _, _, aux = sarssl_pretrain_forward(x, sd_before, idx, ch, train=True, p_drop=p_drop)
    n = len(batches)
    return loss_sum / n, diff_sum / n, aux["pred"].permute(0, 2, 1, 3, 4)


def downstream_train_step(mic_sig, tdoa, sd, opt_state, lr, embed_use="spat", p_drop=0.1, frozen=(), fs=16000):
    """One iteration of ``Learner.train_epoch`` (code/learner.py:186-202) for task 'TDOA': preprocess -> downstream
    forward in train mode -> ``mse_loss(pred, TDOA*fs)`` (learner.py:620-631, 644-647) -> backward -> Adam over the
    parameters that are not ``frozen`` (key prefixes; 'lineareval' freezes what came from the pretraining checkpoint,
    learner.py:441-444).  ``sd`` is updated in place.  Returns (loss, metric, pred, mean_embed)."""
    x = data_preprocess(mic_sig)
    tar = tdoa.reshape(-1, 1).float() * fs
    params = {k: t for k, t in sd.items() if is_param(k) and not k.startswith(tuple(frozen))} if frozen else \
        {k: t for k, t in sd.items() if is_param(k)}
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    pred, emb = sarssl_downstream_forward(x, sd, embed_use=embed_use, train=True, p_drop=p_drop)
    loss = F.mse_loss(pred, tar)
    loss.backward()
    with torch.no_grad():
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
        for p in params.values():
            p.requires_grad_(False)
        used = {k: p for k, p in params.items() if grads[k] is not None}
        adam_step(used, grads, opt_state, lr)
        for p in params.values():
            p.grad = None
        metric = (pred.detach() - tar).abs().mean()
    opt_state["last_grads"] = grads
    return float(loss.detach()), float(metric), pred.detach(), emb.detach()


def downstream_eval_step(mic_sig, tdoa, sd, embed_use="spat", fs=16000):
    """One iteration of ``Learner.test_epoch`` (code/learner.py:236-246): eval-mode forward, MSE loss and MAE metric."""
    with torch.no_grad():
        x = data_preprocess(mic_sig)
        tar = tdoa.reshape(-1, 1).float() * fs
        pred, emb = sarssl_downstream_forward(x, sd, embed_use=embed_use, train=False)
        return float(F.mse_loss(pred, tar)), float((pred - tar).abs().mean()), pred, emb


def sarssl_multich_forward(x, sd, nmic_pair):
    """``SARSSL_MultiCH.forward`` (code/model.py:808-821), eval mode: spat encoder of the single-pair model on every pair ->
    mean over frames -> concatenate the pairs of a segment -> LayerNorm, Linear, ReLU, Linear head."""
    v = x.permute(0, 3, 2, 4, 1)
    e = embed_encoder(v, sd, "model_sch.spat_encoder.", 3, False).mean(dim=1)
    e = e.reshape(-1, nmic_pair * e.shape[-1])
    h = F.layer_norm(e, (e.shape[-1],), sd["head_mch.0.weight"], sd["head_mch.0.bias"], EPS_LN)
    h = F.relu(F.linear(h, sd["head_mch.1.weight"], sd["head_mch.1.bias"]))
    return F.linear(h, sd["head_mch.3.weight"], sd["head_mch.3.bias"]), e


# --------------------------------------------------------------------------------------
# Eval / export path        (code/common/utils_module.py:74-113, code/learner.py:574-618)
# --------------------------------------------------------------------------------------
def istft(spec, win_len=512, win_shift_ratio=0.5, nfft=512, inv=False):
    """``ISTFT.forward`` (utils_module.py:91-113): per channel ``torch.istft(n_fft, hop, win_length, window=None,
    center=inv)``.  Restated without torch.istft: irfft of every frame (imaginary parts of the DC / Nyquist bins are
    ignored by the real inverse transform), overlap-add, division by the rectangular window's envelope (number of
    frames covering a sample); ``inv`` (center=True) drops nfft/2 samples at both ends.
    spec complex (B, nfft/2+1, nt, nch) -> (B, nsample, nch) float32."""
    hop = int(win_len * win_shift_ratio)
    nb, nf, nt, nch = spec.shape
    frames = torch.fft.irfft(spec.permute(0, 3, 2, 1).to(torch.complex64), n=nfft, dim=-1)      # (B, nch, nt, nfft)
    full = nfft + hop * (nt - 1)
    out = torch.zeros((nb, nch, full), dtype=torch.float32)
    env = torch.zeros(full, dtype=torch.float32)
    for t in range(nt):
        out[:, :, t * hop:t * hop + nfft] += frames[:, :, t]
        env[t * hop:t * hop + nfft] += 1.0
    out = out / env
    if inv:
        out = out[:, :, nfft // 2: nfft // 2 + (nt - 1) * hop]
    return out.permute(0, 2, 1).contiguous()


def pretrain_evaluate(pred, gt, mask):
    """``STFTLearner.pretrain_evaluate`` (learner.py:574-618) without the PESQ scores (third-party torchmetrics / pesq,
    absent here: that part of the path is 'parity unpinned').  pred, gt (nb,nf,nt,nreim,nch); mask (nb,nf,nt,nch), 0 = masked."""
    def to_sig(v):
        st = torch.view_as_complex(v.permute(0, 1, 2, 4, 3).contiguous())
        st = torch.cat((torch.zeros_like(st[:, 0:1]), st), dim=1)
        sig = istft(st)
        return sig / torch.max(sig)
    mask_dense = mask[:, :, :, None, :].tile(1, 1, 1, 2, 1)
    diff = (pred - gt) ** 2
    diff_mask = diff * (1 - mask_dense)
    return {"sig_pred": to_sig(pred), "sig_tar": to_sig(gt), "mse": torch.mean(diff),
            "mse_mask": torch.sum(diff_mask) / torch.sum(1 - mask_dense), "mse_mask_ch": torch.mean(torch.sum(diff_mask, dim=4))}
