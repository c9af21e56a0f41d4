"""Generate tests/golden/* by running the REAL reference (build container only).

    python oracle/make_golden.py [--curve]     # --curve adds the 100-step loss curve (~8 min CPU)

Each fixture stores inputs (or the recipe that regenerates them) and the reference's outputs.
Nothing of the reference's source travels; the vectors are data.  See SURVEY.md section 8c (F1-F6); F7-F12 cover the later rows.
"""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim                      # noqa: E402
import recipes                       # noqa: E402
import sarssl_oracle as orc          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def manifest_of(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def load_recipe(module, seed):
    man = manifest_of(module)
    sd = recipes.recipe_state_dict(man, seed)
    module.load_state_dict(sd)
    return man


def set_dropout(module, p):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p


def grads_of(module):
    return {k: p.grad.detach().clone() for k, p in module.named_parameters()}


def sample_idx(numel, n, seed):
    g = np.random.default_rng(seed)
    return np.sort(g.choice(numel, size=min(n, numel), replace=False)).astype(np.int64)


def f_manifest(ref_model):
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    ds = ref_model.SARSSL(sig_shape=(256, 64, 2, 2), pretrain=False, device="cpu", downstream_token="all",
                          downstream_head="mlp", downstream_embed="spat", downstream_dlabel=1)
    out = {"pretrain": manifest_of(net), "downstream": manifest_of(ds),
           "nparams_pretrain": int(sum(p.numel() for p in net.parameters()))}
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump(out, f, indent=0)
    return net


def f1_frontend(ref_learner, ref_model):
    dummy = torch.nn.Linear(1, 1)
    lrn = ref_learner.STFTLearner(dummy, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1,
                                  fs=16000, task=None, ch_mode="M")
    lrn.device = "cpu"
    small = recipes.recipe_signal(2, 2048, 2, seed=1)
    out_small, = lrn.data_preprocess(small, None)
    small4 = recipes.recipe_signal(2, 1536, 4, seed=2)
    out_small4, = lrn.data_preprocess(small4, None)
    full = recipes.recipe_signal(2, 65792, 2, seed=3)
    out_full, = lrn.data_preprocess(full, None)
    idx = sample_idx(out_full.numel(), 4096, 11)
    stft_raw = lrn.stft(signal=small)     # (B, 257, nt, nch) complex
    np.savez_compressed(
        os.path.join(GOLD, "f1_frontend.npz"),
        small_out=out_small.numpy(), small4_out=out_small4.numpy(),
        small_stft_re=stft_raw.real.numpy(), small_stft_im=stft_raw.imag.numpy(),
        full_idx=idx, full_vals=out_full.reshape(-1)[idx].numpy(),
        full_sum=np.float64(out_full.double().sum()), full_sumsq=np.float64((out_full.double() ** 2).sum()),
        full_shape=np.array(out_full.shape))


def f2_blocks(ref_model):
    sys.path.insert(0, ref_shim.REF_CODE)
    from common.conformer.feed_forward import FeedForwardModule
    from common.conformer.attention import MultiHeadedSelfAttentionModule, RelativeMultiHeadAttention
    from common.conformer.convolution import ConformerConvModule
    from common.Conformer import ConformerBlock, ConformerEncoder
    store = {}
    meta = {}

    def run(name, module, x, seed, fwd=None):
        man = load_recipe(module, seed)
        meta[name] = man
        set_dropout(module, 0.0)
        for mode in ("eval", "train"):
            module.train(mode == "train")
            module.load_state_dict(recipes.recipe_state_dict(man, seed))
            xi = x.clone().requires_grad_(True)
            module.zero_grad()
            y = fwd(module, xi) if fwd else module(xi)
            gy = torch.from_numpy(np.random.default_rng(seed + 77).standard_normal(tuple(y.shape)).astype(np.float32))
            (y * gy).sum().backward()
            store["%s.%s.y" % (name, mode)] = y.detach().numpy()
            store["%s.%s.dx" % (name, mode)] = xi.grad.numpy()
            for k, g in grads_of(module).items():
                store["%s.%s.grad.%s" % (name, mode, k)] = g.numpy()
            if mode == "train":
                for k, v in module.state_dict().items():
                    if k.endswith(("running_mean", "running_var")):
                        store["%s.train.after.%s" % (name, k)] = v.detach().clone().numpy()
        store["%s.x" % name] = x.numpy()
        store["%s.gy" % name] = gy.numpy()

    g = np.random.default_rng(5)
    d, T, B = 32, 16, 3
    x = torch.from_numpy(g.standard_normal((B, T, d)).astype(np.float32))
    run("ffn", FeedForwardModule(encoder_dim=d, expansion_factor=4, dropout_p=0.1), x, 21)
    run("mhsa", MultiHeadedSelfAttentionModule(d_model=d, num_heads=4, dropout_p=0.1), x, 22)
    run("convmod", ConformerConvModule(in_channels=d, kernel_size=31, expansion_factor=2, dropout_p=0.1), x, 23)
    run("block", ConformerBlock(encoder_dim=d, num_attention_heads=4), x, 24)
    run("encoder2", ConformerEncoder(encoder_dim=d, num_layers=2, num_attention_heads=4), x, 25,
        fwd=lambda m, xi: m(xi, False))
    # T longer than the depthwise kernel so both conv borders and the interior are exercised
    x40 = torch.from_numpy(g.standard_normal((2, 40, d)).astype(np.float32))
    run("block_T40", ConformerBlock(encoder_dim=d, num_attention_heads=4), x40, 26)
    # EmbedEncoder at reduced size: F=16, T=8
    Fq, Tq = 16, 8
    enc = ref_model.EmbedEncoder(sig_shape=[Fq, Tq, 2, 2], patch_shape=(Fq, 1), dembed=32,
                                 model=["cnn", "conformer"], mode="spat", device="cpu")
    xe = torch.from_numpy(g.standard_normal((B, Tq, Fq * 4)).astype(np.float32))
    run("embed_encoder", enc, xe, 27, fwd=lambda m, xi: m.forward(xi))
    dec = ref_model.EmbedDecoder(sig_shape=[Fq, Tq, 2, 2], patch_shape=(Fq, 1), dembed=48, model=["", "fc"])
    xd = torch.from_numpy(g.standard_normal((B, Tq, 48)).astype(np.float32))
    run("embed_decoder", dec, xd, 28, fwd=lambda m, xi: m.forward(xi))
    # relative-shift tables
    rel = RelativeMultiHeadAttention(d_model=8, num_heads=2)
    for T_ in (4, 5, 7):
        ps = torch.arange(T_ * T_, dtype=torch.float32).reshape(1, 1, T_, T_)
        store["relshift.T%d" % T_] = rel._relative_shift(ps)[0, 0].numpy()
    store["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, "f2_blocks.npz"), **store)


def f3_fullsize(ref_model, ref_learner):
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    man = load_recipe(net, 0)
    set_dropout(net, 0.0)
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1,
                                  fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    sig = recipes.recipe_signal(2, 65792, 2, seed=3)
    x, = lrn.data_preprocess(sig, None)
    store = {}
    for mode in ("eval", "train"):
        net.train(mode == "train")
        net.load_state_dict(recipes.recipe_state_dict(man, 0))
        net.zero_grad()
        random.seed(4321)
        idx, ch = orc.gen_masks(2, 256, 128, 2, random)     # same call order as PatchMask.forward
        random.seed(4321)
        loss, diff, vis = net(x)
        loss.backward()
        pred_patch = vis["pred"].permute(0, 2, 1, 3, 4).contiguous()     # (B,T,F,reim,mic)
        sidx = sample_idx(pred_patch.numel(), 2048, 13)
        store["%s.loss" % mode] = np.float64(loss.item())
        store["%s.diff" % mode] = np.float64(diff.item())
        store["%s.pred_idx" % mode] = sidx
        store["%s.pred_vals" % mode] = pred_patch.reshape(-1)[sidx].numpy()
        store["%s.pred_absmax" % mode] = np.float64(pred_patch.abs().max())
        # the dense mask of the reference tells us which channel/frames it masked: check vs oracle masks
        m = vis["mask"]                                                   # (B,F,T,mic), 0 = masked
        masked_ch = (m[:, 0].sum(dim=1)).argmin(dim=1)                    # (B,)
        assert torch.equal(masked_ch, ch), "mask channel order mismatch"
        mf = [(m[b, 0, :, int(ch[b])] == 0).nonzero().flatten() for b in range(2)]
        for b in range(2):
            assert torch.equal(mf[b], torch.sort(idx[b]).values), "mask frame set mismatch"
        gn = {k: float(p.grad.double().norm()) for k, p in net.named_parameters()}
        store["%s.gradnorm_json" % mode] = np.array(json.dumps(gn))
        if mode == "train":
            for k in ("spec_encoder.patch_embed.4.running_mean", "spec_encoder.patch_embed.4.running_var",
                      "spat_encoder.embed.layers.1.sequential.2.module.sequential.5.running_var"):
                store["train.after." + k] = net.state_dict()[k].numpy().copy()
    store["mask_idx"] = idx.numpy()
    store["mask_ch"] = ch.numpy()
    np.savez_compressed(os.path.join(GOLD, "f3_fullsize.npz"), **store)


def f4_masks(ref_um):
    store = {}
    for seed in (0, 7, 123456):
        pm = ref_um.PatchMask(patch_mode="T", nmasked_patch=128, npatch_shape=[1, 256], device="cpu")
        random.seed(seed)
        _, _, _, idx, ch = pm.forward((4, 256, 256, 2, 2))
        store["seed%d.idx" % seed] = idx.numpy()
        store["seed%d.ch" % seed] = ch.numpy()
    np.savez_compressed(os.path.join(GOLD, "f4_masks.npz"), **store)


def f7_downstream(ref_model):
    ds = ref_model.SARSSL(sig_shape=(256, 64, 2, 2), pretrain=False, device="cpu", downstream_token="all",
                          downstream_head="mlp", downstream_embed="spat", downstream_dlabel=1)
    load_recipe(ds, 5)
    ds.eval()
    g = np.random.default_rng(99)
    x = torch.from_numpy(g.standard_normal((2, 2, 256, 64, 2)).astype(np.float32))
    with torch.no_grad():
        pred, emb = ds(x)
    np.savez_compressed(os.path.join(GOLD, "f7_downstream.npz"), pred=pred.numpy(), embed=emb.numpy())


def f9_downstream_train(ref_model, ref_learner, nstep=3, B=4):
    """TDOA fine-tuning steps of the real reference (code/learner.py:168-222 with code/model.py:667-719): per-step loss /
    metric / predictions, step-1 gradient norms, for 'finetune' (everything trains) and 'lineareval' (encoders frozen,
    learner.py:441-444), plus what Learner.train_epoch / test_epoch return over the same three batches."""
    from sar_ssl_amd import synth
    pool = torch.from_numpy(synth.make_batch(100, nstep * B))[:, :16640].contiguous()           # T = 1.04 s -> nt = 64
    tdoa = torch.from_numpy(np.random.default_rng(77).uniform(-0.2 / 343, 0.2 / 343, size=(nstep * B,)).astype(np.float32))
    store = {"tdoa": tdoa.numpy(), "B": B, "lr": 1e-4, "weight_seed": 5, "sig_seed": 100}
    for mode in ("finetune", "lineareval"):
        ds = ref_model.SARSSL(sig_shape=(256, 64, 2, 2), pretrain=False, device="cpu", downstream_token="all",
                              downstream_head="mlp", downstream_embed="spat", downstream_dlabel=1)
        load_recipe(ds, 5)
        set_dropout(ds, 0.0)
        if mode == "lineareval":
            for k, v in ds.named_parameters():
                if k.startswith(("spec_encoder.", "spat_encoder.")):
                    v.requires_grad = False
        init = {k: v.clone() for k, v in ds.state_dict().items()}
        lrn = ref_learner.STFTLearner(ds, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000,
                                      task="TDOA", ch_mode="M")
        lrn.cpu()
        ds.train()
        opt = torch.optim.Adam(ds.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0)
        losses, metrics, preds = [], [], []
        for s in range(nstep):
            sig, gt = pool[s * B:(s + 1) * B], {"TDOA": tdoa[s * B:(s + 1) * B]}
            x, tar = lrn.data_preprocess(sig, gt)
            pred, emb = ds(x)
            loss = lrn.loss(pred_batch=pred, gt_batch=tar)
            opt.zero_grad()
            loss.backward()
            if s == 0:
                for k, v in ds.named_parameters():
                    store["%s.gradnorm.%s" % (mode, k)] = np.float64(v.grad.double().norm().item()) if v.grad is not None else np.float64(-1.0)
                store[mode + ".embed0"] = emb.detach().numpy()
            opt.step()
            losses.append(loss.item()); metrics.append(lrn.evaluate(pred_batch=pred, gt_batch=tar).item()); preds.append(pred.detach().numpy())
            print(mode, "step", s, losses[-1], metrics[-1], flush=True)
        store[mode + ".loss"], store[mode + ".metric"], store[mode + ".pred"] = np.array(losses), np.array(metrics), np.stack(preds)
        # the reference's own epoch drivers on the same batches, from the same initial state
        ds.load_state_dict(init)
        loader = [(pool[s * B:(s + 1) * B], {"TDOA": tdoa[s * B:(s + 1) * B]}) for s in range(nstep)]
        ltr, mtr = lrn.train_epoch(loader, lr=1e-4, epoch=1, return_metric=True)
        lte, mte = lrn.test_epoch(loader, return_metric=True)
        store[mode + ".train_epoch"] = np.array([float(ltr), float(mtr)])
        store[mode + ".test_epoch"] = np.array([float(lte), float(mte)])
        print(mode, "train_epoch", float(ltr), float(mtr), "test_epoch", float(lte), float(mte), flush=True)
    np.savez_compressed(os.path.join(GOLD, "f9_downstream_train.npz"), **store)


def f10_multich(ref_model, ref_learner):
    """SURVEY.md 8f-2 / BASELINE config 5: (a) ch_mode 'MM' pairing of a 4-mic signal; (b) one 4-mic x 160 000-sample segment
    -> 3 mic pairs at T = 624 through the pretraining forward/backward (train mode, dropout off, replayed masks);
    (c) SARSSL_MultiCH (code/model.py:793-821) eval forward on 3 pairs."""
    store = {}
    dummy = ref_model.SARSSL(sig_shape=(256, 16, 2, 2), pretrain=True, device="cpu")
    lrn_mm = ref_learner.STFTLearner(dummy, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="MM")
    lrn_mm.cpu()
    small4 = recipes.recipe_signal(2, 1536, 4, seed=2)
    store["mm_small4_out"] = lrn_mm.data_preprocess(small4, None)[0].numpy()
    # (b)
    T = 624
    net = ref_model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device="cpu")
    man = load_recipe(net, 0)
    set_dropout(net, 0.0)
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    sig = recipes.recipe_signal(1, 160000, 4, seed=21)
    x, = lrn.data_preprocess(sig, None)
    assert tuple(x.shape) == (3, 2, 256, T, 2), x.shape
    net.train()
    net.zero_grad()
    random.seed(777)
    idx, ch = orc.gen_masks(3, T, T // 2, 2, random)
    random.seed(777)
    loss, diff, vis = net(x)
    loss.backward()
    pred_patch = vis["pred"].permute(0, 2, 1, 3, 4).contiguous()
    sidx = sample_idx(pred_patch.numel(), 2048, 17)
    store.update({"c5.loss": np.float64(loss.item()), "c5.diff": np.float64(diff.item()), "c5.pred_idx": sidx,
                  "c5.pred_vals": pred_patch.reshape(-1)[sidx].numpy(), "c5.pred_absmax": np.float64(pred_patch.abs().max()),
                  "c5.mask_idx": idx.numpy(), "c5.mask_ch": ch.numpy(),
                  "c5.gradnorm_json": np.array(json.dumps({k: float(p.grad.double().norm()) for k, p in net.named_parameters()}))})
    print("config5: loss", loss.item(), "diff", diff.item(), flush=True)
    # (c)
    mch = ref_model.SARSSL_MultiCH(sig_shape=(256, 32, 2, 2), nmic_pair=3, task="TDOA", device="cpu")
    man_m = load_recipe(mch, 11)
    mch.eval()
    xm = torch.from_numpy(np.random.default_rng(5).standard_normal((6, 2, 256, 32, 2)).astype(np.float32))
    with torch.no_grad():
        pred, emb = mch(xm)
    store.update({"mch.manifest_json": np.array(json.dumps(man_m)), "mch.pred": pred.numpy(), "mch.embed": emb.numpy()})
    np.savez_compressed(os.path.join(GOLD, "f10_multich.npz"), **store)


def f11_eval_export(ref_model, ref_learner, ref_um):
    """ISTFT (both modes) on random spectra, and pretest_epoch(return_eval=True) -> pretrain_evaluate of the real
    reference on one eval batch (PESQ replaced by a constant: torchmetrics/pesq are not installed)."""
    g = np.random.default_rng(31)
    spec = torch.from_numpy(g.standard_normal((2, 257, 9, 3, 2)).astype(np.float32))
    spec_c = torch.view_as_complex(spec)
    store = {"spec": spec.numpy()}
    for inv in (False, True):
        m = ref_um.ISTFT(win_len=512, win_shift_ratio=0.5, nfft=512, inv=inv)
        store["istft_inv%d" % int(inv)] = m(spec_c).numpy()
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    load_recipe(net, 0)
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    ref_learner.perceptual_evaluation_speech_quality = lambda *a, **k: torch.tensor(1.0)
    sig = recipes.recipe_signal(2, 65792, 2, seed=3)
    random.seed(2468)
    idx, ch = orc.gen_masks(2, 256, 128, 2, random)
    random.seed(2468)
    loss, diff, vis, res = lrn.pretest_epoch([[sig]], return_diff=True, return_eval=True)
    sidx = sample_idx(res["sig_pred"].numel(), 4096, 23)
    store.update({"eval.loss": np.float64(loss), "eval.diff": np.float64(diff), "eval.mask_idx": idx.numpy(), "eval.mask_ch": ch.numpy(),
                  "eval.sig_idx": sidx, "eval.sig_pred": res["sig_pred"].reshape(-1)[sidx].numpy(),
                  "eval.sig_tar": res["sig_tar"].reshape(-1)[sidx].numpy(), "eval.sig_shape": np.array(res["sig_pred"].shape),
                  "eval.mse": np.float64(res["mse"]), "eval.mse_mask": np.float64(res["mse_mask"]),
                  "eval.mse_mask_ch": np.float64(res["mse_mask_ch"])})
    print("eval: loss", loss, "mse", float(res["mse"]), "mse_mask", float(res["mse_mask"]), flush=True)
    np.savez_compressed(os.path.join(GOLD, "f11_eval_export.npz"), **store)


def f8_schedule():
    sys.path.insert(0, ref_shim.REF_CODE)
    from common.utils import create_learning_rate_schedule
    fn = create_learning_rate_schedule(total_steps=30, base=0.001, decay_type="cosine", warmup_steps=1, linear_end=1e-6)
    lrs = np.array([float(fn(e)) for e in range(1, 31)], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "f8_schedule.npz"), lr=lrs)


def f5_curve(ref_model, ref_learner, nstep=100, B=8):
    from sar_ssl_amd import synth
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    load_recipe(net, 0)
    set_dropout(net, 0.0)
    net.train()
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1,
                                  fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    pool = torch.from_numpy(synth.make_batch(0, 64))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0)   # learner.py:83
    losses, diffs = [], []
    for s in range(nstep):
        sig = pool[(s * B) % 64:(s * B) % 64 + B]
        x, = lrn.data_preprocess(sig, None)
        random.seed(9000 + s)
        loss, diff, _ = net(x)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item()); diffs.append(diff.item())
        print("step", s, losses[-1], diffs[-1], flush=True)
        np.savez_compressed(os.path.join(GOLD, "f5_curve.npz"), loss=np.array(losses), diff=np.array(diffs),
                            B=B, lr=1e-3, mask_seed_base=9000, pool=64, weight_seed=0)


def f12_pretrain_epoch(ref_model, ref_learner):
    """The reference's OWN ``Learner.pretrain_epoch`` (code/learner.py:76-131): two epochs x four batches of four full-size
    segments, dropout 0, masks from Python's RNG seeded once per epoch, a different learning rate in the second epoch and the
    optimiser re-created per epoch (Q12).  Stores the returned (loss, diff) per epoch, samples of the returned vis dict and, per
    parameter, the L2 norm of its total update - the quantity a carried-over Adam state would change."""
    from sar_ssl_amd import synth
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    load_recipe(net, 0)
    set_dropout(net, 0.0)
    init = {k: v.detach().clone() for k, v in net.named_parameters()}
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    B, nb = 4, 4
    pool = torch.from_numpy(synth.make_batch(2000, B * nb))
    dataset = [[pool[i * B:(i + 1) * B]] for i in range(nb)]
    lrs, seeds = [1e-3, 5e-4], [4100, 4101]
    store = {"B": B, "nbatch": nb, "sig_seed": 2000, "weight_seed": 0, "lr": np.array(lrs), "mask_seed": np.array(seeds)}
    for e in range(2):
        random.seed(seeds[e])
        loss, diff, vis = lrn.pretrain_epoch(dataset, lr=lrs[e], epoch=e + 1)
        store["epoch%d.loss" % (e + 1)] = np.float64(loss)
        store["epoch%d.diff" % (e + 1)] = np.float64(diff)
        pred = vis["pred"].detach()
        idx = sample_idx(pred.numel(), 1024, 31 + e)
        store["epoch%d.pred_idx" % (e + 1)] = idx
        store["epoch%d.pred_vals" % (e + 1)] = pred.reshape(-1)[idx].numpy()
        store["epoch%d.pred_absmax" % (e + 1)] = np.float64(pred.abs().max())
        store["epoch%d.pred_shape" % (e + 1)] = np.array(pred.shape)
        store["epoch%d.mask_zero_frac" % (e + 1)] = np.float64((vis["mask"] == 0).float().mean())
        print("pretrain_epoch", e + 1, loss, diff, flush=True)
    store["update_norm_json"] = json.dumps({k: float((p.detach() - init[k]).double().norm()) for k, p in net.named_parameters()})
    store["training_flag"] = np.int64(net.training)
    np.savez_compressed(os.path.join(GOLD, "f12_pretrain_epoch.npz"), **store)


def f6_checkpoint(ref_model, ref_learner):
    """A checkpoint FILE written by the reference's ``save_checkpoint`` (code/learner.py:344-374; fp32 layout: epoch, max_score,
    model) for a reduced-width MCConformer (the full model's file is 121 MB), gzip-compressed, plus what a loader must reproduce:
    the manifest, the recipe seed of the weights, max_score / epoch and the reference's eval-mode output on a seeded input."""
    import gzip
    import shutil
    import tempfile
    net = ref_model.MCConformer(sig_shape=[16, 8, 2, 2], patch_shape=(16, 1), dembed={"spec": 32, "spat": 32}, device="cpu")
    man = load_recipe(net, 77)
    net.eval()
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    assert lrn.is_best_epoch(-0.125)                                      # sets max_score the way run_pretrain.py does
    x = torch.from_numpy(np.random.default_rng(606).standard_normal((2, 2, 16, 8, 2)).astype(np.float32))
    with torch.no_grad():
        y = net(x)
    with tempfile.TemporaryDirectory() as d:
        lrn.save_checkpoint(epoch=7, checkpoints_dir=d, is_best_epoch=True, save_extra_hist=False)
        assert os.path.exists(os.path.join(d, "best_model.tar"))
        with open(os.path.join(d, "latest_model.tar"), "rb") as fi, gzip.open(os.path.join(GOLD, "f6_checkpoint.tar.gz"), "wb", 9) as fo:
            shutil.copyfileobj(fi, fo)
    np.savez_compressed(os.path.join(GOLD, "f6_checkpoint_meta.npz"), manifest_json=json.dumps(man), weight_seed=77, epoch=7,
                        max_score=np.float64(-0.125), x_seed=606, y=y.numpy())
    print("f6 checkpoint written", os.path.getsize(os.path.join(GOLD, "f6_checkpoint.tar.gz")), flush=True)


def f5_curve_dropout(ref_model, ref_learner, nstep=40, B=8):
    """F5(ii): the same training run as F5 with dropout ON (p = 0.1 everywhere, the reference default).  Before every forward:
    ``random.seed(9000 + s)`` (masks) and ``torch.manual_seed(7000 + s)`` (the 28 dropout draws of the step).  The
    oracle is stepped alongside for the first steps under the same seeds: equal losses prove that the draw order / layouts assumed
    by the build's mask replay (runtime.DropoutReplay) are the reference's."""
    from sar_ssl_amd import synth
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    man = load_recipe(net, 0)
    net.train()
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    pool = torch.from_numpy(synth.make_batch(0, 64))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0)   # learner.py:83
    osd, ostate = recipes.recipe_state_dict(man, 0), {}
    losses, diffs = [], []
    for s in range(nstep):
        sig = pool[(s * B) % 64:(s * B) % 64 + B]
        x, = lrn.data_preprocess(sig, None)
        random.seed(9000 + s)
        torch.manual_seed(7000 + s)
        loss, diff, _ = net(x)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item()); diffs.append(diff.item())
        if s < 3:
            random.seed(9000 + s)
            torch.manual_seed(7000 + s)
            ol, od = orc.train_step(sig, osd, ostate, 1e-3, p_drop=0.1)
            print("  oracle", ol, "rel", abs(ol / losses[-1] - 1), flush=True)
            assert abs(ol / losses[-1] - 1) < 2e-5, "oracle does not consume the dropout generator like the reference"
        print("step", s, losses[-1], diffs[-1], flush=True)
        np.savez_compressed(os.path.join(GOLD, "f5_curve_dropout.npz"), loss=np.array(losses), diff=np.array(diffs), B=B, lr=1e-3,
                            mask_seed_base=9000, dropout_seed_base=7000, pool=64, weight_seed=0, p_drop=0.1)


def f13_full_batch(ref_model, ref_learner, B=64):
    """Round-3 verdict, "B = 64 has no reference pin": the reference's own forward (code/model.py:519-601, no_grad) on the batch
    bench.py times - 64 two-channel 65 792-sample segments (synth.make_batch(5000, 16) rolled to 64, as PCM-16: what the loader ships),
    recipe weights, masks drawn by the reference from Python's RNG - in train mode with dropout p = 0 (BatchNorm batch statistics: the
    forward of the captured training step) and in eval mode: loss, diff, 4 096 sampled `pred` bins each."""
    from sar_ssl_amd import synth
    uniq = synth.make_batch(5000, 16)
    segs = np.stack([np.roll(uniq[i % 16], 997 * (i // 16), axis=0) for i in range(B)], axis=0)
    pcm = synth.to_pcm16(segs)
    sig = torch.from_numpy(pcm.astype(np.float32) / 32768.0)                      # what a PCM-16 WAV reader hands the reference
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    man = load_recipe(net, 0)
    set_dropout(net, 0.0)
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    x, = lrn.data_preprocess(sig, None)
    store = {"B": B, "weight_seed": 0, "sig_seed": 5000, "mask_seed": 31}
    for mode in ("train", "eval"):
        net.train(mode == "train")
        net.load_state_dict(recipes.recipe_state_dict(man, 0))
        random.seed(31)
        idx, ch = orc.gen_masks(B, 256, 128, 2, random)
        random.seed(31)
        with torch.no_grad():
            loss, diff, vis = net(x)
        pred_patch = vis["pred"].permute(0, 2, 1, 3, 4).contiguous()                # (B,T,F,reim,mic)
        sidx = sample_idx(pred_patch.numel(), 4096, 19)
        store.update({mode + ".loss": np.float64(loss.item()), mode + ".diff": np.float64(diff.item()), mode + ".pred_idx": sidx,
                      mode + ".pred_vals": pred_patch.reshape(-1)[sidx].numpy(), mode + ".pred_absmax": np.float64(pred_patch.abs().max())})
        print("f13", mode, "loss", loss.item(), "diff", diff.item(), flush=True)
        del vis, pred_patch
    store["mask_idx"], store["mask_ch"] = idx.numpy(), ch.numpy()
    np.savez_compressed(os.path.join(GOLD, "f13_full_batch.npz"), **store)


def f14_full_batch_gradient(ref_model, ref_learner, B=64):
    """The backward pass at the timed batch size, pinned on the reference itself: `loss.backward()` of the reference (train mode,
    dropout p = 0, masks from Python's RNG) on F13's batch - per-parameter gradient norms, 48 sampled entries of every parameter's
    gradient and two BatchNorm running statistics after the step's forward.  (~25 GB of autograd state, a few minutes of CPU.)"""
    from sar_ssl_amd import synth
    uniq = synth.make_batch(5000, 16)
    segs = np.stack([np.roll(uniq[i % 16], 997 * (i // 16), axis=0) for i in range(B)], axis=0)
    sig = torch.from_numpy(synth.to_pcm16(segs).astype(np.float32) / 32768.0)
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    man = load_recipe(net, 0)
    set_dropout(net, 0.0)
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    x, = lrn.data_preprocess(sig, None)
    net.train()
    net.load_state_dict(recipes.recipe_state_dict(man, 0))
    net.zero_grad()
    random.seed(31)
    idx, ch = orc.gen_masks(B, 256, 128, 2, random)
    random.seed(31)
    loss, diff, vis = net(x)
    del vis
    loss.backward()
    names, offs, sidx, svals, gn = [], [0], [], [], {}
    for k, p in net.named_parameters():
        g = p.grad.reshape(-1)
        gn[k] = float(g.double().norm())
        si = sample_idx(g.numel(), min(48, g.numel()), 23)
        names.append(k); sidx.append(si); svals.append(g[si].numpy().copy()); offs.append(offs[-1] + len(si))
    store = {"B": B, "weight_seed": 0, "sig_seed": 5000, "mask_seed": 31, "loss": np.float64(loss.item()), "diff": np.float64(diff.item()),
             "gradnorm_json": np.array(json.dumps(gn)), "names_json": np.array(json.dumps(names)), "sample_offsets": np.array(offs),
             "sample_idx": np.concatenate(sidx), "sample_vals": np.concatenate(svals), "mask_idx": idx.numpy(), "mask_ch": ch.numpy()}
    for k in ("spec_encoder.patch_embed.4.running_mean", "spat_encoder.embed.layers.1.sequential.2.module.sequential.5.running_var"):
        store["after." + k] = net.state_dict()[k].numpy().copy()
    print("f14 loss", loss.item(), "total grad norm", sum(v * v for v in gn.values()) ** 0.5, flush=True)
    np.savez_compressed(os.path.join(GOLD, "f14_full_batch_gradient.npz"), **store)


def f15_edge_cases(ref_model, ref_learner):
    """Round-4 verdict: every fixture used well-scaled input.  The reference's OWN forward + backward (train mode, dropout 0, fp32 CPU)
    on inputs at the edge of the front-end's normalisation (code/learner.py:525-553): a reference microphone 40 / 60 dB below the
    other one, an all-zero reference channel (scale = 1e-6 -> inputs ~1e6), a full-scale clipped PCM recording.  Stored per case: loss,
    diff, 2 048 sampled `pred` bins, per-parameter gradient norms."""
    net = ref_model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    man = load_recipe(net, 0)
    set_dropout(net, 0.0)
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    store = {}
    for name, sig in recipes.edge_case_signals().items():
        x, = lrn.data_preprocess(sig, None)
        net.train()
        net.load_state_dict(recipes.recipe_state_dict(man, 0))
        net.zero_grad()
        random.seed(4321)
        idx, ch = orc.gen_masks(2, 256, 128, 2, random)
        random.seed(4321)
        loss, diff, vis = net(x)
        loss.backward()
        pred_patch = vis["pred"].permute(0, 2, 1, 3, 4).contiguous()
        sidx = sample_idx(pred_patch.numel(), 2048, 13)
        gn = {k: float(p.grad.double().norm()) for k, p in net.named_parameters()}
        finite = bool(torch.isfinite(loss)) and all(np.isfinite(v) for v in gn.values())
        store.update({name + ".loss": np.float64(loss.item()), name + ".diff": np.float64(diff.item()), name + ".pred_idx": sidx,
                      name + ".pred_vals": pred_patch.reshape(-1)[sidx].detach().numpy(), name + ".pred_absmax": np.float64(pred_patch.abs().max()),
                      name + ".input_absmax": np.float64(x.abs().max()), name + ".gradnorm_json": np.array(json.dumps(gn)),
                      name + ".finite": np.bool_(finite)})
        print("f15", name, "loss", loss.item(), "diff", diff.item(), "max|x|", float(x.abs().max()), "max|pred|", float(pred_patch.abs().max()),
              "finite", finite, flush=True)
        # the oracle restates it
        with torch.no_grad():
            ol, od, _ = orc.sarssl_pretrain_forward(orc.data_preprocess(sig), recipes.recipe_state_dict(man, 0), idx, ch, train=True, p_drop=0.0,
                                                    return_pred=False)
        print("    oracle loss rel", abs(float(ol) / loss.item() - 1.0), flush=True)
    store["mask_idx"], store["mask_ch"] = idx.numpy(), ch.numpy()
    store["cases_json"] = np.array(json.dumps(list(recipes.edge_case_signals().keys())))
    np.savez_compressed(os.path.join(GOLD, "f15_edge_cases.npz"), **store)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", action="store_true")
    ap.add_argument("--curve-dropout", action="store_true", help="F5(ii): 40 dropout-on steps (~4 min CPU)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    ref_model, ref_learner, ref_um = ref_shim.load()
    todo = a.only.split(",") if a.only else ["manifest", "f1", "f2", "f3", "f4", "f6", "f7", "f8", "f9", "f10", "f11", "f12", "f13", "f14", "f15"]
    if "manifest" in todo: f_manifest(ref_model)
    if "f1" in todo: f1_frontend(ref_learner, ref_model)
    if "f2" in todo: f2_blocks(ref_model)
    if "f3" in todo: f3_fullsize(ref_model, ref_learner)
    if "f4" in todo: f4_masks(ref_um)
    if "f7" in todo: f7_downstream(ref_model)
    if "f8" in todo: f8_schedule()
    if "f9" in todo: f9_downstream_train(ref_model, ref_learner)
    if "f10" in todo: f10_multich(ref_model, ref_learner)
    if "f11" in todo: f11_eval_export(ref_model, ref_learner, ref_um)
    if "f12" in todo: f12_pretrain_epoch(ref_model, ref_learner)
    if "f6" in todo: f6_checkpoint(ref_model, ref_learner)
    if "f13" in todo: f13_full_batch(ref_model, ref_learner)
    if "f14" in todo: f14_full_batch_gradient(ref_model, ref_learner)
    if "f15" in todo: f15_edge_cases(ref_model, ref_learner)
    if a.curve: f5_curve(ref_model, ref_learner)
    if a.curve_dropout: f5_curve_dropout(ref_model, ref_learner)
    print("golden vectors written to", GOLD)
