"""CPU: the oracle restatement (oracle/sarssl_oracle.py) against golden vectors produced by the
REAL reference (oracle/make_golden.py).  fp32 on both sides -> tight tolerances."""
import json
import os
import random

import numpy as np
import pytest
import torch

import recipes
import sarssl_oracle as orc
from conftest import GOLD


def _npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _close(a, b, rtol=1e-4, atol=1e-5):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    tol = atol + rtol * b.abs().max().item()
    assert err <= tol, "max err %.3e > tol %.3e" % (err, tol)


def test_f1_frontend_small_and_multichannel():
    z = _npz("f1_frontend.npz")
    small = recipes.recipe_signal(2, 2048, 2, seed=1)
    X = orc.stft(small)
    _close(X.real, z["small_stft_re"], 1e-5, 1e-5)
    _close(X.imag, z["small_stft_im"], 1e-5, 1e-5)
    _close(orc.data_preprocess(small), z["small_out"], 1e-5, 1e-6)
    small4 = recipes.recipe_signal(2, 1536, 4, seed=2)
    out4 = orc.data_preprocess(small4)
    assert out4.shape[0] == 2 * 3
    _close(out4, z["small4_out"], 1e-5, 1e-6)


def test_f1_frontend_fullsize_samples():
    z = _npz("f1_frontend.npz")
    out = orc.data_preprocess(recipes.recipe_signal(2, 65792, 2, seed=3))
    assert list(out.shape) == list(z["full_shape"]) == [2, 2, 256, 256, 2]
    _close(out.reshape(-1)[torch.from_numpy(z["full_idx"])], z["full_vals"], 1e-5, 1e-6)
    assert abs(out.double().sum().item() - float(z["full_sum"])) < 1e-2
    assert abs((out.double() ** 2).sum().item() / float(z["full_sumsq"]) - 1) < 1e-5


def test_relative_shift_tables():
    z = _npz("f2_blocks.npz")
    for T in (4, 5, 7):
        ps = torch.arange(T * T, dtype=torch.float32).reshape(1, 1, T, T)
        assert torch.equal(orc.relative_shift(ps)[0, 0], torch.from_numpy(z["relshift.T%d" % T]))
    assert orc.relative_shift(torch.arange(16.).reshape(4, 4)).tolist() == \
        [[3, 0, 4, 5], [6, 7, 0, 8], [9, 10, 11, 0], [12, 13, 14, 15]]


def _run_block(name, fn, seed):
    z = _npz("f2_blocks.npz")
    meta = json.loads(str(z["meta_json"]))[name]
    x0 = torch.from_numpy(z[name + ".x"])
    gy = torch.from_numpy(z[name + ".gy"])
    for mode in ("eval", "train"):
        sd = recipes.recipe_state_dict(meta, seed)
        params = {k: v.requires_grad_(True) for k, v in sd.items() if orc.is_param(k)}
        x = x0.clone().requires_grad_(True)
        y = fn(x, sd, mode == "train")
        (y * gy).sum().backward()
        _close(y.detach(), z["%s.%s.y" % (name, mode)], 2e-4, 2e-5)
        _close(x.grad, z["%s.%s.dx" % (name, mode)], 5e-4, 1e-5)
        for k, p in params.items():
            _close(p.grad, z["%s.%s.grad.%s" % (name, mode, k)], 1e-3, 2e-5)
        if mode == "train":
            for k in meta:
                if k.endswith(("running_mean", "running_var")):
                    _close(sd[k], z["%s.train.after.%s" % (name, k)], 1e-4, 1e-6)


def test_f2_ffn():
    _run_block("ffn", lambda x, sd, tr: orc.feed_forward(x, sd, "sequential.", 0.0, tr), 21)


def test_f2_mhsa():
    _run_block("mhsa", lambda x, sd, tr: orc.mhsa(x, sd, "", 4, 0.0, tr), 22)


def test_f2_conv_module():
    _run_block("convmod", lambda x, sd, tr: orc.conv_module(x, sd, "sequential.", 0.0, tr), 23)


def test_f2_block():
    _run_block("block", lambda x, sd, tr: orc.conformer_block(x, sd, "", 4, 0.0, tr), 24)
    _run_block("block_T40", lambda x, sd, tr: orc.conformer_block(x, sd, "", 4, 0.0, tr), 26)


def test_f2_encoder2():
    _run_block("encoder2", lambda x, sd, tr: orc.conformer_encoder(x, sd, "", 2, 4, 0.0, tr), 25)


def test_f2_embed_encoder_and_decoder():
    def enc(x, sd, tr):
        B, T, _ = x.shape
        return orc.embed_encoder(x.reshape(B, T, 16, 2, 2), sd, "", 3, tr, 0.0)
    _run_block("embed_encoder", enc, 27)
    _run_block("embed_decoder", lambda x, sd, tr: orc.decoder(x, sd, "proj."), 28)


def test_f4_mask_rng_order():
    z = _npz("f4_masks.npz")
    for seed in (0, 7, 123456):
        random.seed(seed)
        idx, ch = orc.gen_masks(4, 256, 128, 2, random)
        assert np.array_equal(idx.numpy(), z["seed%d.idx" % seed])
        assert np.array_equal(ch.numpy(), z["seed%d.ch" % seed][:, 0])


def test_f8_lr_schedule():
    z = _npz("f8_schedule.npz")
    got = [orc.cosine_lr(e, 30, 1e-3, 1) for e in range(1, 31)]
    np.testing.assert_allclose(got, z["lr"], rtol=1e-6)
    assert abs(got[0] - 1e-3) < 1e-9


def test_f7_downstream():
    z = _npz("f7_downstream.npz")
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["downstream"]
    sd = recipes.recipe_state_dict(man, 5)
    x = torch.from_numpy(np.random.default_rng(99).standard_normal((2, 2, 256, 64, 2)).astype(np.float32))
    with torch.no_grad():
        pred, emb = orc.sarssl_downstream_forward(x, sd, "spat", train=False)
    _close(pred, z["pred"], 5e-4, 1e-5)
    _close(emb, z["embed"], 5e-4, 1e-5)


@pytest.mark.parametrize("mode", ["finetune", "lineareval"])
def test_f9_downstream_training_steps(mode):
    """Three TDOA fine-tuning iterations of the oracle against the real reference's."""
    from sar_ssl_amd import synth
    z = _npz("f9_downstream_train.npz")
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["downstream"]
    sd = recipes.recipe_state_dict(man, int(z["weight_seed"]))
    B, lr = int(z["B"]), float(z["lr"])
    n = z[mode + ".loss"].shape[0]
    pool = torch.from_numpy(synth.make_batch(int(z["sig_seed"]), n * B))[:, :16640].contiguous()
    tdoa = torch.from_numpy(z["tdoa"])
    frozen = ("spec_encoder.", "spat_encoder.") if mode == "lineareval" else ()
    state = {}
    for s_ in range(n):
        loss, metric, pred, emb = orc.downstream_train_step(pool[s_ * B:(s_ + 1) * B], tdoa[s_ * B:(s_ + 1) * B], sd, state, lr,
                                                            embed_use="spat", p_drop=0.0, frozen=frozen)
        assert abs(loss - z[mode + ".loss"][s_]) <= 2e-3 * abs(z[mode + ".loss"][s_]), (s_, loss, z[mode + ".loss"][s_])
        assert abs(metric - z[mode + ".metric"][s_]) <= 2e-3 * abs(z[mode + ".metric"][s_])
        _close(pred, z[mode + ".pred"][s_], 2e-3, 1e-4)
        if s_ == 0:
            _close(emb, z[mode + ".embed0"], 5e-4, 1e-5)
            g = state["last_grads"]
            for k in man:
                key = "%s.gradnorm.%s" % (mode, k)
                if key not in z.files:
                    continue
                ref = float(z[key])
                if ref < 0:                                        # frozen or unused (spec branch with embed 'spat'): no gradient
                    assert k not in g or float(g[k].abs().max()) == 0.0
                else:
                    got = float(g[k].double().norm())
                    assert abs(got - ref) <= 2e-3 * ref + 1e-6 * max(1.0, ref) + 2e-4, (k, got, ref)
    # eval-mode pass over the same batches = what Learner.test_epoch averages (after the three updates above the
    # reference's test_epoch ran on its own train_epoch result, which took the same three steps)
    lo = me = 0.0
    for s_ in range(n):
        l, m, _, _ = orc.downstream_eval_step(pool[s_ * B:(s_ + 1) * B], tdoa[s_ * B:(s_ + 1) * B], sd, "spat")
        lo += l / n; me += m / n
    assert abs(lo - z[mode + ".test_epoch"][0]) <= 3e-3 * z[mode + ".test_epoch"][0]
    assert abs(me - z[mode + ".test_epoch"][1]) <= 3e-3 * z[mode + ".test_epoch"][1]
    assert abs(np.mean(z[mode + ".loss"]) - z[mode + ".train_epoch"][0]) <= 1e-3 * z[mode + ".train_epoch"][0]


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_f3_fullsize_loss_pred_grads(mode):
    z = _npz("f3_fullsize.npz")
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd = recipes.recipe_state_dict(man, 0)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if orc.is_param(k)}
    assert sum(p.numel() for p in params.values()) == 17534224
    x = orc.data_preprocess(recipes.recipe_signal(2, 65792, 2, seed=3))
    idx, ch = torch.from_numpy(z["mask_idx"]), torch.from_numpy(z["mask_ch"])
    loss, diff, aux = orc.sarssl_pretrain_forward(x, sd, idx, ch, train=(mode == "train"), p_drop=0.0)
    loss.backward()
    assert abs(loss.item() / float(z[mode + ".loss"]) - 1) < 1e-4          # north_star tolerance is 1e-3
    assert abs(diff.item() / float(z[mode + ".diff"]) - 1) < 1e-5
    pv = aux["pred"].detach().reshape(-1)[torch.from_numpy(z[mode + ".pred_idx"])]
    _close(pv, z[mode + ".pred_vals"], 1e-4 * 0 + 2e-4, 1e-5)
    gn = json.loads(str(z[mode + ".gradnorm_json"]))
    for k, p in params.items():
        ref = gn[k]
        assert abs(p.grad.double().norm().item() - ref) <= 2e-3 * ref + 1e-7, k
    if mode == "train":
        for k in ("spec_encoder.patch_embed.4.running_mean", "spec_encoder.patch_embed.4.running_var",
                  "spat_encoder.embed.layers.1.sequential.2.module.sequential.5.running_var"):
            _close(sd[k], z["train.after." + k], 1e-4, 1e-6)


def test_f10_multichannel_pairs_config5_and_multich_head():
    """'MM' pairing, a 4-mic 10 s segment (3 pairs, T = 624) through forward/backward, SARSSL_MultiCH."""
    z = _npz("f10_multich.npz")
    _close(orc.data_preprocess(recipes.recipe_signal(2, 1536, 4, seed=2), ch_mode="MM"), z["mm_small4_out"], 1e-5, 1e-6)
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd = recipes.recipe_state_dict(man, 0)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if orc.is_param(k)}
    x = orc.data_preprocess(recipes.recipe_signal(1, 160000, 4, seed=21))
    assert tuple(x.shape) == (3, 2, 256, 624, 2)
    loss, diff, aux = orc.sarssl_pretrain_forward(x, sd, torch.from_numpy(z["c5.mask_idx"]), torch.from_numpy(z["c5.mask_ch"]),
                                                  train=True, p_drop=0.0)
    loss.backward()
    assert abs(loss.item() / float(z["c5.loss"]) - 1) < 1e-4 and abs(diff.item() / float(z["c5.diff"]) - 1) < 1e-5
    _close(aux["pred"].detach().reshape(-1)[torch.from_numpy(z["c5.pred_idx"])], z["c5.pred_vals"], 2e-4, 1e-5)
    gn = json.loads(str(z["c5.gradnorm_json"]))
    for k, p in params.items():
        assert abs(p.grad.double().norm().item() - gn[k]) <= 2e-3 * gn[k] + 1e-7, k
    man_m = json.loads(str(z["mch.manifest_json"]))
    sdm = recipes.recipe_state_dict(man_m, 11)
    xm = torch.from_numpy(np.random.default_rng(5).standard_normal((6, 2, 256, 32, 2)).astype(np.float32))
    with torch.no_grad():
        pred, emb = orc.sarssl_multich_forward(xm, sdm, 3)
    _close(pred, z["mch.pred"], 5e-4, 1e-5)
    _close(emb, z["mch.embed"], 5e-4, 1e-5)


def test_f11_istft_and_pretrain_evaluate():
    """inverse STFT (both centre modes) and the eval-export metrics against the real reference."""
    z = _npz("f11_eval_export.npz")
    spec = torch.view_as_complex(torch.from_numpy(z["spec"]))
    _close(orc.istft(spec, inv=False), z["istft_inv0"], 1e-5, 1e-6)
    _close(orc.istft(spec, inv=True), z["istft_inv1"], 1e-5, 1e-6)
    assert orc.istft(spec, inv=False).shape == (2, 10 * 256, 3) and orc.istft(spec, inv=True).shape == (2, 8 * 256, 3)
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd = recipes.recipe_state_dict(man, 0)
    x = orc.data_preprocess(recipes.recipe_signal(2, 65792, 2, seed=3))
    idx, ch = torch.from_numpy(z["eval.mask_idx"]), torch.from_numpy(z["eval.mask_ch"])
    with torch.no_grad():
        loss, diff, aux = orc.sarssl_pretrain_forward(x, sd, idx, ch, train=False)
    assert abs(loss.item() / float(z["eval.loss"]) - 1) < 1e-4 and abs(diff.item() / float(z["eval.diff"]) - 1) < 1e-5
    B, T = 2, 256
    pred = aux["pred"].detach().view(B, T, 256, 2, 2).permute(0, 2, 1, 3, 4)               # (nb,nf,nt,nreim,nch)
    tar = x.permute(0, 2, 3, 4, 1)
    mask = torch.ones(B, 256, T, 2)
    for b in range(B):
        mask[b, :, idx[b], int(ch[b])] = 0.0
    res = orc.pretrain_evaluate(pred, tar, mask)
    assert tuple(res["sig_pred"].shape) == tuple(z["eval.sig_shape"])
    sidx = torch.from_numpy(z["eval.sig_idx"])
    _close(res["sig_pred"].reshape(-1)[sidx], z["eval.sig_pred"], 5e-4, 1e-5)
    _close(res["sig_tar"].reshape(-1)[sidx], z["eval.sig_tar"], 1e-4, 1e-6)
    for k in ("mse", "mse_mask", "mse_mask_ch"):
        assert abs(float(res[k]) / float(z["eval." + k]) - 1) < 2e-4, k


def test_oracle_pretrain_epoch_vs_reference_pretrain_epoch():
    """Fixture F12 (the reference's own ``Learner.pretrain_epoch``, two epochs x four batches): the oracle's restatement returns the
    same per-epoch (loss, diff) and the same total parameter update - which a carried-over Adam state would change by 15 %."""
    import random
    from sar_ssl_amd import synth
    z = np.load(os.path.join(GOLD, "f12_pretrain_epoch.npz"))
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd = recipes.recipe_state_dict(man, int(z["weight_seed"]))
    init = {k: v.clone() for k, v in sd.items()}
    B, nb = int(z["B"]), int(z["nbatch"])
    pool = torch.from_numpy(synth.make_batch(int(z["sig_seed"]), B * nb))
    batches = [pool[i * B:(i + 1) * B] for i in range(nb)]
    torch.set_num_threads(8)
    for e in (1, 2):
        random.seed(int(z["mask_seed"][e - 1]))
        loss, diff, pred = orc.pretrain_epoch(batches, sd, float(z["lr"][e - 1]), p_drop=0.0)
        assert abs(loss / float(z["epoch%d.loss" % e]) - 1) < 2e-4, (e, loss)
        assert abs(diff / float(z["epoch%d.diff" % e]) - 1) < 1e-6
        assert tuple(pred.shape) == tuple(z["epoch%d.pred_shape" % e])
    ref = json.loads(str(z["update_norm_json"]))
    tot_ref = sum(v * v for v in ref.values()) ** 0.5
    tot = sum(float((sd[k].double() - init[k].double()).norm()) ** 2 for k in ref) ** 0.5
    assert abs(tot / tot_ref - 1) < 1e-3


def test_oracle_dropout_on_steps_vs_reference_curve():
    """Fixture F5(ii): with the reference's seeds (python ``random`` for the masks, ``torch.manual_seed`` for the 28 dropout draws
    per step) the oracle reproduces the reference's dropout-ON losses - i.e. it consumes the generator in the same order and layouts."""
    import random
    from sar_ssl_amd import synth
    z = np.load(os.path.join(GOLD, "f5_curve_dropout.npz"))
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd, state = recipes.recipe_state_dict(man, int(z["weight_seed"])), {}
    B = int(z["B"])
    pool = torch.from_numpy(synth.make_batch(0, int(z["pool"])))
    torch.set_num_threads(8)
    for s in range(2):
        random.seed(int(z["mask_seed_base"]) + s)
        torch.manual_seed(int(z["dropout_seed_base"]) + s)
        loss, diff = orc.train_step(pool[s * B:(s + 1) * B], sd, state, float(z["lr"]), p_drop=float(z["p_drop"]))
        assert abs(loss / float(z["loss"][s]) - 1) < 2e-5, (s, loss)
        assert abs(diff / float(z["diff"][s]) - 1) < 1e-6


def test_f13_oracle_forward_at_the_timed_batch_size():
    """Fixture F13: the reference's own forward at B = 64 (the batch bench.py times), train mode with dropout 0.  The oracle on the same
    PCM-16 segments / recipe weights / masks reproduces loss, diff and the 4 096 sampled `pred` bins (train-mode BatchNorm couples the
    segments, so the whole batch is run: ~15 s on 8 threads)."""
    from sar_ssl_amd import synth
    z = _npz("f13_full_batch.npz")
    B = int(z["B"])
    uniq = synth.make_batch(int(z["sig_seed"]), 16)
    segs = np.stack([np.roll(uniq[i % 16], 997 * (i // 16), axis=0) for i in range(B)], axis=0)
    sig = torch.from_numpy(synth.to_pcm16(segs).astype(np.float32) / 32768.0)
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd = recipes.recipe_state_dict(man, int(z["weight_seed"]))
    random.seed(int(z["mask_seed"]))
    idx, ch = orc.gen_masks(B, 256, 128, 2, random)
    assert np.array_equal(idx.numpy(), z["mask_idx"]) and np.array_equal(ch.numpy(), z["mask_ch"])
    with torch.no_grad():
        loss, diff, aux = orc.sarssl_pretrain_forward(orc.data_preprocess(sig), sd, idx, ch, train=True, p_drop=0.0)
    assert abs(float(loss) / float(z["train.loss"]) - 1) < 1e-5 and abs(float(diff) / float(z["train.diff"]) - 1) < 1e-5
    got = aux["pred"].reshape(-1)[torch.from_numpy(z["train.pred_idx"])]
    assert float((got - torch.from_numpy(z["train.pred_vals"])).abs().max()) < 2e-4 * float(z["train.pred_absmax"])


def test_f14_fixture_is_the_backward_of_the_f13_batch():
    """Fixture F14 (the reference's `loss.backward()` at B = 64; consumed by the `-m gpu` tests): same batch, weights and masks as F13's
    train-mode forward - the losses agree to the bit - and one entry per parameter of the state-dict manifest.  (The oracle itself is
    pinned on gradients by F2 / F3; running its backward at B = 64 needs ~25 GB of autograd state and is left to the generator.)"""
    z13, z14 = _npz("f13_full_batch.npz"), _npz("f14_full_batch_gradient.npz")
    assert float(z14["loss"]) == float(z13["train.loss"]) and float(z14["diff"]) == float(z13["train.diff"])
    assert np.array_equal(z14["mask_idx"], z13["mask_idx"]) and np.array_equal(z14["mask_ch"], z13["mask_ch"])
    names = json.loads(str(z14["names_json"]))
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    params = [k for k in man if not k.endswith(("running_mean", "running_var", "num_batches_tracked", ".pe"))]
    assert names == params and len(z14["sample_offsets"]) == len(names) + 1 and int(z14["sample_offsets"][-1]) == len(z14["sample_vals"])
    gn = json.loads(str(z14["gradnorm_json"]))
    assert set(gn) == set(names) and all(np.isfinite(v) for v in gn.values())
    assert abs(sum(v * v for v in gn.values()) ** 0.5 - 2.8583218) < 1e-5


@pytest.mark.parametrize("case", ["ref_mic_minus40dB", "ref_mic_minus60dB", "ref_mic_all_zero", "clipped_full_scale_pcm"])
def test_f15_oracle_on_inputs_at_the_edge_of_the_normalisation(case):
    """Fixture F15 (round 5): the reference's own forward + backward on a reference microphone 40 / 60 dB below the other one, an
    all-zero reference channel (normaliser = its 1e-6 epsilon, inputs ~1e7) and a full-scale clipped PCM recording - the oracle must
    restate all of them (the reference stays finite in fp32 on every case)."""
    z = _npz("f15_edge_cases.npz")
    assert case in json.loads(str(z["cases_json"])) and bool(z[case + ".finite"])
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd = recipes.recipe_state_dict(man, 0)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if orc.is_param(k)}
    x = orc.data_preprocess(recipes.edge_case_signals()[case])
    assert abs(float(x.abs().max()) / float(z[case + ".input_absmax"]) - 1) < 1e-5
    loss, diff, aux = orc.sarssl_pretrain_forward(x, sd, torch.from_numpy(z["mask_idx"]), torch.from_numpy(z["mask_ch"]), train=True, p_drop=0.0)
    loss.backward()
    assert abs(loss.item() / float(z[case + ".loss"]) - 1) < 1e-4
    assert abs(diff.item() / float(z[case + ".diff"]) - 1) < 1e-5
    pv = aux["pred"].detach().reshape(-1)[torch.from_numpy(z[case + ".pred_idx"])]
    assert float((pv - torch.from_numpy(z[case + ".pred_vals"])).abs().max()) < 3e-4 * float(z[case + ".pred_absmax"])
    gn = json.loads(str(z[case + ".gradnorm_json"]))
    top = max(gn.values())
    for k, p in params.items():
        assert abs(p.grad.double().norm().item() - gn[k]) <= 3e-3 * gn[k] + 1e-6 * top, k
