"""GPU parity of the HIP modules / full model against golden vectors from the reference (and the oracle).

Tolerances: 'fp32' = split-bf16 precise path, gated at the north_star's 1e-3 (we assert tighter where the margin
allows); 'bf16' = the fast path (bf16 storage + MFMA inputs), gated at looser, explicitly stated tolerances.
"""
import json
import os

import numpy as np
import pytest
import torch

import recipes
import sarssl_oracle as orc
from conftest import GOLD, check

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _relerr(a, b):
    a = torch.as_tensor(np.asarray(a.detach().float().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _set_dropout(m, p):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = p


# bf16 gradient tolerance is wide at these toy sizes (B*T = 48 rows, 384 stem pixels): BatchNorm/LayerNorm backward subtracts
# batch means of bf16-rounded tensors; the full-size bf16 test below is the meaningful fast-path gate.
TOL = {"fp32": dict(y=2e-4, dx=5e-4, g=1e-3), "bf16": dict(y=3e-2, dx=1.2e-1, g=6e-1, gall=1e-1),
       # fp16 forward / bf16 backward: outputs 8x closer than bf16's, gradients bf16-class
       "fp16": dict(y=4e-3, dx=1.2e-1, g=6e-1, gall=1e-1),
       # hybrid: f32 stream, fp16-pair products - outputs in the f32 class behind the fp16 stem, gradients between the two
       "hybrid": dict(y=4e-3, dx=1.2e-1, g=6e-1, gall=1e-1)}


def _run_block(name, build, seed, prec, call=None, check_dx=True, residual=False):
    from sar_ssl_amd import runtime
    runtime.set_precision(prec)
    try:
        dev = _dev()
        z = _npz("f2_blocks.npz")
        meta = json.loads(str(z["meta_json"]))[name]
        tol = TOL[prec]
        for mode in ("eval", "train"):
            mod = build()
            missing = mod.load_state_dict(recipes.recipe_state_dict(meta, seed), strict=True)
            _set_dropout(mod, 0.0)
            mod.to(dev).train(mode == "train")
            x = torch.from_numpy(z[name + ".x"]).to(dev).requires_grad_(True)
            gy = torch.from_numpy(z[name + ".gy"]).to(dev)
            y = call(mod, x) if call else mod(x)
            (y.float() * gy).sum().backward()
            yref, dxref = z["%s.%s.y" % (name, mode)], z["%s.%s.dx" % (name, mode)]
            if residual:          # module called in its fused x + f(x) form (what the Conformer block uses)
                yref, dxref = yref + z[name + ".x"], dxref + z[name + ".gy"]
            assert _relerr(y, yref) < tol["y"], (name, mode, "y")
            if check_dx:
                assert _relerr(x.grad, dxref) < tol["dx"], (name, mode, "dx")
            num = den = 0.0
            for k, p in mod.named_parameters():
                ref = z["%s.%s.grad.%s" % (name, mode, k)]
                assert p.grad is not None, k
                e = _relerr(p.grad, ref) if np.abs(ref).max() > 1e-6 else float(p.grad.abs().max())
                assert e < tol["g"], (name, mode, k, e)
                num += float(((p.grad.double().cpu() - torch.from_numpy(ref).double()) ** 2).sum())
                den += float((torch.from_numpy(ref).double() ** 2).sum())
            if "gall" in tol:     # whole-gradient relative L2 error (bf16: individual cancellation-dominated entries are noisy)
                assert (num / den) ** 0.5 < tol["gall"], (name, mode, (num / den) ** 0.5)
            if mode == "train":
                for k, v in mod.state_dict().items():
                    if k.endswith(("running_mean", "running_var")):
                        assert _relerr(v, z["%s.train.after.%s" % (name, k)]) < (1e-4 if prec == "fp32" else 2e-2), k
    finally:
        runtime.set_precision("bf16")


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "hybrid"])
def test_ffn_module(prec):
    from sar_ssl_amd.common.conformer.feed_forward import FeedForwardModule
    _run_block("ffn", lambda: FeedForwardModule(encoder_dim=32, expansion_factor=4, dropout_p=0.1), 21, prec,
               call=lambda m, x: m.forward_residual(x, 1.0), residual=True)
    if prec == "fp32":            # plain (non-residual) reference signature
        _run_block("ffn", lambda: FeedForwardModule(encoder_dim=32, expansion_factor=4, dropout_p=0.1), 21, prec)


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "hybrid"])
def test_mhsa_module(prec):
    from sar_ssl_amd.common.conformer.attention import MultiHeadedSelfAttentionModule
    _run_block("mhsa", lambda: MultiHeadedSelfAttentionModule(d_model=32, num_heads=4, dropout_p=0.1), 22, prec,
               call=lambda m, x: m.forward_residual(x), residual=True)
    if prec == "fp32":
        _run_block("mhsa", lambda: MultiHeadedSelfAttentionModule(d_model=32, num_heads=4, dropout_p=0.1), 22, prec)


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "hybrid"])
def test_conv_module(prec):
    from sar_ssl_amd.common.conformer.convolution import ConformerConvModule
    mk = lambda: ConformerConvModule(in_channels=32, kernel_size=31, expansion_factor=2, dropout_p=0.1)
    _run_block("convmod", mk, 23, prec, call=lambda m, x: m.forward_residual(x), residual=True)
    if prec == "fp32":
        _run_block("convmod", mk, 23, prec)


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "hybrid"])
def test_conformer_block_and_encoder(prec):
    from sar_ssl_amd.common.Conformer import ConformerBlock, ConformerEncoder
    _run_block("block", lambda: ConformerBlock(encoder_dim=32, num_attention_heads=4), 24, prec)
    _run_block("block_T40", lambda: ConformerBlock(encoder_dim=32, num_attention_heads=4), 26, prec)
    _run_block("encoder2", lambda: ConformerEncoder(encoder_dim=32, num_layers=2, num_attention_heads=4), 25, prec,
               call=lambda m, x: m(x, False))


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "hybrid"])
def test_embed_encoder_decoder(prec):
    from sar_ssl_amd import model
    _run_block("embed_encoder", lambda: model.EmbedEncoder(sig_shape=[16, 8, 2, 2], patch_shape=(16, 1), dembed=32,
                                                           model=["cnn", "conformer"], mode="spat", device="cuda"),
               27, prec, call=lambda m, x: m.forward(x), check_dx=False)
    _run_block("embed_decoder", lambda: model.EmbedDecoder(sig_shape=[16, 8, 2, 2], patch_shape=(16, 1), dembed=48,
                                                           model=["", "fc"]), 28, prec, call=lambda m, x: m.forward(x))


def _is_stem_param(name):
    """Parameters of the CNN stem below the frame-patch product (patch_embed.0 ... patch_embed.10: 1x1 / 3x3 convolutions and BatchNorms)."""
    import re
    m = re.search(r"patch_embed\.(\d+)\.", name)
    return m is not None and int(m.group(1)) < 12


def _is_cancelling_bias(name):
    return name.endswith(("attention.u_bias", "attention.v_bias"))


def _check_gradnorms(net, gn, rtol, tag="gradnorm", body_rtol=None):
    """Per-parameter gradient L2 norms vs the reference.  Gradients that are analytically zero (e.g. the key-projection
    bias: softmax is invariant to it) are round-off in the reference too, so they get an absolute bound instead."""
    top = max(gn.values())
    report = []
    for k, p in net.named_parameters():
        got = p.grad.double().norm().item()
        if gn[k] < 1e-6 * top:
            assert got < 1e-4 * top, (k, got, gn[k])
        else:
            report.append((abs(got - gn[k]) / gn[k], k))
    worst = max(report)
    print("GRADNORM top5 %s: %s" % (tag, ", ".join("%s=%.2e" % (k, e) for e, k in sorted(report, reverse=True)[:5])))
    body = [r for r in report if not _is_stem_param(r[1]) and not _is_cancelling_bias(r[1])]
    if body:
        print("GRADNORM top3 of the f32-stream parameters %s: %s" % (tag, ", ".join("%s=%.2e" % (k, e) for e, k in sorted(body, reverse=True)[:3])))
    check("%s[worst=%s]" % (tag, worst[1]), worst[0], rtol)
    if body_rtol is not None and body:      # hybrid mode: parameters whose gradient flows along the f32 stream (outside the bf16 stem backward
        wb = max(body)                      # and the attention biases' near-cancelling column sums) at their own, tighter gate
        check("%s_body[worst=%s]" % (tag, wb[1]), wb[0], body_rtol)
    return worst


def _fullsize(prec, mode):
    from sar_ssl_amd import model, runtime, hip
    runtime.set_precision(prec)
    try:
        dev = _dev()
        z = _npz("f3_fullsize.npz")
        man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
        net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device=dev)
        net.load_state_dict(recipes.recipe_state_dict(man, 0))
        _set_dropout(net, 0.0)
        net.to(dev).train(mode == "train")
        x = hip.stft_frontend(recipes.recipe_signal(2, 65792, 2, seed=3).to(dev))
        net.set_masks(z["mask_idx"], z["mask_ch"])
        loss, diff, vis = net(x)
        loss.backward()
        return net, loss, diff, vis, z
    finally:
        runtime.set_precision("bf16")


# Full-size tolerances.  fp32 (split-bf16 MFMA) path: the north_star's 1e-3.  bf16 path (what bench.py times): 2.5-5x the deviation
# measured on MI355X (round 2: loss 3.6e-4, sampled pred 9.6e-3 of range, worst per-parameter gradient norm 2.6e-2, BN running
# stats 9.6e-4; recorded by conftest.check in gpurun_out/parity_measured.jsonl).
# (round 4: the gates live in sar_ssl_amd/parity.py, which bench.py's `parity_class` prints - the line claims what is asserted here.
#  fp16 forward / bf16 backward, the timed mode: per-bin max 9.1e-4 / 1.21e-3 of range (eval / train) as oracle/operand_rounding_study.py
#  predicted on the CPU (8.9e-4 / 1.10e-3); gated at the north_star's 1e-3 x 1.5, the rms of the same deviations at 5e-4.)
def _full_tol(prec):
    from sar_ssl_amd.parity import GATES
    g = GATES[prec]
    return dict(loss=g["loss"], pred=g["per_bin_max"], pred_rms=g["per_bin_rms"], grad=g["grad_norm"], bn=g["bn_running"], grad_body=g.get("grad_norm_body"))


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp32_1pass", "fp16", "hybrid"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_fullsize_forward_backward(mode, prec):
    """north_star gate: loss and per-bin outputs within 1e-3 relative of the reference CPU path (fp32 mode); the bf16 fast path is
    gated at a small multiple of its measured deviation."""
    tol = _full_tol(prec)
    net, loss, diff, vis, z = _fullsize(prec, mode)
    tag = "fullsize.%s.%s." % (prec, mode)
    check(tag + "loss", abs(loss.item() / float(z[mode + ".loss"]) - 1), tol["loss"])
    check(tag + "diff", abs(diff.item() / float(z[mode + ".diff"]) - 1), 1e-4)       # diff only involves the f32 front-end
    pred = vis["pred"].permute(0, 2, 1, 3, 4).reshape(-1).cpu()          # back to (B,T,F,reim,mic) order
    got = pred[torch.from_numpy(z[mode + ".pred_idx"])]
    want = torch.from_numpy(z[mode + ".pred_vals"])
    check(tag + "pred", ((got - want).abs().max() / float(z[mode + ".pred_absmax"])).item(), tol["pred"])
    check(tag + "pred_rms", ((got - want).pow(2).mean().sqrt() / float(z[mode + ".pred_absmax"])).item(), tol["pred_rms"])
    gn = json.loads(str(z[mode + ".gradnorm_json"]))
    _check_gradnorms(net, gn, tol["grad"], tag + "gradnorm", body_rtol=tol["grad_body"])
    if mode == "train":
        sd = net.state_dict()
        for k in ("spec_encoder.patch_embed.4.running_mean", "spec_encoder.patch_embed.4.running_var",
                  "spat_encoder.embed.layers.1.sequential.2.module.sequential.5.running_var"):
            check(tag + "bn." + k, _relerr(sd[k], z["train.after." + k]), tol["bn"])


def test_lazy_vis_and_eval_nograd():
    from sar_ssl_amd import model, runtime, hip
    dev = _dev()
    net = model.SARSSL(sig_shape=(16, 8, 2, 2), patch_shape=(16, 1), pretrain=True, device=dev).to(dev).eval()
    x = torch.randn((3, 2, 16, 8, 2), device=dev)
    with torch.no_grad():
        loss, diff, vis = net(x)
    assert set(vis.keys()) == {"mask", "pred", "tar"}
    assert vis["pred"].shape == (3, 16, 8, 2, 2) and vis["tar"].shape == (3, 16, 8, 2, 2) and vis["mask"].shape == (3, 16, 8, 2)
    assert torch.isfinite(loss) and float((vis["mask"] == 0).float().mean()) == 0.25
    with pytest.raises(Exception):
        net(x.cpu())                       # no CPU fallback


def test_downstream_forward():
    from sar_ssl_amd import model, runtime
    dev = _dev()
    runtime.set_precision("fp32")
    try:
        z = _npz("f7_downstream.npz")
        man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["downstream"]
        ds = model.SARSSL(sig_shape=(256, 64, 2, 2), pretrain=False, device=dev, downstream_token="all", downstream_head="mlp",
                          downstream_embed="spat", downstream_dlabel=1)
        ds.load_state_dict(recipes.recipe_state_dict(man, 5))
        ds.to(dev).eval()
        x = torch.from_numpy(np.random.default_rng(99).standard_normal((2, 2, 256, 64, 2)).astype(np.float32)).to(dev)
        with torch.no_grad():
            pred, emb = ds(x)
        assert _relerr(pred, z["pred"]) < 1e-3 and _relerr(emb, z["embed"]) < 1e-3
    finally:
        runtime.set_precision("bf16")


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16"])
def test_reloaded_weights_are_picked_up_without_flat_params(prec):
    """Re-laid-out weight caches (3x3 taps, patch-GEMM weight, bf16 shadows) must follow ``load_state_dict`` / in-place updates on a
    model that is NOT flattened: forward -> load_state_dict -> forward equals a fresh model carrying the second weights."""
    from sar_ssl_amd import model, runtime
    runtime.set_precision(prec)
    try:
        dev = _dev()
        z = _npz("f2_blocks.npz")
        meta = json.loads(str(z["meta_json"]))["embed_encoder"]
        mk = lambda: model.EmbedEncoder(sig_shape=[16, 8, 2, 2], patch_shape=(16, 1), dembed=32, model=["cnn", "conformer"], mode="spat",
                                        device="cuda")
        x = torch.from_numpy(z["embed_encoder.x"]).to(dev)
        a = mk(); a.load_state_dict(recipes.recipe_state_dict(meta, 27)); a.to(dev).eval()
        with torch.no_grad():
            y_first = a(x).float().clone()
            a.load_state_dict(recipes.recipe_state_dict(meta, 31))
            y_second = a(x).float().clone()
            for p in a.parameters():                              # in-place update through torch (what torch.optim does)
                p.mul_(1.25)
            y_third = a(x).float().clone()
        b = mk(); b.load_state_dict(recipes.recipe_state_dict(meta, 31)); b.to(dev).eval()
        with torch.no_grad():
            want_second = b(x).float().clone()
            for p in b.parameters():
                p.mul_(1.25)
            want_third = mk_out = b(x).float().clone()
        assert not torch.allclose(y_first, y_second)
        assert torch.equal(y_second, want_second) and torch.equal(y_third, want_third)
    finally:
        runtime.set_precision("bf16")


# ---------------------------------------------------------------- inputs at the edge of the front-end's normalisation (fixture F15)
F15_CASES = ["ref_mic_minus40dB", "ref_mic_minus60dB", "ref_mic_all_zero", "clipped_full_scale_pcm"]


def _f15(prec, case):
    from sar_ssl_amd import model, runtime, hip
    runtime.set_precision(prec)
    dev = _dev()
    z = _npz("f15_edge_cases.npz")
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device=dev)
    net.load_state_dict(recipes.recipe_state_dict(man, 0))
    _set_dropout(net, 0.0)
    net.to(dev).train()
    x = hip.stft_frontend(recipes.edge_case_signals()[case].to(dev))
    net.set_masks(z["mask_idx"], z["mask_ch"])
    loss, diff, vis = net(x)
    loss.backward()
    return net, loss, diff, vis, z, x


@pytest.mark.parametrize("prec", ["fp32", "fp16", "hybrid", "bf16"])
@pytest.mark.parametrize("case", F15_CASES)
def test_edge_case_inputs_vs_the_reference(case, prec):
    """Round-4 verdict: every fixture used well-scaled input.  F15 = the reference's own forward + backward (train mode, fp32 CPU) on a
    reference microphone 40 / 60 dB below the other one (inputs up to 650 / 6 500 after the normalisation of code/learner.py:539-542), an
    all-zero reference channel (normaliser = 1e-6: inputs up to 1e7) and a full-scale clipped PCM recording.  fp32 mode: the north_star
    gates on every case.  bf16: its usual class on every case (bf16 has f32's range).  fp16 forward: its usual class where the inputs fit
    fp16's range (|x| < 65 504: all but the all-zero reference channel) - and on that case the forward OVERFLOWS: the loss is not
    finite, which is what the device-side guard of the optimizer step keys on (test_nonfinite_loss_skips_the_optimizer_step)."""
    from sar_ssl_amd import runtime
    try:
        net, loss, diff, vis, z, x = _f15(prec, case)
        tol = _full_tol(prec)
        tag = "f15.%s.%s." % (case, prec)
        assert abs(float(x.abs().max()) / float(z[case + ".input_absmax"]) - 1) < 1e-4           # the f32 front-end itself is exact on every case
        check(tag + "diff", abs(diff.item() / float(z[case + ".diff"]) - 1), 1e-4)
        if prec in ("fp16", "hybrid") and case == "ref_mic_all_zero":          # (hybrid: the stem - and the network input - are fp16)
            assert float(z[case + ".input_absmax"]) > 65504.0 and not np.isfinite(loss.item()), loss.item()
            return
        check(tag + "loss", abs(loss.item() / float(z[case + ".loss"]) - 1), tol["loss"])
        pred = vis["pred"].permute(0, 2, 1, 3, 4).reshape(-1).cpu()
        got, want = pred[torch.from_numpy(z[case + ".pred_idx"])], torch.from_numpy(z[case + ".pred_vals"])
        check(tag + "pred", ((got - want).abs().max() / float(z[case + ".pred_absmax"])).item(), tol["pred"])
        check(tag + "pred_rms", ((got - want).pow(2).mean().sqrt() / float(z[case + ".pred_absmax"])).item(), tol["pred_rms"])
        # worst per-parameter gradient norm: the first 1x1 convolution's weight (4 -> 64 on the raw input) on the cases with a 100x / 1000x /
        # 1e7x louder second channel - measured fp16 3.9-4.9e-2 (-40 dB), 1.8e-2 (-60 dB), bf16 5.1e-2 / 4.6e-2 / 8.8e-2 (all-zero reference);
        # gated at 2.5x the mode's usual class there, at the usual class on the clipped recording
        gtol = tol["grad"] * (2.5 if case.startswith("ref_mic") and prec != "fp32" else 1.0)
        _check_gradnorms(net, json.loads(str(z[case + ".gradnorm_json"])), gtol, tag + "gradnorm", body_rtol=tol["grad_body"])
    finally:
        runtime.set_precision("bf16")


@pytest.mark.parametrize("form", ["captured", "eager"])
def test_nonfinite_loss_skips_the_optimizer_step(form):
    """fp16 forward on the all-zero-reference-channel input of F15 (|x| ~ 1e7 > 65 504): the loss is not finite.  The reference's fp16
    autocast path carries a GradScaler that skips such a step (code/learner.py:105-108); here the Adam launch reads the step's loss on
    the device and skips the update - parameters, moments and 16-bit shadow copies bit-identical, BatchNorm running statistics still
    finite, the step counted as skipped, the gradient buffer cleared - and the NEXT (well-scaled) step is an ordinary first Adam step:
    same parameters as a run that never saw the bad batch.  Both step forms: the captured step (device-resident step count) and the
    launch-by-launch learner step (FusedAdam)."""
    from sar_ssl_amd import hip, model, runtime
    from sar_ssl_amd.graph import PretrainStepGraph
    dev = _dev()
    runtime.set_precision("fp16")
    try:
        T, B = 16, 2
        nsample = 512 + 256 * (T - 1)
        good = recipes.recipe_signal(B, nsample, 2, seed=8).to(dev)
        bad = good.clone()
        bad[:, :, 0] = 0.0
        idx = np.stack([np.arange(0, T, 2) for _ in range(B)])
        ch = np.array([0, 1])

        def make():
            torch.manual_seed(3)
            net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
            _set_dropout(net, 0.0)
            net.to(dev).train()
            return net, runtime.FlatParams(net)

        def run(sigs):
            net, flat = make()
            p0 = flat.flat.clone()
            losses = []
            if form == "captured":
                g = PretrainStepGraph(net, flat, lr=1e-3)
                for s in sigs:
                    net.set_masks(idx, ch)
                    losses.append(float(g.step(x=hip.stft_frontend(s))[0]))
                torch.cuda.synchronize()
                return net, flat, p0, losses, g.skipped_steps(), g
            opt = runtime.FusedAdam(flat, lr=1e-3)
            opt.zero_grad()
            for s in sigs:
                net.set_masks(idx, ch)
                loss, _, _ = net(hip.stft_frontend(s))
                loss.backward()
                opt.step(guard=loss.detach())
                opt.zero_grad()
                losses.append(float(loss))
            torch.cuda.synchronize()
            return net, flat, p0, losses, int(opt.nskipped.item()), opt

        net, flat, p0, losses, nskip, o = run([bad])
        assert not np.isfinite(losses[0]) and nskip == 1
        assert torch.equal(flat.flat, p0) and float(o.m.abs().max()) == 0.0 and float(o.v.abs().max()) == 0.0
        assert torch.equal(flat.wh16.float(), p0.half().float()) and float(flat.grad.abs().max()) == 0.0
        for k, b in net.named_buffers():
            assert bool(torch.isfinite(b.float()).all()), k                     # BatchNorm running statistics did not absorb the overflow
        # bad batch followed by a good one == the good one alone (captured form: the device step count was taken back)
        net_a, flat_a, _, la, nskip_a, _ = run([bad, good])
        net_b, flat_b, _, lb, nskip_b, _ = run([good])
        assert nskip_a == 1 and nskip_b == 0 and np.isfinite(la[1]) and la[1] == lb[0]
        if form == "captured":
            assert torch.equal(flat_a.flat, flat_b.flat)
        else:       # FusedAdam's host-side step count advanced over the skipped step: bias corrections of step 2 instead of step 1
            check("guard.eager.param_delta_rel", float((flat_a.flat - flat_b.flat).norm() / (flat_b.flat - p0).norm()), 0.5)
    finally:
        runtime.set_precision("bf16")


@pytest.mark.parametrize("prec", ["fp16", "hybrid", "bf16"])
def test_decoder_on_the_masked_frames_only_equals_the_full_decoder(prec, monkeypatch):
    """Training steps run EmbedDecoder (code/model.py:321-334) - and the row-wise tail of each encoder's last Conformer block: second
    feed-forward module + closing LayerNorm (code/common/Conformer.py:84-90) - on the masked frames only: gen_loss (code/model.py:721-747)
    reads nothing else and these layers treat frames separately.  Against the full-frame decoder (SARSSL_DEC_MASKED=0) on the same inputs, weights and
    masks: the loss bit for bit (same rows, same arithmetic), every parameter's gradient to summation-order noise (the decoder's weight
    gradients contract over half the rows - the other half were exact zeros), and vis["pred"] - formed on request from the step's decoder
    input - equal to the full decoder's prediction."""
    from sar_ssl_amd import engine, hip, model, runtime
    dev = _dev()
    runtime.set_precision(prec)
    try:
        T, B = 32, 4
        nsample = 512 + 256 * (T - 1)
        sig = recipes.recipe_signal(B, nsample, 2, seed=4).to(dev)
        g = np.random.default_rng(3)
        idx = np.stack([g.choice(T, T // 2, replace=False) for _ in range(B)])        # unsorted on purpose: the model sorts
        ch = g.integers(0, 2, size=B)
        res = {}
        for masked in (True, "decoder_only", False):
            monkeypatch.setattr(engine, "_DEC_MASKED", bool(masked))
            monkeypatch.setattr(engine, "_TAIL_MASKED", masked is True)      # True: decoder + the row-wise tails of the encoders' last blocks
            torch.manual_seed(9)
            net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
            _set_dropout(net, 0.0)
            net.to(dev).train()
            flat = runtime.FlatParams(net)
            net.set_masks(idx, ch)
            loss, diff, vis = net(hip.stft_frontend(sig))
            loss.backward()
            res[masked] = (float(loss), float(diff), flat.grad.clone(), vis["pred"].clone(), vis["mask"].clone())
        b = res[False]
        for key in (True, "decoder_only"):
            a = res[key]
            assert a[0] == b[0] and a[1] == b[1]
            assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
            check("dec_masked.%s.%s.grad_rel_l2" % (prec, key), float((a[2] - b[2]).norm() / b[2].norm()), 2e-3)
            check("dec_masked.%s.%s.grad_max_over_max" % (prec, key), float((a[2] - b[2]).abs().max() / b[2].abs().max()), 5e-3)
        # dropout ON (round-5 verdict): vis["pred"] of a compact training step at the frames that entered the loss is the step's OWN
        # prediction - the rows its decoder ran on - bit for bit (the reference clones the step's pred, code/model.py:595-599); the loss of
        # those very rows is the returned loss
        for masked in (True, "decoder_only"):
            monkeypatch.setattr(engine, "_DEC_MASKED", True)
            monkeypatch.setattr(engine, "_TAIL_MASKED", masked is True)
            torch.manual_seed(9)
            net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
            net.to(dev).train()                                        # dropout 0.1 everywhere
            runtime.FlatParams(net)
            runtime.RT.manual_seed(77)
            net.set_masks(idx, ch)
            x = hip.stft_frontend(sig)
            loss, diff, vis = net(x)
            pv = vis["pred"]                                           # (B, F, T, reim, mic), formed now
            sidx = torch.from_numpy(np.sort(idx, axis=1)).to(dev)
            mch = torch.from_numpy(ch).to(dev)
            got = pv.permute(0, 2, 1, 3, 4)[torch.arange(B, device=dev)[:, None], sidx]           # (B, nm, F, reim, mic)
            got = got[torch.arange(B, device=dev), :, :, :, mch]                                   # masked channel: (B, nm, F, reim)
            tar = x.permute(0, 3, 2, 4, 1)[torch.arange(B, device=dev)[:, None], sidx]            # x (B,mic,F,T,reim) -> (B, nm, F, reim, mic)
            tar = tar[torch.arange(B, device=dev), :, :, :, mch]
            relo = abs(float(((got.float() - tar.float()) ** 2).mean()) / float(loss) - 1)
            check("dec_masked.%s.%s.dropout_on.loss_of_vis_rows_vs_step_loss" % (prec, masked), relo, 1e-5 if prec != "bf16" else 1e-4)
    finally:
        runtime.set_precision("bf16")
