"""Live check of the oracle against the REAL reference (only where /root/reference exists, i.e. the build container).
The committed golden vectors (tests/test_oracle_golden.py) are the portable form of the same pin."""
import random

import numpy as np
import pytest
import torch

import ref_shim
import recipes
import sarssl_oracle as orc

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


def test_oracle_matches_reference_small_pretrain_step():
    ref_model, ref_learner, ref_um = ref_shim.load()
    T = 8
    net = ref_model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device="cpu")
    man = {k: list(v.shape) for k, v in net.state_dict().items()}
    sd = recipes.recipe_state_dict(man, 3)
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.train()
    lrn = ref_learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cpu()
    sig = recipes.recipe_signal(3, 512 + 256 * (T - 1), 2, seed=9)
    x_ref, = lrn.data_preprocess(sig, None)
    x = orc.data_preprocess(sig)
    assert torch.allclose(x, x_ref, rtol=1e-5, atol=1e-6)
    random.seed(11)
    idx, ch = orc.gen_masks(3, T, T // 2, 2, random)
    random.seed(11)
    loss_ref, diff_ref, _ = net(x_ref)
    loss_ref.backward()
    osd = recipes.recipe_state_dict(man, 3)
    params = {k: v.requires_grad_(True) for k, v in osd.items() if orc.is_param(k)}
    loss, diff, _ = orc.sarssl_pretrain_forward(x, osd, idx, ch, train=True, p_drop=0.0)
    loss.backward()
    assert abs(loss.item() / loss_ref.item() - 1) < 1e-5 and abs(diff.item() / diff_ref.item() - 1) < 1e-6
    top = max(p.grad.abs().max().item() for p in net.parameters())
    for k, p in net.named_parameters():
        ref = p.grad
        got = params[k].grad
        scale = max(ref.abs().max().item(), 1e-6 * top)      # analytically-zero grads (key bias) are round-off on both sides
        assert (got - ref).abs().max().item() / scale < 1e-2, k          # tiny batch (6144 stem pixels): f32 summation-order noise through BatchNorm
