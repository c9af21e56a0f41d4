"""GPU: SURVEY.md 8f-2 / BASELINE config 5 - 4-mic recordings: 'M' and 'MM' pairing in the fused front-end, a 10 s segment
(3 mic pairs, T = 624 frames) through the pretraining forward/backward, and the multi-pair head SARSSL_MultiCH, against the real
reference (fixture F10)."""
import json
import os

import numpy as np
import pytest
import torch

import recipes
from conftest import GOLD, check
from test_gpu_model import _check_gradnorms, _relerr, _set_dropout

pytestmark = pytest.mark.gpu


def _z():
    return np.load(os.path.join(GOLD, "f10_multich.npz"), allow_pickle=False)


def test_frontend_all_pairs_mode_mm():
    from sar_ssl_amd import hip
    z = _z()
    sig = recipes.recipe_signal(2, 1536, 4, seed=2).cuda()
    out = hip.stft_frontend(sig, ch_mode="MM")
    assert tuple(out.shape) == (12, 2, 256, 5, 2)
    assert _relerr(out, torch.from_numpy(z["mm_small4_out"])) < 1e-5
    pcm = (sig * 32767).round().clamp(-32768, 32767).to(torch.int16)                  # int16 PCM input, same pairing
    out_i = hip.stft_frontend(pcm, ch_mode="MM")
    assert _relerr(out_i, hip.stft_frontend(pcm.float() / 32768.0, ch_mode="MM").cpu()) < 1e-5


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "hybrid"])
def test_config5_ten_second_four_mic_segment(prec):
    """T = 624 is not a multiple of the conv tile (64) or the GEMM tile: ragged tiles everywhere."""
    from sar_ssl_amd import hip, model, runtime
    z = _z()
    tol = {"fp32": (1e-3, 1e-3, 5e-3), "bf16": (1e-3, 3e-2, 6e-2), "fp16": (1e-3, 4e-3, 4e-2), "hybrid": (1e-3, 4e-3, 4e-2)}[prec]       # bf16: 3-5x measured (5.3e-5, 1.0e-2, 1.9e-2)
    runtime.set_precision(prec)
    try:
        man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
        net = model.SARSSL(sig_shape=(256, 624, 2, 2), pretrain=True, device="cuda:0")
        net.load_state_dict(recipes.recipe_state_dict(man, 0))
        _set_dropout(net, 0.0)
        net.cuda().train()
        x = hip.stft_frontend(recipes.recipe_signal(1, 160000, 4, seed=21).cuda())
        assert tuple(x.shape) == (3, 2, 256, 624, 2)
        net.set_masks(z["c5.mask_idx"], z["c5.mask_ch"])
        loss, diff, vis = net(x)
        loss.backward()
        check("config5.%s.loss" % prec, abs(loss.item() / float(z["c5.loss"]) - 1), tol[0])
        check("config5.%s.diff" % prec, abs(diff.item() / float(z["c5.diff"]) - 1), 1e-4)
        pred = vis["pred"].permute(0, 2, 1, 3, 4).reshape(-1).cpu()
        got, want = pred[torch.from_numpy(z["c5.pred_idx"])], torch.from_numpy(z["c5.pred_vals"])
        check("config5.%s.pred" % prec, ((got - want).abs().max() / float(z["c5.pred_absmax"])).item(), tol[1])
        _check_gradnorms(net, json.loads(str(z["c5.gradnorm_json"])), tol[2], "config5.%s.gradnorm" % prec)
    finally:
        runtime.set_precision("bf16")


def test_multich_head_forward():
    from sar_ssl_amd import model, runtime
    z = _z()
    runtime.set_precision("fp32")
    try:
        man = json.loads(str(z["mch.manifest_json"]))
        mch = model.SARSSL_MultiCH(sig_shape=(256, 32, 2, 2), nmic_pair=3, task="TDOA", device="cuda:0")
        assert {k: list(v.shape) for k, v in mch.state_dict().items()} == man        # checkpoint-compatible keys / shapes
        mch.load_state_dict(recipes.recipe_state_dict(man, 11))
        mch.cuda().eval()
        xm = torch.from_numpy(np.random.default_rng(5).standard_normal((6, 2, 256, 32, 2)).astype(np.float32)).cuda()
        with torch.no_grad():
            pred, emb = mch(xm)
        assert _relerr(pred, z["mch.pred"]) < 1e-3 and _relerr(emb, z["mch.embed"]) < 1e-3
    finally:
        runtime.set_precision("bf16")
