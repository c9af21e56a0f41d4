"""GPU: SURVEY.md 8f-3 - the eval / export path: HIP inverse STFT (both centre modes, odd channel counts), STFT -> ISTFT round
trip, pretest_epoch(return_eval=True) -> pretrain_evaluate against the real reference (fixture F11), `run_pretrain.py --test`."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

import recipes
import sarssl_oracle as orc
from conftest import GOLD, ROOT, check
from test_gpu_model import _relerr

pytestmark = pytest.mark.gpu


def _z():
    return np.load(os.path.join(GOLD, "f11_eval_export.npz"), allow_pickle=False)


@pytest.mark.parametrize("inv", [False, True])
def test_istft_vs_reference_fixture(inv):
    from sar_ssl_amd.common import utils_module as um
    z = _z()
    spec = torch.view_as_complex(torch.from_numpy(z["spec"])).cuda()
    out = um.ISTFT(win_len=512, win_shift_ratio=0.5, nfft=512, inv=inv)(spec)
    ref = z["istft_inv%d" % int(inv)]
    assert tuple(out.shape) == ref.shape
    assert _relerr(out, ref) < 1e-5


@pytest.mark.parametrize("nch,nt", [(1, 1), (2, 37), (4, 64), (5, 3)])
def test_istft_vs_oracle_ragged_shapes(nch, nt):
    """frame counts that do not fill a 16-frame workgroup, odd channel counts (unpaired last channel)."""
    from sar_ssl_amd import hip
    g = torch.Generator().manual_seed(nch * 100 + nt)
    spec = torch.view_as_complex(torch.randn((2, 257, nt, nch, 2), generator=g))
    for center in ([False, True] if nt >= 2 else [False]):
        got = hip.istft(spec.cuda(), center=center)
        assert _relerr(got, orc.istft(spec, inv=center)) < 1e-5


def test_stft_istft_round_trip_full_size():
    """size-independent property at BASELINE's segment length: Hann analysis + rectangular synthesis gives back
    x[n] * (w[n] + w[n+256]) / 2 = x[n] / 2 in the interior (periodic Hann at 50 % overlap sums to one)."""
    from sar_ssl_amd import hip
    sig = recipes.recipe_signal(4, 65792, 2, seed=8).cuda()
    spec = hip.stft_raw(sig)                                                       # (B, 257, 256, 2) complex
    back = hip.istft(spec, center=False)
    assert tuple(back.shape) == (4, 65792, 2)
    inner = slice(256, 65792 - 256)
    assert (back[:, inner] - 0.5 * sig[:, inner]).abs().max() < 2e-5 * sig.abs().max()


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "hybrid"])
def test_pretest_epoch_with_eval_vs_reference(prec):
    from sar_ssl_amd import learner, model, runtime
    z = _z()
    tol = {"fp32": (1e-3, 1e-3), "bf16": (1e-3, 2e-2), "fp16": (1e-3, 3e-3), "hybrid": (1e-3, 3e-3)}[prec]                    # bf16: 3-5x measured (2.1e-4, 6.3e-3)
    try:
        man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
        net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cuda:0")
        net.load_state_dict(recipes.recipe_state_dict(man, 0))
        lrn = learner.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
        lrn.cuda()
        if prec != "fp32":
            lrn.amp(prec)
        sig = recipes.recipe_signal(2, 65792, 2, seed=3)
        random.seed(2468)                                                          # masks come from Python's RNG, as in the reference
        loss, diff, vis, res = lrn.pretest_epoch([[sig]], return_diff=True, return_eval=True)
        check("pretest.%s.loss" % prec, abs(loss / float(z["eval.loss"]) - 1), tol[0])
        check("pretest.%s.diff" % prec, abs(diff / float(z["eval.diff"]) - 1), 1e-4)
        assert tuple(res["sig_pred"].shape) == tuple(z["eval.sig_shape"])
        sidx = torch.from_numpy(z["eval.sig_idx"])
        sp, st = res["sig_pred"].reshape(-1).cpu()[sidx], res["sig_tar"].reshape(-1).cpu()[sidx]
        check("pretest.%s.sig_pred" % prec, (sp - torch.from_numpy(z["eval.sig_pred"])).abs().max(), tol[1])   # waveforms are normalised to max 1
        assert (st - torch.from_numpy(z["eval.sig_tar"])).abs().max() < 1e-4
        for k in ("mse", "mse_mask", "mse_mask_ch"):
            check("pretest.%s.%s" % (prec, k), abs(float(res[k]) / float(z["eval." + k]) - 1), tol[0])
        assert tuple(res["pesq"].shape) == (2, 2)
    finally:
        runtime.set_precision("bf16")


def test_run_pretrain_test_modes(tmp_path):
    """`run_pretrain.py --test --simu-exp` with --test-mode all (loss over the pretest set) and ins (per-instance export)."""
    import scipy.io
    from sar_ssl_amd import dataset, model, synth
    work = tmp_path / "work"
    for split, n, base in (("pretest", 6, 0), ("pretest_ins_T1000", 2, 50)):
        d = work / "SAR-SSL" / "data" / "MicSig" / "simu" / split
        d.mkdir(parents=True)
        pcm = synth.to_pcm16(synth.make_batch(base, n))
        for i in range(n):
            dataset.write_wav_pcm16(str(d / ("%d.wav" % i)), pcm[i])
            if "ins" in split:
                dataset.write_wav_pcm16(str(d / ("%d_dp.wav" % i)), pcm[i])
    ck = work / "SAR-SSL" / "exp" / "pretrain" / "t2"
    ck.mkdir(parents=True)
    net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    torch.save({"epoch": 7, "max_score": -1.0, "model": net.state_dict()}, str(ck / "best_model.tar"))
    base_cmd = [sys.executable, os.path.join(ROOT, "run_pretrain.py"), "--test", "--simu-exp", "--gpu-id", "0,", "--work-dir", str(work),
                "--bs", "4", "4", "4", "--workers", "2", "--time", "t2", "--use-amp"]
    r = subprocess.run(base_cmd + ["--test-mode", "all"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["epoch"] == 7 and np.isfinite(rec["loss_test"])
    r = subprocess.run(base_cmd + ["--test-mode", "ins"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = ck / "test_result"
    mat = scipy.io.loadmat(str(out / "rt1000_ins_epoch7_test.mat"))
    assert mat["pred"].shape == (2, 256, 256, 2, 2) and mat["mask"].shape == (2, 256, 256, 2)
    pcm, fs = dataset.read_wav_pcm16(str(out / "rt1000_ins1_epoch7_test_pred.wav"))
    assert fs == 16000 and pcm.shape == (65792, 2) and np.abs(pcm).max() > 1000
