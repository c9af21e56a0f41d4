"""GPU: training-loop level parity - the 100-step loss curve of the reference (fixture F5: replayed masks, dropout off),
learner epoch / checkpoint round trip, and the end-to-end entry point on pre-generated WAV segments (BASELINE config 1)."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

import recipes
from conftest import GOLD, ROOT

pytestmark = pytest.mark.gpu


def _set_dropout(m, p):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = p


def _curve(prec, nstep):
    from sar_ssl_amd import hip, model, runtime, synth
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLD, "f5_curve.npz"))
    B = int(z["B"])
    runtime.set_precision(prec)
    try:
        man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
        net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device=dev)
        net.load_state_dict(recipes.recipe_state_dict(man, int(z["weight_seed"])))
        _set_dropout(net, 0.0)
        net.to(dev).train()
        flat = runtime.FlatParams(net)
        opt = runtime.FusedAdam(flat, lr=float(z["lr"]))
        opt.zero_grad()
        pool = torch.from_numpy(synth.make_batch(0, int(z["pool"]))).to(dev)
        losses, diffs = [], []
        for s in range(nstep):
            sig = pool[(s * B) % 64:(s * B) % 64 + B]
            x = hip.stft_frontend(sig)
            random.seed(int(z["mask_seed_base"]) + s)
            loss, diff, _ = net(x)
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(loss.detach()); diffs.append(diff.detach())
        return torch.stack(losses).cpu().numpy().astype(np.float64), torch.stack(diffs).cpu().numpy().astype(np.float64), z
    finally:
        runtime.set_precision("bf16")


def test_loss_curve_100_steps_vs_reference_fp32():
    """north_star: 100-step reconstruction-loss curve within 1e-3 (relative) of the reference CPU run."""
    got, gdiff, z = _curve("fp32", 100)
    ref = z["loss"][:100]
    rel = np.abs(got - ref) / ref
    print("max rel dev over 100 steps: %.3e (first 10: %.3e)" % (rel.max(), rel[:10].max()))
    assert np.abs(gdiff - z["diff"][:100]).max() / z["diff"].max() < 1e-4       # 'diff' depends on data + masks only
    assert rel[:10].max() < 1e-3
    assert rel.max() < 1e-3


def test_loss_curve_bf16_tracks_reference():
    """Fast path (bf16 storage): stated tolerance 5 % per step over the first 30 steps of the same curve."""
    got, _, z = _curve("bf16", 30)
    ref = z["loss"][:30]
    rel = np.abs(got - ref) / ref
    print("bf16 max rel dev over 30 steps: %.3e" % rel.max())
    assert rel.max() < 5e-2


def test_learner_epoch_and_checkpoint_roundtrip(tmp_path):
    from sar_ssl_amd import learner as L, model, runtime, synth
    dev = torch.device("cuda:0")
    T = 8
    nsample = 512 + 256 * (T - 1)
    net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
    lrn = L.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn.cuda()
    lrn.amp()
    data = torch.from_numpy(synth.make_batch(0, 8, nsample=nsample))
    loader = [[data[i:i + 4]] for i in (0, 4)]
    random.seed(0)
    loss1, diff1, vis = lrn.pretrain_epoch(loader, lr=1e-3, epoch=1)
    loss2, _, _ = lrn.pretrain_epoch(loader, lr=1e-3, epoch=2)
    assert np.isfinite(loss1) and np.isfinite(loss2) and diff1 > 0 and vis["pred"].shape == (4, 256, T, 2, 2)
    lv, dv, _ = lrn.pretest_epoch(loader)
    assert np.isfinite(lv)
    stop, best = lrn.early_stopping(-lv, patience=100)
    assert best and not stop
    lrn.save_checkpoint(epoch=2, checkpoints_dir=str(tmp_path), is_best_epoch=True, save_extra_hist=True)
    ck = torch.load(str(tmp_path / "best_model.tar"), map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"epoch", "max_score", "model"} and ck["epoch"] == 2
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    assert [k for k in ck["model"].keys()] == [k for k in man.keys()]
    # load into a fresh learner, also from a DataParallel-style 'module.'-prefixed checkpoint
    ck2 = dict(ck); ck2["model"] = {"module." + k: v for k, v in ck["model"].items()}
    torch.save(ck2, str(tmp_path / "latest_model.tar"))
    net2 = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev)
    lrn2 = L.STFTLearner(net2, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
    lrn2.cuda()
    lrn2.resume_checkpoint(str(tmp_path), from_latest=True)
    assert lrn2.start_epoch == 3
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    random.seed(5); l_a = lrn.pretest_epoch(loader)[0]
    random.seed(5); l_b = lrn2.pretest_epoch(loader)[0]       # lrn2 is fp32-mode (no .amp()): close, not identical
    assert abs(l_a - l_b) / abs(l_b) < 3e-2
    runtime.set_precision("bf16")


def test_run_pretrain_entry_point_on_wav_segments(tmp_path):
    """BASELINE config 1 shape of the plumbing: pre-generated 2-mic WAV segments, batch 8, one epoch through the CLI."""
    from sar_ssl_amd import dataset, synth
    work = tmp_path / "work"
    for split, n, base in (("pretrain", 16, 0), ("preval", 8, 500)):
        d = work / "SAR-SSL" / "data" / "MicSig" / "simu" / split
        d.mkdir(parents=True)
        pcm = synth.to_pcm16(synth.make_batch(base, n))
        for i in range(n):
            dataset.write_wav_pcm16(str(d / ("%d.wav" % i)), pcm[i])
    cmd = [sys.executable, os.path.join(ROOT, "run_pretrain.py"), "--pretrain", "--simu-exp", "--gpu-id", "0,", "--work-dir", str(work),
           "--bs", "8", "8", "8", "--nepoch", "2", "--workers", "2", "--time", "t0", "--use-amp"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    logd = work / "SAR-SSL" / "exp" / "pretrain" / "t0"
    rec = json.loads(open(logd / "scalars.jsonl").read().strip().splitlines()[-1])
    assert rec["epoch"] == 2 and np.isfinite(rec["loss_train"]) and np.isfinite(rec["loss_val"]) and rec["lr"] >= 0
    assert (logd / "latest_model.tar").exists() and (logd / "best_model.tar").exists() and (logd / "model1.tar").exists()
    assert (logd / "config.json").exists()


def test_two_rank_bench_path_over_gloo():
    """The data-parallel step (stage hooks, bucketed all-reduce, max-over-ranks timing) with 2 ranks sharing this GPU over gloo -
    a functional check of the code path the 8-GPU RCCL run uses (RCCL itself needs one GPU per rank)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SARSSL_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["scaling"] == "weak" and out["value"] > 0
    assert np.isfinite(out["final_loss"]) and "cpu_baseline" not in out
