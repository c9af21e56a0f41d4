"""CPU tests of the host side: C-ABI exports, state_dict layout, mask RNG order, schedule, WAV I/O, flat-parameter
bookkeeping, and that the product path refuses to run without a GPU (no CPU fallback)."""
import json
import os
import random
import re

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT


def test_cabi_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from sar_ssl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "sarssl_hip.h")).read()
    names = sorted(set(re.findall(r"\b(sarssl_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert lib.sarssl_abi_version() == 1


def test_state_dict_layout_matches_reference_manifest():
    from sar_ssl_amd import model
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device="cpu")
    sd = net.state_dict()
    assert list(sd.keys()) == list(man["pretrain"].keys())
    assert all(list(v.shape) == man["pretrain"][k] for k, v in sd.items())
    assert sum(p.numel() for p in net.parameters()) == man["nparams_pretrain"] == 17534224
    ds = model.SARSSL(sig_shape=(256, 64, 2, 2), pretrain=False, device="cpu", downstream_embed="spat")
    assert list(ds.state_dict().keys()) == list(man["downstream"].keys())
    # sinusoid table identical to the reference's persistent buffer recipe
    import recipes
    assert torch.equal(sd["spec_encoder.embed.layers.0.sequential.1.module.positional_encoding.pe"], recipes.pe_table(512))


def test_no_cpu_fallback():
    from sar_ssl_amd import model, _lib
    net = model.SARSSL(sig_shape=(16, 8, 2, 2), patch_shape=(16, 1), pretrain=True, device="cpu")
    with pytest.raises(_lib.SarsslHipError):
        net(torch.randn(2, 2, 16, 8, 2))
    from sar_ssl_amd.common.Conformer import ConformerBlock
    with pytest.raises(_lib.SarsslHipError):
        ConformerBlock(encoder_dim=32, num_attention_heads=4)(torch.randn(2, 16, 32))
    from sar_ssl_amd import learner
    with pytest.raises(_lib.SarsslHipError):
        learner.STFTLearner(net, 512, 0.5, 512, 1, 16000).cpu()


def test_patch_mask_rng_order_matches_reference():
    from sar_ssl_amd.common.utils_module import PatchMask
    z = np.load(os.path.join(GOLD, "f4_masks.npz"))
    for seed in (0, 7, 123456):
        pm = PatchMask(patch_mode="T", nmasked_patch=128, npatch_shape=[1, 256], device="cpu")
        random.seed(seed)
        idx, ch = pm.sample(4, 2)
        assert np.array_equal(idx, z["seed%d.idx" % seed]) and np.array_equal(ch, z["seed%d.ch" % seed][:, 0])
        random.seed(seed)
        md, mpd, mcd, idx_t, ch_t = pm.forward((4, 256, 8, 2, 2))       # dense API form
        assert np.array_equal(idx_t.numpy(), idx) and md.shape == (4, 256, 8, 2)
        b = 1
        masked_frames = (mpd[b, :, 0, 0] == 0).nonzero().flatten().numpy()
        assert np.array_equal(np.sort(idx[b]), masked_frames)
        assert float(mcd[b, 0, 0, int(ch[b])]) == 0.0 and float(mcd[b, 0, 0, 1 - int(ch[b])]) == 1.0
        assert float(md.sum()) == 4 * 256 * 8 * 2 - 4 * 128 * 8


def test_lr_schedule_and_opt():
    from sar_ssl_amd.common.utils import create_learning_rate_schedule
    from sar_ssl_amd.opt import opt_pretrain
    z = np.load(os.path.join(GOLD, "f8_schedule.npz"))
    fn = create_learning_rate_schedule(total_steps=30, base=0.001, decay_type="cosine", warmup_steps=1, linear_end=1e-6)
    np.testing.assert_allclose([float(fn(e)) for e in range(1, 31)], z["lr"], rtol=1e-6)
    o = opt_pretrain()
    a = o.parse(["--pretrain", "--simu-exp", "--gpu-id", "0,", "--work-dir", "/tmp/w"])
    assert a.bs == [128, 128, 128] and a.lr == 0.001 and a.nepoch == 30 and a.seed == 1 and a.workers == 8
    assert o.dir()["micsig_simu_pretrain"] == "/tmp/w/SAR-SSL/data/MicSig/simu/pretrain"
    with pytest.raises(AssertionError):
        opt_pretrain().parse(["--pretrain", "--test"])


def test_wav_dataset_roundtrip(tmp_path):
    from sar_ssl_amd import dataset, synth
    segs = synth.make_batch(0, 3, nsample=4096)
    pcm = synth.to_pcm16(segs)
    for i in range(3):
        dataset.write_wav_pcm16(str(tmp_path / ("%d.wav" % i)), pcm[i])
    dataset.write_wav_pcm16(str(tmp_path / "0_dp.wav"), pcm[0])            # must be ignored
    ds = dataset.FixMicSigDataset(str(tmp_path), fs=16000, load_anno=False, dataset_sz=None)
    assert len(ds) == 3
    got = {tuple(ds[i][0].shape) for i in range(3)}
    assert got == {(4096, 2)} and ds[0][0].dtype == np.float32
    names = [os.path.basename(str(f)) for f in ds.files]
    k = names.index("1.wav")
    np.testing.assert_array_equal(ds[k][0], pcm[1].astype(np.float32) / 32768.0)
    raw = dataset.FixMicSigDataset(str(tmp_path), fs=16000, load_anno=False, dataset_sz=2, raw_pcm=True)
    assert len(raw) == 2 and raw[0][0].dtype == np.int16


def test_flat_params_bookkeeping_cpu():
    from sar_ssl_amd import runtime
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    ref = [p.detach().clone() for p in m.parameters()]
    flat = runtime.FlatParams(m)
    assert not flat.on_gpu and flat.numel % 8 == 0
    for p, r, o in zip(m.parameters(), ref, flat.offsets):
        assert torch.equal(p.data, r) and o % 8 == 0
        assert p.data.data_ptr() == flat.flat.data_ptr() + 4 * o and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
    m(torch.randn(4, 5)).sum().backward()                      # autograd accumulates into the flat views
    assert float(flat.grad.abs().sum()) > 0
    flat.zero_grad()
    assert float(flat.grad.abs().sum()) == 0
    sl = flat.bucket_slices(2)
    assert sl[0][0] == 0 and sl[-1][1] == flat.numel and all(a[1] == b[0] for a, b in zip(sl[:-1], sl[1:]))


def test_synth_segments_are_deterministic():
    from sar_ssl_amd import synth
    a, b = synth.make_segment(5, nsample=2048), synth.make_segment(5, nsample=2048)
    assert np.array_equal(a, b) and a.shape == (2048, 2) and abs(np.abs(a).max() - 0.9) < 1e-6
    assert not np.array_equal(a, synth.make_segment(6, nsample=2048))
