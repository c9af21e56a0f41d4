"""Deterministic weight recipes for parity tests (TEST INFRASTRUCTURE ONLY).

Full-size parity fixtures would need a 70 MB weight blob; instead every tensor is generated
from ``numpy.random.default_rng(crc32(key) + seed)`` so the reference (in the build
container), the oracle and the HIP path (on the GPU box) all see bit-identical weights
without any file travelling.  Shapes come from a key->shape manifest
(``tests/golden/state_dict_manifest.json``, dumped from the real reference by
``oracle/make_golden.py``).
"""
import math
import zlib

import numpy as np
import torch


def pe_table(d_model, max_len=10000):
    """PositionalEncoding buffer (code/common/conformer/embedding.py:31-39), shape (1, max_len, d)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def _rng(key, seed):
    return np.random.default_rng((zlib.crc32(key.encode()) + 1000003 * seed) & 0xFFFFFFFF)


def recipe_tensor(key, shape, seed=0):
    shape = tuple(shape)
    g = _rng(key, seed)
    if key.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    if key.endswith("positional_encoding.pe"):
        return pe_table(shape[2], shape[1])
    if key.endswith("running_mean"):
        return torch.from_numpy(g.uniform(-0.1, 0.1, shape).astype(np.float32))
    if key.endswith("running_var"):
        return torch.from_numpy(g.uniform(0.5, 1.5, shape).astype(np.float32))
    if len(shape) == 1:
        # norm gains (LayerNorm/BatchNorm '.weight' that are 1-D) near 1, biases small non-zero
        is_gain = key.endswith(".weight")
        lo, hi = (0.8, 1.2) if is_gain else (-0.05, 0.05)
        return torch.from_numpy(g.uniform(lo, hi, shape).astype(np.float32))
    fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
    fan_in = shape[1] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[1]
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return torch.from_numpy(g.uniform(-a, a, shape).astype(np.float32))


def recipe_state_dict(manifest, seed=0):
    """manifest: {key: [shape...]} -> {key: tensor} in manifest order."""
    return {k: recipe_tensor(k, shp, seed) for k, shp in manifest.items()}


def recipe_signal(nbatch, nsample, nch=2, seed=0):
    """Seeded white-ish test signal with cross-channel structure, (B, nsample, nch) float32."""
    g = np.random.default_rng(977 + seed)
    src = g.standard_normal((nbatch, nsample + 16)).astype(np.float32)
    out = np.empty((nbatch, nsample, nch), dtype=np.float32)
    for c in range(nch):
        d = 3 * c + 1
        out[:, :, c] = 0.6 * src[:, d:d + nsample] + 0.3 * src[:, d + 2:d + 2 + nsample] \
            + 0.1 * g.standard_normal((nbatch, nsample)).astype(np.float32)
    return torch.from_numpy(out * 0.2)


def edge_case_signals():
    """Inputs of fixture F15 (shared with tests/): recipe signal seed 3 (as F3), B = 2 full-size segments, modified per case.
    Reference microphone = channel 0: data_preprocess divides by mean|X0| + 1e-6 (code/learner.py:539-542)."""
    base = recipe_signal(2, 65792, 2, seed=3)
    cases = {}
    for name, att in (("ref_mic_minus40dB", 1e-2), ("ref_mic_minus60dB", 1e-3)):
        s = base.clone()
        s[:, :, 0] *= att                                   # a quiet reference microphone next to a loud second one
        cases[name] = s
    s = base.clone()
    s[:, :, 0] = 0.0                                        # an all-zero reference channel: the normaliser is its 1e-6 epsilon
    cases["ref_mic_all_zero"] = s
    s = (base * (8.0 / base.abs().max())).clamp(-1.0, 32767.0 / 32768.0)      # full-scale clipped recording, as PCM-16 hands it over
    cases["clipped_full_scale_pcm"] = torch.round(s * 32768.0).clamp(-32768, 32767) / 32768.0
    return cases
