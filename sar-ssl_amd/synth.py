"""Synthetic 2-channel (or N-channel) microphone segments for benchmarks and loss-curve tests.

There is no network / corpus here, so segments are generated: a
speech-like source (white noise through a random 2-pole resonator, slow random envelope)
reaches each microphone through a short impulse response (integer delay 0..9 samples =
+-0.2 m at 343 m/s, 16 kHz, plus a 64-tap exponentially decaying tail); independent white
noise is added at 15..30 dB SNR and the segment is scaled to peak 0.9, like the reference's
simulated data (code/data_generation/utils_simu_rir_sig.py:819-825).  The cross-channel
structure matters: on plain white noise the reconstruction loss plateaus immediately.
"""
import numpy as np

FS = 16000
NSAMPLE = 65792          # 4.112 s @ 16 kHz  (code/opt.py:16-21)


def make_segment(index, nsample=NSAMPLE, nch=2, seed=1234):
    g = np.random.default_rng(seed + int(index))
    n = nsample + 128
    w = g.standard_normal(n)
    # 2-pole resonator
    fc = g.uniform(150.0, 1200.0)
    r = g.uniform(0.90, 0.98)
    a1, a2 = -2.0 * r * np.cos(2 * np.pi * fc / FS), r * r
    from scipy.signal import lfilter
    s = lfilter([1.0], [1.0, a1, a2], w)
    # slow envelope
    env_pts = g.uniform(0.05, 1.0, size=n // 2048 + 2)
    env = np.interp(np.arange(n), np.arange(env_pts.size) * 2048, env_pts)
    s = s * env
    out = np.empty((nsample, nch), dtype=np.float64)
    sig_pow = None
    for c in range(nch):
        h = np.zeros(10 + 64)
        d = int(g.integers(0, 10))
        h[d] = 1.0
        tail = g.standard_normal(64) * np.exp(-np.arange(64) / 12.0) * 0.25
        h[10:] += tail
        x = np.convolve(s, h)[64:64 + nsample]
        if sig_pow is None:
            sig_pow = float(np.mean(x ** 2)) + 1e-12
        out[:, c] = x
    snr_db = g.uniform(15.0, 30.0)
    noise = g.standard_normal((nsample, nch)) * np.sqrt(sig_pow / (10 ** (snr_db / 10)))
    out = out + noise
    out = out * (0.9 / (np.max(np.abs(out)) + 1e-12))
    return out.astype(np.float32)


def make_batch(start, count, nsample=NSAMPLE, nch=2, seed=1234):
    """(count, nsample, nch) float32; segment i uses rng(seed + start + i)."""
    return np.stack([make_segment(start + i, nsample, nch, seed) for i in range(count)], axis=0)


def to_pcm16(batch):
    """float [-1,1) -> int16 PCM as soundfile's default 16-bit WAV subtype would store it."""
    return np.clip(np.round(batch * 32768.0), -32768, 32767).astype(np.int16)
