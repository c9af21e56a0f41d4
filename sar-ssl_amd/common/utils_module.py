"""Front-end transform and patch/mask plumbing with the reference's interfaces
(code/common/utils_module.py: STFT :28-72, ISTFT :74-113, AddChToBatch :116-148, PatchSplit :175-207, PatchRecover :210-244,
PatchMask :247-308)."""
import random

import numpy as np
import torch
import torch.nn as nn

from .. import hip


class STFT(nn.Module):
    """signal (nbatch, nsample, nch) -> complex64 (nbatch, nf, nt, nch); periodic Hann, center=False, computed by the
    fused 2-channel HIP FFT kernel (csrc/stft.hip).  Only win_len = nfft = 512, hop = 256 (the reference's values,
    code/run_pretrain.py:67-69) are implemented."""

    def __init__(self, win_len, win_shift_ratio, nfft, win="hann", inv=False):
        super().__init__()
        self.win_len, self.win_shift_ratio, self.nfft, self.win, self.inv = win_len, win_shift_ratio, nfft, win, inv
        if win != "hann" or inv:
            raise NotImplementedError("only the forward Hann STFT of the pretraining path is implemented")

    def forward(self, signal):
        return hip.stft_raw(signal, self.win_len, int(self.win_len * self.win_shift_ratio), self.nfft)


class ISTFT(nn.Module):
    """stft complex (nbatch, nf = nfft/2+1, nt, nch) -> signal (nbatch, nsample, nch): ``torch.istft`` with the default
    rectangular window, center = ``inv`` (code/common/utils_module.py:74-113), computed by the paired inverse FFT + overlap-add
    kernels in csrc/stft.hip.  nsample = (nt+1)*hop (inv=False) or (nt-1)*hop (inv=True)."""

    def __init__(self, win_len, win_shift_ratio, nfft, inv=False):
        super().__init__()
        self.win_len, self.win_shift_ratio, self.nfft, self.inv = win_len, win_shift_ratio, nfft, inv

    def forward(self, stft):
        return hip.istft(stft, center=self.inv, win_len=self.win_len, hop=int(self.win_len * self.win_shift_ratio), nfft=self.nfft)


class AddChToBatch(nn.Module):
    """(nb, nch, ...) -> (nb*(nch-1), 2, ...) pairing every mic with reference mic 0 ('M'), all pairs ('MM') or
    identity ('1').  Pure indexing (no arithmetic): done with torch index ops; the fused front-end
    (hip.stft_frontend) emits the 'M' pairing directly and never calls this."""

    def __init__(self, ch_mode):
        super().__init__()
        self.ch_mode = ch_mode
        assert self.ch_mode in ["MM", "M", "1"], "Unrecognized microphone channel mode~"

    def forward(self, data):
        nb, nch = data.shape[0], data.shape[1]
        if self.ch_mode == "M":
            ref = data[:, 0:1].expand(nb, nch - 1, *data.shape[2:])
            return torch.stack([ref, data[:, 1:]], dim=2).reshape(nb * (nch - 1), 2, *data.shape[2:]).contiguous()
        if self.ch_mode == "MM":
            i0, i1 = zip(*[(a, b) for a in range(nch - 1) for b in range(a + 1, nch)])
            out = torch.stack([data[:, list(i0)], data[:, list(i1)]], dim=2)
            return out.reshape(nb * len(i0), 2, *data.shape[2:]).contiguous()
        return data.clone().contiguous()


class PatchSplit(nn.Module):
    """With frame patches (patch_shape = (nf, 1)) unfold is a pure permutation:
    (nb, nf, nt, nreim, nmic) -> (nb, npatch = nt, dpatch = nf, nreim, nmic)."""

    def __init__(self, patch_shape, f_first=False):
        super().__init__()
        self.patch_shape, self.f_first = patch_shape, f_first
        if f_first or patch_shape[1] != 1:
            raise NotImplementedError("only frame patches (nf, 1) are on the pretraining path")

    def forward(self, data):
        assert data.shape[1] == self.patch_shape[0]
        return data.transpose(1, 2)


class PatchRecover(nn.Module):
    def __init__(self, output_shape, patch_shape, f_first=False):
        super().__init__()
        self.output_shape, self.patch_shape, self.f_first = output_shape, patch_shape, f_first
        if f_first or patch_shape[1] != 1:
            raise NotImplementedError("only frame patches (nf, 1) are on the pretraining path")

    def forward(self, data):
        return data.transpose(1, 2)


def _python_sample(nbatch, npatch, nmasked, nmic):
    idx = np.empty((nbatch, nmasked), dtype=np.int64)
    ch = np.empty((nbatch,), dtype=np.int64)
    for b in range(nbatch):
        idx[b] = random.sample(range(0, npatch), nmasked)
        ch[b] = random.randint(0, nmic - 1)
    return idx, ch


def _native_sample(nbatch, npatch, nmasked, nmic):
    """Same draws from Python's global generator, advanced natively (csrc/wavio.hip: sarssl_mask_sample)."""
    import ctypes
    from .. import _lib
    version, state, gauss = random.getstate()
    mt = np.array(state[:624], dtype=np.uint32)
    pos = ctypes.c_int(state[624])
    idx = np.empty((nbatch, nmasked), dtype=np.int64)
    ch = np.empty((nbatch,), dtype=np.int64)
    _lib.call("sarssl_mask_sample", mt.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), ctypes.c_int(nbatch), ctypes.c_int(npatch),
              ctypes.c_int(nmasked), ctypes.c_int(nmic), idx.ctypes.data_as(ctypes.c_void_p), ch.ctypes.data_as(ctypes.c_void_p))
    random.setstate((version, tuple(int(v) for v in mt) + (pos.value,), gauss))
    return idx, ch


_NATIVE_OK = None


def _native_sampler_ok():
    """One-time self check: the native sampler must reproduce ``random.sample`` / ``randint`` bit for bit on this interpreter
    (CPython's algorithm is an implementation detail); otherwise the Python loop is used."""
    global _NATIVE_OK
    if _NATIVE_OK is None:
        saved = random.getstate()
        try:
            random.seed(987654321)
            want = _python_sample(3, 256, 128, 2)
            after_py = random.getstate()
            random.seed(987654321)
            got = _native_sample(3, 256, 128, 2)
            _NATIVE_OK = bool(np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]) and random.getstate() == after_py)
        except Exception:
            _NATIVE_OK = False
        finally:
            random.setstate(saved)
    return _NATIVE_OK


class PatchMask(nn.Module):
    """Frame / channel masks.  ``sample`` draws the compact form the kernels use with the reference's exact host RNG
    call order (python ``random``: ``sample(range(npatch), nmasked)`` then ``randint(0, nmic-1)`` per batch item,
    utils_module.py:263-267, 305-308); ``forward`` builds the reference's dense (nbatch, npatch, dpatch, nmic)
    masks from it for API compatibility."""

    def __init__(self, patch_mode, nmasked_patch, npatch_shape, device):
        super().__init__()
        self.patch_mode, self.nmasked_patch, self.npatch_shape, self.device = patch_mode, nmasked_patch, npatch_shape, device
        if patch_mode != "T":
            raise NotImplementedError("only patch_mode 'T' is on the pretraining path")

    def gen_mask_idx(self, npatch_shape=[16, 16], nmasked_patch=10, cluster=1, patch_mode="T"):
        npatch = npatch_shape[0] * npatch_shape[1]
        if nmasked_patch > npatch:
            raise Exception("Number of masked patches is out of range")
        return torch.tensor(random.sample(range(0, npatch), nmasked_patch))

    def sample(self, nbatch, nmic=2):
        npatch = self.npatch_shape[0] * self.npatch_shape[1]
        if _native_sampler_ok() and 5 < self.nmasked_patch <= npatch <= 1045 and nmic == 2:
            return _native_sample(nbatch, npatch, self.nmasked_patch, nmic)
        return _python_sample(nbatch, npatch, self.nmasked_patch, nmic)

    def forward(self, data_shape):
        nbatch, npatch, dpatch, _, nmic = data_shape
        idx, ch = self.sample(nbatch, nmic)
        dev = self.device
        idx_t = torch.from_numpy(idx).to(dev)
        ch_t = torch.from_numpy(ch).to(dev).view(nbatch, 1)
        mp = torch.ones((nbatch, npatch), device=dev).scatter_(1, idx_t, 0.0)
        mc = torch.ones((nbatch, nmic), device=dev).scatter_(1, ch_t, 0.0)
        mask_patch_dense = mp.view(nbatch, npatch, 1, 1).expand(nbatch, npatch, dpatch, nmic).contiguous()
        mask_ch_dense = mc.view(nbatch, 1, 1, nmic).expand(nbatch, npatch, dpatch, nmic).contiguous()
        mask_dense = 1.0 - (1.0 - mask_patch_dense) * (1.0 - mask_ch_dense)
        return mask_dense, mask_patch_dense, mask_ch_dense, idx_t, ch_t
