"""``FixMicSigDataset`` (code/dataset.py:107-178) for pre-generated simulated microphone signals: ``{idx}.wav`` files
(16 kHz, nch channels, 16-bit PCM as written by code/data_generation/utils_simu_rir_sig.py:855-856), minus ``*_dp.wav``.

``soundfile`` is not available here, so RIFF/WAVE PCM-16 is parsed directly with numpy.  ``raw_pcm=True`` returns the
int16 samples untouched (half the PCIe bytes); the STFT kernel converts on the fly.
"""
import struct
from pathlib import Path

import numpy as np
from torch.utils.data import Dataset


def read_wav_pcm16(path):
    """-> (int16 array (nsample, nch), fs).  Supports plain PCM-16 and WAVE_FORMAT_EXTENSIBLE PCM-16."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("%s: not a RIFF/WAVE file" % path)
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, nch, fs, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, nch, fs, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("%s: missing fmt/data chunk" % path)
    tag, nch, fs, bits = fmt
    if tag != 1 or bits != 16:
        raise ValueError("%s: only 16-bit PCM is supported (format tag %d, %d bits)" % (path, tag, bits))
    x = np.frombuffer(pcm, dtype="<i2")
    return x[: (x.size // nch) * nch].reshape(-1, nch), fs


def write_wav_pcm16(path, pcm, fs=16000):
    """pcm: int16 (nsample, nch)."""
    pcm = np.ascontiguousarray(pcm.astype("<i2"))
    nch = pcm.shape[1]
    body = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, nch, fs, fs * nch * 2, nch * 2, 16)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(body)) + body)


class Selecting(object):
    """Keep samples [select_range[0], select_range[1]) of a (nsample, nch) signal (code/dataset.py:386-395)."""

    def __init__(self, select_range):
        self.select_range = select_range

    def __call__(self, mic_sig):
        nsample = mic_sig.shape[0]
        assert self.select_range[-1] <= nsample, f"Selecting range ({self.select_range[-1]}) is larger than signal length ({nsample})~"
        return mic_sig[self.select_range[0]:self.select_range[1], ...]


class FixMicSigDataset(Dataset):
    def __init__(self, data_dir, fs, load_anno, dataset_sz, load_dp=False, transforms=None, raw_pcm=False):
        dirs = data_dir if isinstance(data_dir, list) else [data_dir]
        files, dp_files = [], []
        for d in dirs:
            files += list(Path(d).rglob("*.wav"))
            dp_files += list(Path(d).rglob("*_dp.wav"))
        if isinstance(data_dir, list):
            np.random.shuffle(files)
        dp = set(dp_files)
        self.files = [f for f in files if f not in dp]
        self.dataset_sz = len(self.files) if dataset_sz is None else int(np.min([len(self.files), dataset_sz]))
        self.fs, self.load_anno, self.load_dp, self.transforms, self.raw_pcm = fs, load_anno, load_dp, transforms, raw_pcm

    def __len__(self):
        return self.dataset_sz

    def __getitem__(self, idx):
        file_name = str(self.files[idx])
        pcm, fs = read_wav_pcm16(file_name)
        if self.fs != fs:
            import scipy.signal
            mic_sig = scipy.signal.resample_poly(pcm.astype(np.float64) / 32768.0, self.fs, fs)
        elif self.raw_pcm and self.transforms is None:
            return [np.ascontiguousarray(pcm)]
        else:
            mic_sig = pcm.astype(np.float32) / 32768.0            # soundfile's float conversion of PCM-16
        if self.transforms is not None:
            for t in self.transforms:
                mic_sig = t(mic_sig)
        return_data = [mic_sig.astype(np.float32)]
        if self.load_anno:
            info = dict(np.load(file_name.replace(".wav", "_info.npz")))
            vol = info["room_sz"][0] * info["room_sz"][1] * info["room_sz"][2]
            sur = info["room_sz"][0] * info["room_sz"][1] + info["room_sz"][0] * info["room_sz"][2] + info["room_sz"][1] * info["room_sz"][2]
            return_data += [{"TDOA": info["TDOA"].astype(np.float32), "T60": info["T60_edc"].astype(np.float32),
                             "DRR": info["DRR"].astype(np.float32), "C50": info["C50"].astype(np.float32),
                             "ABS": np.array(0.161 * vol / sur / info["T60_edc"]).astype(np.float32)}]
        if self.load_dp:
            dp, fs_dp = read_wav_pcm16(file_name.replace(".wav", "_dp.wav"))
            dp_sig = dp.astype(np.float32) / 32768.0
            if self.fs != fs_dp:
                import scipy.signal
                dp_sig = scipy.signal.resample_poly(dp_sig, self.fs, fs_dp)
            if self.transforms is not None:
                for t in self.transforms:
                    dp_sig = t(dp_sig)
            return_data += [dp_sig]
        return return_data
