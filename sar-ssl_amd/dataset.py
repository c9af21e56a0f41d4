"""``FixMicSigDataset`` (code/dataset.py:107-178) for pre-generated simulated microphone signals: ``{idx}.wav`` files
(16 kHz, nch channels, 16-bit PCM as written by code/data_generation/utils_simu_rir_sig.py:855-856), minus ``*_dp.wav``.

``soundfile`` is not available here, so RIFF/WAVE PCM-16 is parsed directly with numpy.  ``raw_pcm=True`` returns the
int16 samples untouched (half the PCIe bytes); the STFT kernel converts on the fly.
"""
import ctypes
import queue
import struct
import threading
from pathlib import Path

import numpy as np
import torch
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


def read_wav_batch(paths, nsample, nch, fs=0, offset=0, out=None, nthreads=0):
    """Raw int16 PCM of samples [offset, offset+nsample) of every file -> int16 tensor (len(paths), nsample, nch), filled by the
    threaded reader of the C-ABI library (csrc/wavio.hip, ``sarssl_wav_read_batch``; the GIL is released during the call)."""
    from . import _lib
    n = len(paths)
    if out is None:
        out = torch.empty((n, nsample, nch), dtype=torch.int16)
    assert out.dtype == torch.int16 and out.is_contiguous() and not out.is_cuda and out.numel() >= n * nsample * nch
    arr = (ctypes.c_char_p * n)(*[str(p).encode() for p in paths])
    _lib.call("sarssl_wav_read_batch", arr, ctypes.c_int(n), ctypes.c_long(nsample), ctypes.c_int(nch), ctypes.c_int(fs),
              ctypes.c_long(offset), ctypes.c_void_p(out.data_ptr()), ctypes.c_int(nthreads))
    return out[:n] if out.shape[0] != n else out


def wav_probe(path):
    """-> (nch, fs, nsample) of a PCM-16 WAV file."""
    from . import _lib
    nch, fs, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_long()
    _lib.call("sarssl_wav_probe", str(path).encode(), ctypes.byref(nch), ctypes.byref(fs), ctypes.byref(ns))
    return nch.value, fs.value, ns.value


class PcmSegmentLoader:
    """Batch iterator over a FixMicSigDataset-style directory of equal-length PCM-16 segments that replaces
    ``DataLoader(FixMicSigDataset(...), num_workers=N)`` on the pretraining path (code/run_pretrain.py:160-186): a background
    thread fills (pinned) int16 batch buffers with the native reader and, when ``device`` is a GPU, uploads them on a copy stream
    while the previous step computes.  Yields ``[pcm]`` with pcm int16 (B, nsample, nch) - the STFT kernel converts on the fly.
    Without a device the yielded host tensor is a view of a recycled buffer: valid until the next batch is requested.

    shuffle / rank / world / seed reproduce ``DistributedSampler`` semantics (rank-strided slice of one global permutation per
    epoch, padded by wrap-around); ``set_epoch`` changes the permutation."""

    def __init__(self, files, batch_size, nsample=None, nch=None, fs=16000, shuffle=False, seed=0, rank=0, world=1, drop_last=False,
                 device=None, nthreads=8, prefetch=2):
        self.files = [str(f) for f in files]
        assert self.files, "no segments"
        if nsample is None or nch is None:
            c, _, ns = wav_probe(self.files[0])
            nch, nsample = nch or c, nsample or ns
        self.batch_size, self.nsample, self.nch, self.fs = int(batch_size), int(nsample), int(nch), fs
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, seed, rank, world, drop_last
        self.device = torch.device(device) if device is not None else None
        self.nthreads, self.prefetch, self.epoch = nthreads, max(1, prefetch), 0
        self.per_rank = (len(self.files) + world - 1) // world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.per_rank // self.batch_size if self.drop_last else (self.per_rank + self.batch_size - 1) // self.batch_size

    def _order_all(self, epoch):
        n = len(self.files)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        total = self.per_rank * self.world
        return idx + idx[: total - n]

    def _order_for(self, epoch):
        return self._order_all(epoch)[self.rank::self.world]

    def _order(self):
        return self._order_for(self.epoch)

    def __iter__(self):
        order = self._order()
        nb = len(self)
        cuda = self.device is not None and self.device.type == "cuda"
        pin = cuda and torch.cuda.is_available()
        bufs = [torch.empty((self.batch_size, self.nsample, self.nch), dtype=torch.int16, pin_memory=pin) for _ in range(self.prefetch + 1)]
        free, ready = queue.Queue(), queue.Queue(maxsize=self.prefetch)
        for b in bufs:
            free.put(b)
        copy_stream = torch.cuda.Stream(self.device) if cuda else None
        stop = threading.Event()

        def producer():
            try:
                for i in range(nb):
                    if stop.is_set():
                        return
                    ids = order[i * self.batch_size:(i + 1) * self.batch_size]
                    buf = free.get()
                    host = read_wav_batch([self.files[j] for j in ids], self.nsample, self.nch, self.fs, 0, out=buf, nthreads=self.nthreads)
                    if cuda:
                        with torch.cuda.stream(copy_stream):
                            dev = host.to(self.device, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                        ready.put((dev, ev, buf))
                    else:
                        ready.put((host, None, buf))                            # valid until the consumer asks for the next batch
                ready.put(None)
            except BaseException as e:                                              # surface reader errors in the consumer
                ready.put(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                t, ev, buf = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    ev.synchronize()                                                 # the pinned buffer is reusable once the copy is done
                    t.record_stream(torch.cuda.current_stream(self.device))
                    free.put(buf)
                    yield [t]
                else:
                    yield [t]
                    free.put(buf)
        finally:
            stop.set()
            while th.is_alive():
                try:
                    ready.get(timeout=0.05)
                except queue.Empty:
                    pass
                free.put(bufs[0])
            th.join()


def segment_files(data_dir):
    """The ``*.wav`` files FixMicSigDataset would list (code/dataset.py:113-126): recursive, minus ``*_dp.wav``."""
    dirs = data_dir if isinstance(data_dir, list) else [data_dir]
    files, dp = [], set()
    for d in dirs:
        files += list(Path(d).rglob("*.wav"))
        dp |= set(Path(d).rglob("*_dp.wav"))
    return [f for f in files if f not in dp]
