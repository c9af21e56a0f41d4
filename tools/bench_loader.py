"""Host-side input pipeline throughput: segments/s delivered by the native PcmSegmentLoader vs a
torch DataLoader over FixMicSigDataset(raw_pcm=True) with N worker processes, on BASELINE-sized segments (2 ch x 65 792 samples).

    python tools/bench_loader.py [--n 1024] [--bs 64] [--threads 8] [--dir /tmp/sarssl_loader_bench]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa
import numpy as np
import torch
from sar_ssl_amd import dataset


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--dir", default="/tmp/sarssl_loader_bench")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    rng = np.random.default_rng(0)
    for i in range(a.n):
        p = os.path.join(a.dir, "%d.wav" % i)
        if not os.path.exists(p):
            dataset.write_wav_pcm16(p, rng.integers(-20000, 20000, size=(65792, 2), dtype=np.int16))
    files = dataset.segment_files(a.dir)[: a.n]
    dev = "cuda:0" if torch.cuda.is_available() else None
    for name, mk in (("native PcmSegmentLoader (%d threads)" % a.threads,
                      lambda: dataset.PcmSegmentLoader(files, a.bs, 65792, 2, nthreads=a.threads, device=dev)),
                     ("torch DataLoader (%d workers)" % a.threads,
                      lambda: torch.utils.data.DataLoader(dataset.FixMicSigDataset(a.dir, 16000, False, a.n, raw_pcm=True),
                                                          batch_size=a.bs, num_workers=a.threads, pin_memory=dev is not None))):
        for rep in range(2):                                                        # second pass = warm page cache
            t0 = time.perf_counter()
            nseg = 0
            for batch in mk():
                x = batch[0]
                if dev is not None and not x.is_cuda:
                    x = x.to(dev, non_blocking=True)
                nseg += x.shape[0]
            if dev is not None:
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("%-42s %8.0f segments/s  (%.2f GB/s)" % (name, nseg / dt, nseg * 65792 * 4 / dt / 1e9), flush=True)


if __name__ == "__main__":
    main()
