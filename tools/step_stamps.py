#!/usr/bin/env python
"""Timeline of an UNPROFILED graph replay: one-thread marker launches (sarssl_stamp: the constant-rate device clock) captured between
the phases of the two encoder streams, read back after the replays.  rocprofv3 delays the second hardware queue's packets (the host
needs ~7 ms to submit a profiled replay), so its traces under-state how much the two streams overlap; this shows what the replay does
when nobody is watching.    python tools/step_stamps.py [--steps 30]"""
import argparse
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarssl_boot  # noqa: E402,F401
from sar_ssl_amd import hip, model, runtime, synth, _lib  # noqa: E402
from sar_ssl_amd.graph import PretrainStepGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--precision", default="fp16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    runtime.set_precision(args.precision)
    torch.manual_seed(1)
    random.seed(1)
    net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device=dev).to(dev).train()
    flat = runtime.FlatParams(net)
    pcm = torch.from_numpy(synth.to_pcm16(synth.make_batch(0, 16))).repeat(4, 1, 1).to(dev)
    buf = torch.zeros(64, dtype=torch.int64, device=dev)
    labels = []
    hip._stamps = (buf, labels)
    g = PretrainStepGraph(net, flat, None, lr=1e-3)
    g.step(pcm=pcm, static=True)          # warm-up pass + capture (markers recorded twice: keep the captured set = the last len/2)
    hip._stamps = None
    n = len(labels) // 2
    labels = labels[n:]
    khz = _lib.lib().sarssl_wall_clock_khz()
    rows = []
    for _ in range(args.steps):
        g.step(pcm=pcm, static=True)
        torch.cuda.synchronize()
        rows.append(buf[n:2 * n].cpu().numpy().astype(np.float64))
    t = np.array(rows[5:])
    t = (t - t[:, :1]) / khz * 1e3          # us since the first marker of the replay
    med = np.median(t, axis=0)
    order = np.argsort(med)
    print("markers of %d unprofiled replays (us since %s, median; min .. max)" % (len(t), labels[0]))
    for i in order:
        print("%9.1f   %-28s  (%8.1f .. %8.1f)" % (med[i], labels[i], t[:, i].min(), t[:, i].max()))


if __name__ == "__main__":
    main()
