"""Operand-rounding study (TEST INFRASTRUCTURE, CPU only): which MFMA operand type reaches north_star's per-bin 1e-3?

    python oracle/operand_rounding_study.py [--modes bf16,fp16,...] [--train]

Runs the oracle's full-size pretraining forward on fixture F3's inputs (recipe weights seed 0, recipe signal seed 3, the fixture's
masks) with the operands of every dense contraction rounded the way a matrix-core path would round them - the contraction itself
stays f32 - and prints the deviation of the loss and of F3's 2 048 sampled `pred` bins (of the output range, the quantity
`tests/test_gpu_model.py::test_fullsize_forward_backward` gates).  Nothing is measured on a GPU: this answers, for free, what the
HIP kernels' operand type has to be.

Modes (applied to both operands unless noted):
  f32        no rounding (sanity: must reproduce F3 to ~1e-6)
  bf16       operands rounded to bf16                       (the timed mode of rounds 1-3)
  fp16       operands rounded to IEEE half                  (v_mfma_f32_32x32x16_f16, same rate as bf16)
  bf16x2a    activations hi+lo bf16 (two passes), weights one bf16
  bf16x2w    weights hi+lo, activations one bf16
  bf16x3     hi*hi + hi*lo + lo*hi  (the repo's `fp32` mode)
  fp16x2a / fp16x2w   fp16 with the activations / the weights split hi+lo (two passes)
  fp16x3     fp16 hi*hi + hi*lo + lo*hi
  fp16a      activation operand rounded to fp16, weights exact
`--store` additionally rounds every contraction OUTPUT to the mode's storage type (what bf16 / fp16 activation storage adds).
`--family NAME=MODE,...` overrides the mode per family: stem1 (4->64 1x1), conv3 (3x3), stem4 (64->4), patch, lin (Conformer
Linear / pointwise; or one of its layers: ffn1, ffn2, q, k, v, pos, out, pw1, pw2), attn (score / PV products), dec (decoder; dec1, dec2).  `NAME=MODE:n` keeps that family's outputs in f32 under --store.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import recipes                      # noqa: E402
import sarssl_oracle as orc         # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _r(x, kind):
    if kind == "bf16":
        return x.bfloat16().float()
    if kind == "fp16":
        return x.half().float()
    return x


def _split(x):
    hi = x.bfloat16().float()
    return hi, (x - hi).bfloat16().float()


def _split16(x):
    hi = x.half().float()
    return hi, (x - hi).half().float()


class Policy:
    def __init__(self, default, store, overrides):
        self.default, self.store, self.over = default, store, overrides

    GROUP = {"ffn1": "lin", "ffn2": "lin", "q": "lin", "k": "lin", "v": "lin", "pos": "lin", "out": "lin", "pw1": "lin", "pw2": "lin",
             "dec1": "dec", "dec2": "dec"}

    def _spec(self, fam):
        """a layer name (ffn1, ffn2, q, k, v, pos, out, pw1, pw2, dec1, dec2) falls back to its group (lin / dec), then to the default"""
        if fam in self.over:
            return self.over[fam]
        return self.over.get(self.GROUP.get(fam, fam), self.default)

    def mode(self, fam):
        return self._spec(fam).split(":")[0]

    def stores(self, fam):
        """`fam=mode:n` keeps that family's outputs in f32 under --store (the hybrid mode: f32 activation storage behind the stem);
        `fam=mode:s` stores them in the mode's 16-bit type even without --store."""
        o = self._spec(fam)
        return o.endswith(":s") or (self.store and not o.endswith(":n"))

    def contract(self, fam, fn, a, w):
        """fn(a, w) is the f32 contraction; a = activation-side operand, w = weight-side operand."""
        m = self.mode(fam)
        if m == "f32":
            y = fn(a, w)
        elif m in ("bf16", "fp16"):
            y = fn(_r(a, m), _r(w, m))
        elif m == "bf16x2a":
            ah, al = _split(a)
            wb = _r(w, "bf16")
            y = fn(ah, wb) + fn(al, wb)
        elif m == "bf16x2w":
            wh, wl = _split(w)
            ab = _r(a, "bf16")
            y = fn(ab, wh) + fn(ab, wl)
        elif m == "fp16a":                   # activation operand rounded to fp16, weights exact (a VALU layer on an fp16-stored input)
            y = fn(_r(a, "fp16"), w)
        elif m == "fp16x2a":                 # activations hi+lo fp16 (two passes), weights one fp16
            ah, al = _split16(a)
            wb = _r(w, "fp16")
            y = fn(ah, wb) + fn(al, wb)
        elif m == "fp16x2w":
            wh, wl = _split16(w)
            ab = _r(a, "fp16")
            y = fn(ab, wh) + fn(ab, wl)
        elif m == "fp16x3":
            ah, al = _split16(a)
            wh, wl = _split16(w)
            y = fn(ah, wh) + fn(ah, wl) + fn(al, wh)
        elif m == "bf16x3":
            ah, al = _split(a)
            wh, wl = _split(w)
            y = fn(ah, wh) + fn(ah, wl) + fn(al, wh)
        else:
            raise ValueError(m)
        if self.stores(fam) and m != "f32":
            y = _r(y, "fp16" if m.startswith("fp16") else "bf16")
        return y


_attn_calls = [0]


def family_of_linear(x, w):
    out_f, in_f = w.shape
    if (out_f, in_f) == (3072, 768):
        return "dec1"
    if (out_f, in_f) == (1024, 3072):
        return "dec2"
    if out_f == 4 * in_f:
        return "ffn1"
    if in_f == 4 * out_f:
        return "ffn2"
    if out_f == 2 * in_f:
        return "pw1"
    if out_f == in_f:                          # oracle mhsa(): query, key, value, pos, out - in that order
        _attn_calls[0] += 1
        return ("q", "k", "v", "pos", "out")[(_attn_calls[0] - 1) % 5]
    return "lin"


def family_of_conv(w):
    co, ci, kh, kw = w.shape
    if (kh, kw) == (3, 3):
        return "conv3"
    if (kh, kw) == (1, 1):
        return "stem1" if ci == 4 else "stem4"
    return "patch"


class FProxy:
    """Stands in for `torch.nn.functional` inside the oracle module."""
    def __init__(self, pol):
        self.pol = pol

    def __getattr__(self, name):
        return getattr(F, name)

    def linear(self, x, w, b=None):
        y = self.pol.contract(family_of_linear(x, w), lambda a, ww: F.linear(a, ww), x, w)
        return y if b is None else y + b

    def conv2d(self, x, w, b=None, **kw):
        return self.pol.contract(family_of_conv(w), lambda a, ww: F.conv2d(a, ww, None, **kw), x, w)

    def conv1d(self, x, w, b=None, **kw):
        if kw.get("groups", 1) != 1:                                     # depthwise: VALU kernel, f32 arithmetic
            return F.conv1d(x, w, b, **kw)
        y = self.pol.contract("pw2", lambda a, ww: F.conv1d(a, ww, None, **kw), x, w)
        return y if b is None else y + b[None, :, None]


class TorchProxy:
    def __init__(self, pol):
        self.pol = pol

    def __getattr__(self, name):
        return getattr(torch, name)

    def einsum(self, eq, a, b):
        return self.pol.contract("attn", lambda x, y: torch.einsum(eq, x, y), a, b)


def run(pol, train, z, sd, x):
    orc.F, orc.torch = FProxy(pol), TorchProxy(pol)
    _attn_calls[0] = 0
    try:
        sd = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            loss, diff, aux = orc.sarssl_pretrain_forward(x, sd, torch.from_numpy(z["mask_idx"]), torch.from_numpy(z["mask_ch"]),
                                                          train=train, p_drop=0.0)
    finally:
        orc.F, orc.torch = F, torch
    mode = "train" if train else "eval"
    got = aux["pred"].reshape(-1)[torch.from_numpy(z[mode + ".pred_idx"])]
    want = torch.from_numpy(z[mode + ".pred_vals"])
    err = (got - want).abs()
    return dict(loss_rel=abs(float(loss) / float(z[mode + ".loss"]) - 1.0),
                pred_max_of_range=float(err.max() / float(z[mode + ".pred_absmax"])),
                pred_rms_of_range=float((err ** 2).mean().sqrt() / float(z[mode + ".pred_absmax"])),
                pred_max_rel_bin=float((err / want.abs().clamp_min(0.05 * float(z[mode + ".pred_absmax"]))).max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="f32,bf16,fp16,bf16x2a,bf16x2w,bf16x3")
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--store", action="store_true")
    ap.add_argument("--family", default="")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    z = np.load(os.path.join(GOLD, "f3_fullsize.npz"), allow_pickle=False)
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))["pretrain"]
    sd = recipes.recipe_state_dict(man, 0)
    x = orc.data_preprocess(recipes.recipe_signal(2, 65792, 2, seed=3))
    over = dict(kv.split("=") for kv in a.family.split(",") if kv)
    rows = {}
    for m in a.modes.split(","):
        for store in ((False, True) if a.store else (False,)):
            r = run(Policy(m, store, over), a.train, z, sd, x)
            tag = m + ("+store" if store else "") + ("" if not over else " " + a.family)
            rows[tag] = r
            print("%-28s loss %.2e  pred max %.2e  rms %.2e of range   (train=%s)" %
                  (tag, r["loss_rel"], r["pred_max_of_range"], r["pred_rms_of_range"], a.train), flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
