"""Yardstick for the HBM-bound passes: what torch's own streaming kernels reach on 537 MB tensors on this part (pure read, pure
write, read + write mixes).  MI355X: pure write 6.5 TB/s, mixes 5.8-6.1 TB/s, a single sequential read stream 3.4-3.9 TB/s (NOTES.md 4.6)."""
import torch
a = torch.randn(268435456 // 2, device="cuda").bfloat16().repeat(2)
b = torch.empty_like(a)
f = a.float()[:a.numel() // 2].contiguous()
def t(name, fn, nbytes):
    fn(); torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) / 20 * 1e3
    print("%-34s %8.1f us  %5.2f TB/s" % (name, us, nbytes / us / 1e6))
t("sum bf16 537MB (pure read)", lambda: a.sum(), a.numel() * 2)
t("amax bf16 537MB (pure read)", lambda: a.amax(), a.numel() * 2)
t("sum f32 537MB (pure read)", lambda: f.sum(), f.numel() * 4)
t("zero_ 537MB (pure write)", lambda: b.zero_(), a.numel() * 2)
t("mul 537MB (R+W)", lambda: torch.mul(a, 2.0, out=b), a.numel() * 4)
t("add 2R+W", lambda: torch.add(a, b, out=b), a.numel() * 6)
