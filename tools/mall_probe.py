"""Does the 256 MB Infinity Cache speed up producer -> consumer chains of HBM-bound passes?  A 3-pass elementwise chain over 537 MB
tensors, whole or cut into chunks that fit the cache.  MI355X: +10 % at 134 MB chunks, lost again to launch overhead below (NOTES.md 4.6)."""
import torch, time
dev = torch.device("cuda:0")
N = 64 * 256 * 256 * 64          # bf16 elements: 537 MB
a = torch.randn(N, device=dev, dtype=torch.bfloat16)
b = torch.empty_like(a); c = torch.empty_like(a)
def run(nchunk, reps=20):
    cs = N // nchunk
    def once():
        for i in range(nchunk):
            s = slice(i * cs, (i + 1) * cs)
            torch.mul(a[s], 2.0, out=b[s])
            torch.add(b[s], 1.0, out=c[s])
            torch.add(b[s], c[s], out=c[s])
    for _ in range(3): once()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (1, 2, 4, 8, 16, 32, 64):
    ms = run(n)
    # bytes: mul r+w, add r+w, add 2r+w = 7 * 1.07GB/2 ...
    gb = 7 * N * 2 / 1e9
    print(f"chunks {n:3d} ({N*2/n/1e6:6.1f} MB per tensor chunk): {ms:.3f} ms  -> {gb/ms:.2f} TB/s effective")
