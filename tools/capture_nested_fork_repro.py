"""Minimal reproduction of a ROCm 7.2 stream-capture crash (no sar_ssl_amd code involved): a stream that joins a capture by waiting
on an event recorded on another NON-origin stream ("a fork off a fork") makes hipStreamEndCapture segfault; any number of streams
forked directly off the origin stream is fine.

    python tools/capture_nested_fork_repro.py ws 3          # origin + one fork               -> OK
    python tools/capture_nested_fork_repro.py ws+side 3     # origin + two forks              -> OK
    python tools/capture_nested_fork_repro.py side+ws2 3    # origin -> side -> ws2 (nested)  -> segfault in capture_end

Consequence for this package (engine.wgrad_no_fork): inside a captured step only the origin stream may have a companion stream."""
import os, sys, torch
flags = sys.argv[1].split("+")
n = int(sys.argv[2])
dev = torch.device("cuda:0")
a = torch.randn(1024, 1024, device=dev)
o1 = [torch.zeros(1024, 1024, device=dev) for _ in range(n)]
o2 = [torch.zeros(1024, 1024, device=dev) for _ in range(n)]
o3 = [torch.zeros(1024, 1024, device=dev) for _ in range(n)]
cap, ws, side, ws2 = (torch.cuda.Stream() for _ in range(4))
keep = []
def wait(w, o):
    ev = torch.cuda.Event(); ev.record(o); w.wait_event(ev); keep.append(ev)
def body():
    cur = torch.cuda.current_stream()
    x = a * 1.0
    if "side" in flags:
        wait(side, cur)
        with torch.cuda.stream(side):
            y = a + 1.0
    for i in range(n):
        x = x * 1.0001
        if "ws" in flags:
            wait(ws, cur)
            with torch.cuda.stream(ws):
                o1[i].add_(x)
        if "side" in flags:
            with torch.cuda.stream(side):
                y = y * 1.0001
                o2[i].add_(y)
                if "ws2" in flags:
                    wait(ws2, side)
                    with torch.cuda.stream(ws2):
                        o3[i].add_(y)
    if "ws" in flags:
        wait(cur, ws)
    if "side" in flags:
        with torch.cuda.stream(side):
            if "ws2" in flags:
                wait(side, ws2)
        wait(cur, side)
    return x
with torch.cuda.stream(cap):
    body(); cap.synchronize()
    g = torch.cuda.CUDAGraph()
    g.capture_begin(capture_error_mode="thread_local")
    body()
    g.capture_end()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
print("OK", flags, n, float(o1[0].sum()), float(o2[0].sum()), float(o3[0].sum()))
