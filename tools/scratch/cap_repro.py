import os, sys, torch
mode = sys.argv[1]
n = int(sys.argv[2])
dev = torch.device("cuda:0")
a = torch.randn(1024, 1024, device=dev)
outs = [torch.zeros(1024, 1024, device=dev) for _ in range(n)]
cap = torch.cuda.Stream()
ws = torch.cuda.Stream()
side = torch.cuda.Stream()
ws2 = torch.cuda.Stream()
keep = []
def wait(w, o):
    ev = torch.cuda.Event(); ev.record(o); w.wait_event(ev); keep.append(ev)
def body():
    cur = torch.cuda.current_stream()
    x = a
    if "side" in mode:
        wait(side, cur)
        if "prefork" in mode:
            wait(ws2, cur)                      # ws2 enters the capture through the origin stream first
    for i in range(n):
        x = x * 1.0001
        wait(ws, cur)
        with torch.cuda.stream(ws):
            if "alloc" in mode:
                t = torch.empty_like(x); torch.mul(x, 2.0, out=t); outs[i].add_(t)
            else:
                outs[i].add_(x)
        if "side" in mode:
            with torch.cuda.stream(side):
                y = x + 1
                wait(ws2, side)
                with torch.cuda.stream(ws2):
                    outs[i].mul_(1.0) if False else y.mul_(2.0)
        if "midjoin" in mode and i % 4 == 3:
            wait(cur, ws)
    wait(cur, ws)
    if "side" in mode:
        with torch.cuda.stream(side):
            wait(side, ws2)
        wait(cur, side)
        if "dirjoin" in mode:
            wait(cur, ws2)
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
print("OK", mode, n, float(outs[0].sum()))
