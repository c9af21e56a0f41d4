"""Does a replayed HIP graph run forked branches concurrently?  Two chains of sleep kernels (100-300 nodes each) forked off one
root node, captured in different host orders; replay time in units of one sleep.  ROCm 7.2 on MI355X: yes, in every capture order
(206 node times for two 200-node branches) - the little overlap between the two encoder branches of the captured training step
(NOTES.md 4.7) is a matter of resources (LDS / registers of the persistent convolution kernels), not of the graph runtime."""
import sys, time, torch
dev = torch.device("cuda:0")
def run(mode, NA, NB, CY, root=1):
    cap, side = torch.cuda.Stream(), torch.cuda.Stream()
    def body():
        cur = torch.cuda.current_stream()
        for _ in range(root): torch.cuda._sleep(CY)
        side.wait_stream(cur)
        if mode == "side_first":
            with torch.cuda.stream(side):
                for _ in range(NB): torch.cuda._sleep(CY)
            for _ in range(NA): torch.cuda._sleep(CY)
        elif mode == "main_first":
            for _ in range(NA): torch.cuda._sleep(CY)
            with torch.cuda.stream(side):
                for _ in range(NB): torch.cuda._sleep(CY)
        else:
            for i in range(max(NA, NB) // 10):
                with torch.cuda.stream(side):
                    for _ in range(10 if i * 10 < NB else 0): torch.cuda._sleep(CY)
                for _ in range(10 if i * 10 < NA else 0): torch.cuda._sleep(CY)
        cur.wait_stream(side)
        torch.cuda._sleep(CY)
    with torch.cuda.stream(cap):
        body(); cap.synchronize()
        g = torch.cuda.CUDAGraph()
        g.capture_begin(capture_error_mode="thread_local"); body(); g.capture_end()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3
for CY in (100000, 20000):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        torch.cuda._sleep(CY); s.synchronize()
        g.capture_begin(capture_error_mode="thread_local")
        for _ in range(100): torch.cuda._sleep(CY)
        g.capture_end()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    unit = (time.perf_counter() - t0) * 1e3 / 100
    print("CY %d: one sleep in a linear graph = %.4f ms" % (CY, unit))
    for mode in ("side_first", "main_first", "chunks10"):
        for NA, NB in ((200, 200), (300, 100), (100, 300)):
            t = run(mode, NA, NB, CY)
            print("  %-11s main %3d side %3d : replay %.2f ms = %.0f sleeps (ideal %d, serial %d)" % (mode, NA, NB, t, t / unit, 1 + max(NA, NB) + 1, 1 + NA + NB + 1))
