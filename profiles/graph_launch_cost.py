"""Device-side cost of a hipGraph launch on this runtime: a chain of n tiny kernels eagerly vs as one graph vs as n one-kernel graphs."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
dev = torch.device("cuda", 0)
x = torch.zeros(256, device=dev)
s = torch.cuda.Stream(dev)
def timed(fn, reps):
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(s)
        for _ in range(reps): fn()
        t1 = time.perf_counter(); e1.record(s); s.synchronize()
    return (t1 - t0) / reps * 1e6, e0.elapsed_time(e1) / reps * 1e3
def capture(n):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        s.synchronize(); g.capture_begin(pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local")
        for _ in range(n): x.add_(1.0)
        g.capture_end()
    return g
for n in (1, 10, 40):
    def eager():
        for _ in range(n): x.add_(1.0)
    g = capture(n)
    g1 = capture(1)
    def many():
        for _ in range(n): g1.replay()
    he, de = timed(eager, 200); hg, dg = timed(g.replay, 200); hm, dm = timed(many, 200)
    print("n = %2d tiny kernels: eager host %.1f us, device %.1f us | one graph: host %.1f, device %.1f | n one-kernel graphs: host %.1f, device %.1f"
          % (n, he, de, hg, dg, hm, dm))
