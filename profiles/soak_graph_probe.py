"""Soak of the runner the product uses (hipGraph replay unless PRCNN_GRAPHS=0): N steps over a ring of 14 resident batches; every time a batch comes
round again its detections must be bit-identical to the first time (same kernels, same inputs), and the step time must not drift.
usage: python profiles/soak_graph_probe.py [steps]"""
import importlib, os, sys, time, collections, hashlib
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
NS = 14
batches = [torch.from_numpy((S.lidar_scenes if s % 2 else S.scenes)(8, 16384, seed0=s * 8)).to(dev) for s in range(NS)]
runner = E.make_runner(model, cfg, dev); print("runner:", type(runner).__name__)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
M = cfg.TEST.RPN_POST_NMS_TOP_N
host = [(torch.empty((8, M, 7), pin_memory=True), torch.empty((8, M), pin_memory=True), torch.empty((8,), dtype=torch.int32, pin_memory=True)) for _ in range(8)]
first, bad, pend, t_block = {}, 0, collections.deque(), []
def check(i, ev, slot):
    global bad
    ev.synchronize()
    h = hashlib.sha1(b"".join(t.numpy().tobytes() for t in host[slot])).hexdigest()
    if first.setdefault(i % NS, h) != h:
        bad += 1
        print("step %d (batch %d): detections differ from the first pass" % (i, i % NS))
came = 0
t0 = time.perf_counter()
for i in range(N):
    det = runner.submit(batches[i % NS], [batches[(i + d) % NS] for d in range(1, runner.depth + 1) if i + d < N])
    if det is not None:                                  # an EARLIER batch's detections, in submit order (paired members: up to 3 submits late)
        j = came; came += 1
        slot = j % 8
        with torch.cuda.stream(det["stream"]):
            for h, k in zip(host[slot], ("boxes", "scores", "num")):
                h.copy_(det[k], non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
        pend.append((j, ev, slot))
        if len(pend) > 3:
            check(*pend.popleft())
    if (i + 1) % 200 == 0:
        torch.cuda.synchronize(); t1 = time.perf_counter(); t_block.append((t1 - t0) / 200 * 1e3); t0 = t1
runner.drain()
while pend:
    check(*pend.popleft())
print("steps %d, repeats that differ: %d, ms per step in blocks of 200: %s" % (N, bad, ["%.3f" % t for t in t_block]))
assert bad == 0
