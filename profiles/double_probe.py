"""tools/cfgs/double.yaml (NUM_POINTS 32768) through the product runner: scenes/s of a closed loop of K batches of B scenes, and the
sampling kernel's time at that size.  usage: python profiles/double_probe.py [K] [B]"""
import importlib, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "3d_adapt_auto_driving_amd"
C, E, S = (importlib.import_module(PKG + "." + m) for m in ("config", "eval_rcnn", "synth"))
pu = importlib.import_module(PKG + ".pointnet2.pointnet2_utils")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda:0"
cfg = C.default_eval_cfg()
C.merge_into({"RPN": {"NUM_POINTS": 32768}}, cfg)
model = E.build_model(cfg, dev, seed=0)
batches = [torch.from_numpy(S.scenes(B, 32768, seed0=1000 + 8 * s)).to(dev) for s in range(16)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
pu.furthest_point_sample(batches[0], 4096)
ev[0].record(); pu.furthest_point_sample(batches[0], 4096); ev[1].record(); torch.cuda.synchronize()
print("furthest_point_sample %d x (32768 -> 4096): %.2f ms" % (B, ev[0].elapsed_time(ev[1])))
runner = E.make_runner(model, cfg, dev)
print("runner:", type(runner).__name__)
def loop(k):
    n = 0
    for i in range(k):
        nxt = [batches[(i + d) % 16] for d in range(1, runner.depth + 1) if i + d < k]
        d = runner.submit(batches[i % 16], nxt)
        n += d is not None
    n += len(runner.drain())
    torch.cuda.synchronize()
    return n
loop(8)
for _ in range(2):
    t0 = time.perf_counter(); n = loop(K); dt = time.perf_counter() - t0
    print("closed loop of %d batches x %d scenes: %.1f scenes/s (%.2f ms per batch; %d batches back)" % (K, B, K * B / dt, dt / K * 1e3, n))
