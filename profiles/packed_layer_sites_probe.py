"""Which of the per-point layers (prcnn_packed_layer: the FP modules' coarse products G and second layers, RPN SA2's per-point parts, the RCNN
heads) cost the step how much?  The product runner with the calls of ONE shape (rows, K, N) launched twice, captured into the same graphs,
against the unchanged runner -- profiles/sensitivity_probe.py's question per call site.
usage: python profiles/packed_layer_sites_probe.py [uniform|lidar] [steps]"""
import importlib, os, sys, time, collections
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth")
P = importlib.import_module(PKG + ".pointnet2.pointnet2_utils").pointnet2
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
make = S.lidar_scenes if kind == "lidar" else S.scenes
batches = [torch.from_numpy(make(8, 16384, seed0=s * 8)).to(dev) for s in range(14)]
real = P.packed_layer_wrapper
seen = collections.Counter()


def run(shape):
    def wrapped(a, wt, bias, relu, out, pack=None):
        key = (a.shape[0], wt.shape[0], wt.shape[1], pack is not None)
        if shape is None:
            seen[key] += 1
        elif key == shape:
            real(a, wt, bias, relu, out, pack)
        return real(a, wt, bias, relu, out, pack)
    P.packed_layer_wrapper = wrapped
    runner = E.make_runner(model, cfg, dev)
    best = 1e9
    for rep in range(2):
        for i in range(24):
            runner.submit(batches[i % 14], [batches[(i + d) % 14] for d in range(1, runner.depth + 1)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            runner.submit(batches[i % 14], [batches[(i + d) % 14] for d in range(1, runner.depth + 1)])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e3)
    while runner.flush() is not None:
        pass
    P.packed_layer_wrapper = real
    return best


base = run(None)
print("%s scenes, unchanged: %.4f ms per step" % (kind, base), flush=True)
shapes = sorted(seen, key=lambda k: -k[0] * k[1] * k[2])
for sh in shapes:
    ms = run(sh)
    print("rows %7d K %4d N %4d%s (x%d in the captures, %.2f GFLOP): %.4f ms per step (+%.1f us)" % (
        sh[0], sh[1], sh[2], " packed rows" if sh[3] else "", seen[sh], 2e-9 * sh[0] * sh[1] * sh[2], ms, (ms - base) * 1e3), flush=True)
print("unchanged again: %.4f ms per step" % run(None))
