"""What do the zero fills of the step cost it?  The product runner with ONE MORE fill of each arena of a kind (the RCNN stage's pooled outputs +
list headers, 160 MB per pair of batches; the geometry group's, 126 MB per group), captured into the same graphs: steady-state ms per step
against the unchanged runner.  usage: python profiles/fill_cost_probe.py [uniform|lidar] [steps]"""
import importlib, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth")
F = importlib.import_module(PKG + ".net.fast_infer")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
make = S.lidar_scenes if kind == "lidar" else S.scenes
batches = [torch.from_numpy(make(8, 16384, seed0=s * 8)).to(dev) for s in range(14)]
real_init = F.ZeroArena.__init__


def run(extra):
    def init(self, key, device):
        real_init(self, key, device)
        if self.buf is not None and key[0] == extra:
            self.buf.zero_()
    F.ZeroArena.__init__ = init
    runner = E.make_runner(model, cfg, dev)
    out = []
    for rep in range(2):
        for i in range(24):
            runner.submit(batches[i % 14], [batches[(i + d) % 14] for d in range(1, runner.depth + 1)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            runner.submit(batches[i % 14], [batches[(i + d) % 14] for d in range(1, runner.depth + 1)])
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / N * 1e3)
    while runner.flush() is not None:
        pass
    F.ZeroArena.__init__ = real_init
    return min(out)


sizes = {}
for extra in (None, "rcnn", "geo", None):
    ms = run(extra)
    print("%s scenes, one more fill of the %-5s arenas: %.4f ms per step" % (kind, extra, ms), flush=True)
print({k[0]: v * 4 / 1e6 for k, v in F.ZeroArena.SIZES.items()}, "MB per arena")
