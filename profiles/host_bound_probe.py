"""Is the step host-bound?  Time the Python enqueue loop (no sync) against the wall time including the final sync."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
batches = [torch.from_numpy(synth.scenes(8, 16384, seed0=s * 8)).to(dev) for s in range(6)]
runner = E.PipelinedRunner(model, cfg, dev)
def loop(n):
    for i in range(n):
        runner.submit(batches[i % 6], [batches[(i + d) % 6] for d in range(1, runner.depth + 1)])
    runner.flush()
loop(12); torch.cuda.synchronize()
K = 60
t0 = time.perf_counter(); loop(K); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.3f ms/step, wall %.3f ms/step, GPU tail after last enqueue %.1f ms" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
