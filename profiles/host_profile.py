"""cProfile of the host side of the bench's steady loop (runner.submit + async D2H of the detections): where the ~1.4 ms of
host time per step go.  usage: python profiles/host_profile.py [steps]"""
import cProfile, importlib, io, os, pstats, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
batches = [torch.from_numpy(S.scenes(8, 16384, seed0=s * 8)).to(dev) for s in range(16)]   # 16 > look-ahead 12 + 2
runner = E.PipelinedRunner(model, cfg, dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
def loop(n):
    pend = collections.deque()
    for i in range(n):
        det = runner.submit(batches[i % 16], [batches[(i + d) % 16] for d in range(1, runner.depth + 1) if i + d < n])
        if det is not None:
            pend.append(det["ready"])
            if len(pend) > 3: pend.popleft().synchronize()
    runner.drain(); torch.cuda.synchronize()
loop(40)
pr = cProfile.Profile(); pr.enable(); loop(N); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:7000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:9000])
