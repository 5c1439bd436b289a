"""600 steps of the pipelined runner, step time per block of 50 (a synchronize between blocks: each block restarts the
pipeline cold), then clocks / power: no drift over time (1.96-2.07 ms per step, 2.4 GHz, ~560 W)."""
import importlib, os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
batches = [torch.from_numpy(S.scenes(8, 16384, seed0=s * 8)).to(dev) for s in range(16)]
runner = E.PipelinedRunner(model, cfg, dev)
N = 600
import collections
pend = collections.deque()
t_marks = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    det = runner.submit(batches[i % 16], [batches[(i + d) % 16] for d in range(1, runner.depth + 1) if i + d < N])
    if det is not None:
        pend.append(det["ready"])
        if len(pend) > 3: pend.popleft().synchronize()
    if i % 50 == 49:
        torch.cuda.synchronize(); t_marks.append(time.perf_counter())
runner.drain(); torch.cuda.synchronize()
prev = t0
for k, t in enumerate(t_marks):
    print("steps %3d-%3d: %.3f ms/step" % (50 * k, 50 * k + 49, (t - prev) / 50 * 1e3)); prev = t
import subprocess
print(subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|power' | head -4", shell=True, capture_output=True, text=True).stdout)
