"""Unprofiled stage timeline of the pipelined runner: HIP events around every stage on the stream it runs on
(RPN stage, RCNN stage on the feature stream; proposals / final stage on the tail stream; the two links of a geometry
chain on their side stream), over steady-state steps.  Prints per-stage durations and, per step, when each stage
started and ended relative to the step's RPN start."""
import importlib, os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
KIND = sys.argv[1] if len(sys.argv) > 1 else "uniform"          # uniform | lidar
make = S.lidar_scenes if KIND == "lidar" else S.scenes
NB = E.PipelinedRunner.default_depth() + 2
batches = [torch.from_numpy(make(8, 16384, seed0=s * 8)).to(dev) for s in range(NB)]
runner = E.PipelinedRunner(model, cfg, dev)
eng = runner.engine
log = []
origin = torch.cuda.Event(enable_timing=True)

def wrap(obj, name, tag):
    fn = getattr(obj, name)
    def w(*a, **k):
        st = torch.cuda.current_stream(dev)
        a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True)
        a0.record(st); r = fn(*a, **k); a1.record(st)
        log.append((tag, a0, a1)); return r
    setattr(obj, name, w)
wrap(eng, "rpn_stage", "rpn"); wrap(eng, "rcnn_stage", "rcnn"); wrap(eng, "propose", "proposals")
wrap(eng, "rcnn_geometry", "rcnn_geo"); wrap(eng, "rcnn_features", "rcnn")
wrap(eng, "geometry_begin", "geo_begin"); wrap(eng, "geometry_finish", "geo_finish"); wrap(eng, "geometry_group", "geo_group"); wrap(E, "postprocess", "final")
def loop(n):
    for i in range(n):
        runner.submit(batches[i % NB], [batches[(i + d) % NB] for d in range(1, runner.depth + 1)])
    runner.drain()
loop(16); torch.cuda.synchronize(); log.clear()
origin.record(torch.cuda.current_stream(dev))
K = 24
loop(K); torch.cuda.synchronize()
rows = [(tag, origin.elapsed_time(a0), origin.elapsed_time(a1)) for tag, a0, a1 in log]
dur = collections.defaultdict(list)
for tag, s, e in rows: dur[tag].append(e - s)
print(KIND, "stage durations (ms, median / max):", {k: (round(float(np.median(v)), 2), round(max(v), 2)) for k, v in dur.items()})
rp = [r for r in rows if r[0] == "rpn"]
print("step period (ms):", [round(rp[i + 1][1] - rp[i][1], 2) for i in range(8, 20)])
for i in range(10, 14):
    t0 = rp[i][1]
    near = [(tag, round(s - t0, 2), round(e - t0, 2)) for tag, s, e in rows if t0 - 0.01 <= s < rp[i + 1][1]]
    print("step", i, near)
