"""Per-extension-call timing of one stage of the engine, standalone (HIP events around every call of the drop-in modules).
usage: python profiles/call_probe.py [uniform|lidar] [stage: rcnn_geo|rcnn|rpn|proposals|final|geometry]"""
import importlib, os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth")
F = importlib.import_module(PKG + ".net.fast_infer"); pu = importlib.import_module(PKG + ".pointnet2.pointnet2_utils")
ru = importlib.import_module(PKG + ".roipool3d_utils"); iu = importlib.import_module(PKG + ".iou3d_utils")
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
kind = sys.argv[1] if len(sys.argv) > 1 else "lidar"
stage = sys.argv[2] if len(sys.argv) > 2 else "rcnn_geo"
make = S.lidar_scenes if kind == "lidar" else S.scenes
eng = F.FastPointRCNN(model, cfg)
pts = torch.from_numpy(make(8, 16384, seed0=0)).to(dev)
log = []


class Timed:
    def __init__(self, mod):
        self._m = mod

    def __getattr__(self, name):
        fn = getattr(self._m, name)
        if not callable(fn) or name.endswith("_supported"):
            return fn

        def call(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a, **k); e1.record()
            log.append((name, e0, e1))
            return r
        return call


def run():
    if stage == "geometry":
        return eng.geometry(pts)
    st = eng.rpn_stage(pts)
    if stage == "rpn":
        return st
    rois, _ = eng.propose(st)
    if stage == "proposals":
        return rois
    rg = eng.rcnn_geometry(st, rois)
    if stage == "rcnn_geo":
        return rg
    out = eng.rcnn_features(rg)
    return out


for _ in range(3):
    run()
torch.cuda.synchronize()
saved = (pu.pointnet2, ru.roipool3d_cuda)
pu.pointnet2, ru.roipool3d_cuda = Timed(saved[0]), Timed(saved[1])
# time only the calls of the requested stage: everything before it runs un-instrumented
pre = {"rcnn_geo": 2, "rcnn": 3}.get(stage, 0)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
try:
    for rep in range(5):
        log.clear()
        torch.cuda.synchronize()
        run()
        torch.cuda.synchronize()
finally:
    pu.pointnet2, ru.roipool3d_cuda = saved
agg = collections.OrderedDict()
for name, e0, e1 in log:
    agg.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
print("## %s scenes, calls up to and including stage %s (one pass, us per call incl. its torch glue in between excluded)" % (kind, stage))
tot = 0.0
for name, v in agg.items():
    print("%-38s x%-2d  %s" % (name, len(v), " ".join("%7.1f" % x for x in v)))
    tot += sum(v)
print("sum of extension calls: %.1f us" % tot)
