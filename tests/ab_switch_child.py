"""Child process of tests/test_gpu_switches.py: one closed loop of the eager pipelined runner + one serial engine pass on fixed scenes with the
environment it was started in (an A/B switch set, or none); writes the detections to argv[1].  Not a test module."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C, E, S, F = (importlib.import_module(PKG + "." + m) for m in ("config", "eval_rcnn", "synth", "net.fast_infer"))
dev = "cuda:0"
cfg = C.default_eval_cfg()
model = E.build_model(cfg, dev, seed=3)
B = 4
batches = [torch.from_numpy((S.lidar_scenes if k % 2 else S.scenes)(B, cfg.RPN.NUM_POINTS, seed0=500 + 10 * k)).to(dev) for k in range(6)]
runner = E.PipelinedRunner(model, cfg, dev)
dets = []
for i, x in enumerate(batches):
    d = runner.submit(x, batches[i + 1:])
    if d is not None:
        dets.append(d)
dets += runner.drain()
assert len(dets) == len(batches)
out = {}
for i, d in enumerate(dets):
    if "ready" in d:
        d["ready"].synchronize()
    for k in ("rois", "boxes", "scores", "num", "rcnn_cls", "rcnn_reg"):
        out["r%d_%s" % (i, k)] = d[k].detach().cpu().numpy()
det = E.infer_batch(model, cfg, batches[1], engine=F.FastPointRCNN(model, cfg))
torch.cuda.synchronize()
for k in ("rois", "boxes", "scores", "num"):
    out["serial_%s" % k] = det[k].detach().cpu().numpy()
np.savez(sys.argv[1], **out)
print("ok", sum(int(d["num"].sum()) for d in dets))
