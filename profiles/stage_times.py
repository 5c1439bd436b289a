"""Per-stage GPU time of the feature pass (main stream only, geometry precomputed), batch of 8 scenes.
usage (on the MI355X box):  python profiles/stage_times.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config")
E = importlib.import_module(PKG + ".eval_rcnn")
F = importlib.import_module(PKG + ".net.fast_infer")
synth = importlib.import_module(PKG + ".synth")
pu = importlib.import_module(PKG + ".pointnet2.pointnet2_utils")

dev = torch.device("cuda", 0)
cfg = C.default_eval_cfg()
model = E.build_model(cfg, dev, seed=0)
eng = F.FastPointRCNN(model, cfg)
pts = torch.from_numpy(synth.scenes(8, 16384, seed0=0)).to(dev)
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, ev))


orig_scale = F.FastPointRCNN._sa_scale
state = {"tag": ""}


def timed_scale(*a, **k):
    orig_scale(*a, **k)
    mark(state["tag"] + " scale(ns=%d,cin=%d)" % (a[3].shape[2], a[5]))


F.FastPointRCNN._sa_scale = staticmethod(timed_scale)


def run():
    marks.clear()
    geo = eng.geometry(pts)
    torch.cuda.synchronize()
    mark("start")
    state["tag"] = "RPN"
    feats = eng._backbone(pts, geo)
    mark("RPN FP levels (4)")
    B, N, _ = pts.shape
    flat = feats.view(B * N, -1)
    rpn_cls = eng.rpn_cls(flat).view(B, N, -1)
    rpn_reg = eng.rpn_reg(flat).view(B, N, -1)
    mark("RPN heads")
    raw = rpn_cls[:, :, 0]
    seg = (torch.sigmoid(raw) > cfg.RPN.SCORE_THRESH).float()
    depth = torch.norm(pts, p=2, dim=2)
    rois, _ = model.rpn.proposal_layer(raw, rpn_reg, pts)
    mark("proposal layer")
    state["tag"] = "RCNN"
    out = eng._rcnn(pts, feats, seg, depth, rois)
    mark("RCNN tail (SA3 + heads)")
    ret = {"rois": rois, "rcnn_cls": out["rcnn_cls"], "rcnn_reg": out["rcnn_reg"]}
    E.postprocess(cfg, ret, B)
    mark("final stage")
    torch.cuda.synchronize()


for _ in range(3):
    run()
acc = {}
for _ in range(5):
    run()
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        acc.setdefault(n1, []).append(e0.elapsed_time(e1))
tot = 0.0
for k, v in acc.items():
    ms = sum(v) / len(v)
    tot += ms
    print("%-34s %7.3f ms" % (k, ms))
print("%-34s %7.3f ms" % ("total feature pass", tot))
