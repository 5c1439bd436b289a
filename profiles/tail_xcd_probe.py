"""rpn_tail_lin_kernel on the bench's own roofline leg (real three-NN tables of 8 synthetic scenes): launch time with the tiles drawn per XCD
partition (default) or from one counter (PRCNN_TAIL_XCD=0).  usage: [PRCNN_TAIL_XCD=0] python profiles/tail_xcd_probe.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench, torch
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn")
dev = torch.device("cuda", 0); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
for rep in range(3):
    r = bench.roofline_rpn_tail(dev, cfg, model)
    print("PRCNN_TAIL_XCD=%s: launch %.4f ms, frac %.4f" % (os.environ.get("PRCNN_TAIL_XCD", "1"), r["launch_ms"], r["frac"]))
