"""Single-stream run of the product step (engine forward + final stage on one LAUNCH of the graphed runner: PRCNN_PAIR = 2 batches of 8 scenes
since the second session of round 4; argv[2] = scenes) for rocprofv3 --pmc passes:
counter collection serialises dispatches, so the multi-stream pipeline of bench.py is not used here; the kernels and
their arguments are the same.  usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python profiles/pmc_step_probe.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn")
S = importlib.import_module(PKG + ".synth"); F = importlib.import_module(PKG + ".net.fast_infer")
dev = torch.device("cuda:0")
cfg = C.default_eval_cfg()
model = E.build_model(cfg, dev, seed=0)
eng = F.FastPointRCNN(model, cfg)
NSC = int(sys.argv[2]) if len(sys.argv) > 2 else 8 * max(1, E.RCNN_PAIR)
pts = torch.from_numpy((S.lidar_scenes if (len(sys.argv) > 3 and sys.argv[3] == "lidar") else S.scenes)(NSC, 16384, seed0=0)).to(dev)   # argv[3] = lidar: LiDAR-shaped scenes
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    det = E.infer_batch(model, cfg, pts, engine=eng)
torch.cuda.synchronize()
print("detections", det["num"].tolist())
