"""The reference's blocking NMS entries (iou3d_cuda.nms_gpu / nms_normal_gpu: mask kernel + resolve kernel + D2H of the keep list) alone, on
boxes that (almost) all survive -- the resolve's worst case: every row is walked.  usage: python profiles/nms_probe.py"""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
import iou3d_cuda as I
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
def boxes(n, rot):
    c = rng.uniform(-35, 35, (n, 2)); d = rng.uniform(1.5, 4.5, (n, 2))
    a = rng.uniform(-3, 3, (n, 1)) if rot else np.zeros((n, 1))
    return torch.from_numpy(np.concatenate([c - d / 2, c + d / 2, a], 1).astype(np.float32)).to(dev)
for n, rot, fn in ((6300, False, I.nms_normal_gpu), (2700, False, I.nms_normal_gpu), (6300, True, I.nms_gpu), (2000, True, I.nms_gpu)):
    b = boxes(n, rot); keep = torch.zeros(n, dtype=torch.int64)
    for _ in range(3): k = fn(b, keep, 0.8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): k = fn(b, keep, 0.8)
    torch.cuda.synchronize()
    print("n %d rotated %s: %.3f ms per call, kept %d" % (n, rot, (time.perf_counter() - t0) / 10 * 1e3, k))
