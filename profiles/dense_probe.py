"""Worst case for the distinct-row kernels: scenes shrunk until every ball is FULL (no copies to skip)."""
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth"); F = importlib.import_module(PKG + ".net.fast_infer")
dev = torch.device("cuda:0"); cfg = C.default_eval_cfg(); model = E.build_model(cfg, dev, seed=0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
batches = [torch.from_numpy((S.scenes(8, 16384, seed0=8 * s) * np.float32(scale)).astype(np.float32)).to(dev) for s in range(10)]
eng = F.FastPointRCNN(model, cfg)
d = E.infer_batch(model, cfg, batches[0], engine=eng)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(d["boxes"]).all()), "num", d["num"].tolist())
F.USE_PACKED, F.USE_POOL_DEDUP = False, False
d2 = E.infer_batch(model, cfg, batches[0], engine=F.FastPointRCNN(model, cfg))
F.USE_PACKED, F.USE_POOL_DEDUP = True, True
print("packed vs all-rows engine: rois max|d| %.3g, boxes max|d| %.3g" % (float((d["rois"] - d2["rois"]).abs().max()), float((d["boxes"] - d2["boxes"]).abs().max())))
runner = E.PipelinedRunner(model, cfg, dev)
def loop(n):
    for i in range(n):
        runner.submit(batches[i % 10], [batches[(i + k) % 10] for k in range(1, runner.depth + 1)])
    runner.drain()
loop(10); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(40); torch.cuda.synchronize()
print("dense scenes (scale %.3f): %.1f scenes/s" % (scale, 40 * 8 / (time.perf_counter() - t0)))
# packed vs all-rows engines on the dense batch: the network outputs before any discrete decision
geo = eng.geometry(batches[0])
st1 = eng.rpn_stage(batches[0], geo)
F.USE_PACKED, F.USE_POOL_DEDUP, F.PAD128 = False, False, False
eng2 = F.FastPointRCNN(model, cfg)
st2 = eng2.rpn_stage(batches[0], eng2.geometry(batches[0]))
F.USE_PACKED, F.USE_POOL_DEDUP, F.PAD128 = True, True, True
for k in ("rpn_features", "rpn_cls", "rpn_reg"):
    a, b = st1[k], st2[k]
    print(k, "max|d| %.3g  max|v| %.3g" % (float((a - b).abs().max()), float(b.abs().max())))
rois, _ = eng.propose(st1)
r1 = eng.rcnn_stage(st1, rois)
F.USE_PACKED, F.USE_POOL_DEDUP, F.PAD128 = False, False, False
r2 = eng2.rcnn_stage(st1, rois)
F.USE_PACKED, F.USE_POOL_DEDUP, F.PAD128 = True, True, True
for k in ("rcnn_cls", "rcnn_reg"):
    print(k, "same RoIs: max|d| %.3g  max|v| %.3g" % (float((r1[k] - r2[k]).abs().max()), float(r2[k].abs().max())))
