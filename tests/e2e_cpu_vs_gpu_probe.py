import importlib, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import ext_cpu
PKG = "3d_adapt_auto_driving_amd"
C = importlib.import_module(PKG + ".config"); E = importlib.import_module(PKG + ".eval_rcnn"); S = importlib.import_module(PKG + ".synth"); F = importlib.import_module(PKG + ".net.fast_infer")
from test_gpu_e2e import match_boxes
cfg = C.default_eval_cfg()
model_c = E.build_model(cfg, "cpu", seed=3)
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for name, p in model_c.named_parameters():
        if ("reg_layer" in name or "cls_layer" in name) and p.dim() > 1:
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    model_c.rcnn_net.cls_layer[-1].conv.weight.mul_(0.05); model_c.rcnn_net.cls_layer[-1].conv.bias.fill_(0.5)
model_g = E.build_model(cfg, "cuda:0", seed=3); model_g.load_state_dict(model_c.state_dict())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pts = torch.from_numpy(S.scenes(B, 16384, seed0=77))
t = time.time()
with ext_cpu.patch_package():
    dc = E.infer_batch(model_c, cfg, pts, engine=F.FastPointRCNN(model_c, cfg))
print("cpu engine %.1f s" % (time.time() - t))
dg = E.infer_batch(model_g, cfg, pts.cuda(), engine=F.FastPointRCNN(model_g, cfg))
for k in ("rois", "rcnn_cls", "rcnn_reg", "boxes", "scores"):
    a, b = dg[k].cpu(), dc[k]
    print(k, "max|d| %.3g" % float((a - b).abs().max()), "equal rows %d / %d" % (int(((a - b).abs().reshape(a.shape[0] if a.dim()<3 else -1, a.shape[-1]).max(-1).values < 1e-4).sum()), a.reshape(-1, a.shape[-1]).shape[0]))
print("num", dg["num"].tolist(), dc["num"].tolist())
for b in range(B):
    w, m = match_boxes(dg["rois"][b].cpu().numpy(), dc["rois"][b].numpy()); print("rois scene", b, "worst %.3g matched %d" % (w, m))
    ng, nc = int(dg["num"][b]), int(dc["num"][b])
    w, m = match_boxes(dg["boxes"][b, :ng].cpu().numpy(), dc["boxes"][b, :nc].numpy()); print("boxes scene", b, "worst %.3g matched %d of %d/%d" % (w, m, ng, nc))
