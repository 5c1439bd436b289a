"""fps_spec_kernel under an experiment switch (PRCNN_FPS_LAYOUT, PRCNN_FPS_* of csrc/fps.hip): time of the product's launches -- 32 clouds,
16384 -> 4096 and 4096 -> 1024 (the level-1 centres of the same clouds) -- on uniform and LiDAR-shaped scenes, HIP events, median of 7;
the picks are written to gpurun_out/r06/fps_picks_<tag>.npz so that two runs can be compared bit for bit (the switch is read once per
process).   usage: python profiles/fps_layout_probe.py <tag> [compare_tag]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
if os.environ.get("FPS_PROBE_LIB"):                 # A/B against another build of the library on the same box (experiments only)
    L = importlib.import_module("3d_adapt_auto_driving_amd._lib"); L.LIB_PATH = os.environ["FPS_PROBE_LIB"]
import pointnet2_cuda as P
synth = importlib.import_module("3d_adapt_auto_driving_amd.synth")
dev = torch.device("cuda", 0)
tag = sys.argv[1]
out = {}
for kind in ("uniform", "lidar"):
    make = synth.lidar_scenes if kind == "lidar" else synth.scenes
    xyz = torch.from_numpy(make(32, 16384, seed0=0)).to(dev)
    cur = xyz
    for n, m in ((16384, 4096), (4096, 1024)):
        temp = torch.empty((32, n), device=dev); idx = torch.empty((32, m), dtype=torch.int32, device=dev)
        def run():
            temp.fill_(1e10); P.furthest_point_sampling_wrapper(32, n, m, cur, temp, idx)
        for _ in range(2): run()
        ts = []
        for _ in range(7):
            temp.fill_(1e10)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); P.furthest_point_sampling_wrapper(32, n, m, cur, temp, idx); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print("%s %-7s 32 x (%5d -> %4d): %.3f ms (min %.3f)" % (tag, kind, n, m, float(np.median(ts)), min(ts)), flush=True)
        out["%s_%d" % (kind, n)] = idx.cpu().numpy(); out["%s_%d_t" % (kind, n)] = temp.cpu().numpy()
        cur = torch.gather(cur, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "r06", "fps_picks_%s.npz" % tag), **out)
if len(sys.argv) > 2 and os.path.exists(os.path.join(ROOT, "gpurun_out", "r06", "fps_picks_%s.npz" % sys.argv[2])):
    ref = np.load(os.path.join(ROOT, "gpurun_out", "r06", "fps_picks_%s.npz" % sys.argv[2]))
    bad = [k for k in out if not np.array_equal(out[k], ref[k])]
    print("%s vs %s: %s" % (tag, sys.argv[2], "IDENTICAL picks and running minima" if not bad else "DIFFERS in %s" % bad))
