"""FPS timings at the RPN level sizes (B = 8), standalone."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
synth = importlib.import_module("3d_adapt_auto_driving_amd.synth")
dev = torch.device("cuda", 0)
KIND = sys.argv[1] if len(sys.argv) > 1 else "uniform"        # uniform | lidar
xyz = torch.from_numpy((synth.lidar_scenes if KIND == "lidar" else synth.scenes)(8, 16384, seed0=0)).to(dev)
print("(%s scenes)" % KIND)
for n, m in ((16384, 4096), (4096, 1024), (1024, 256), (256, 64)):
    pts = xyz[:, :n].contiguous()
    temp = torch.empty((8, n), device=dev); idx = torch.empty((8, m), dtype=torch.int32, device=dev)
    def run():
        temp.fill_(1e10); P.furthest_point_sampling_wrapper(8, n, m, pts, temp, idx)
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    print("fps %5d -> %4d : %.3f ms" % (n, m, (time.perf_counter() - t0) / 5 * 1e3))
