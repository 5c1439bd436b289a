"""The xyz-only geometry chain of one GROUP (4 batches = 32 clouds), op by op, standalone, on the uniform scene of SURVEY 8d
and on LiDAR-shaped scenes (synth.lidar_scene): FPS, ball query, three-NN, row packing per level and scale.
usage: python profiles/geo_probe.py [uniform|lidar|both] [clouds]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
synth = importlib.import_module("3d_adapt_auto_driving_amd.synth")
pu = importlib.import_module("3d_adapt_auto_driving_amd.pointnet2.pointnet2_utils")
dev = torch.device("cuda", 0)
kinds = ("uniform", "lidar") if len(sys.argv) < 2 or sys.argv[1] == "both" else (sys.argv[1],)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
LEVELS = ((4096, ((0.1, 16), (0.5, 32))), (1024, ((0.5, 16), (1.0, 32))), (256, ((1.0, 16), (2.0, 32))), (64, ((2.0, 16), (4.0, 32))))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for kind in kinds:
    make = synth.lidar_scenes if kind == "lidar" else synth.scenes
    xyz = torch.from_numpy(make(8, 16384, seed0=0)).to(dev).repeat((B + 7) // 8, 1, 1)[:B].contiguous()
    print("## %s scenes, %d clouds" % (kind, B))
    cur, l_xyz, total = xyz, [xyz], 0.0
    for m, scales in LEVELS:
        n = cur.shape[1]
        t = timed(lambda: pu.furthest_point_sample(cur, m), 3)
        total += t
        print("fps %5d -> %4d : %8.3f ms" % (n, m, t))
        sel = pu.furthest_point_sample(cur, m)
        new = torch.gather(cur, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        for r, ns in scales:
            t = timed(lambda: pu.ball_query(r, ns, cur, new))
            idx = pu.ball_query(r, ns, cur, new)
            t2 = timed(lambda: P.ball_pack_wrapper(idx, cur, new))
            pk = P.ball_pack_wrapper(idx, cur, new)
            hdr = pk.hdr.cpu().numpy()
            total += t + t2
            print("  ball_query n=%5d m=%4d r=%.1f ns=%2d : %8.3f ms   pack %6.3f ms  distinct rows %.3f" % (n, m, r, ns, t, t2, hdr[1] / float(B * m * ns)))
        cur = new
        l_xyz.append(new)
    for k in range(4):
        t = timed(lambda: pu.three_nn(l_xyz[k], l_xyz[k + 1]))
        total += t
        print("three_nn unknown %5d known %4d : %8.3f ms" % (l_xyz[k].shape[1], l_xyz[k + 1].shape[1], t))
    print("sum of the chain's geometry ops: %.2f ms" % total)
