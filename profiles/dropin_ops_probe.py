"""The operators a user of the reference's Python calls through the drop-in modules, timed with HIP events on the
launch stream at the RPN backbone's shapes (B = 8): group_points, three_interpolate, ball_query, gather, FPS excluded (fps_probe.py).
Achieved = algorithmic bytes / time against the 8 TB/s HBM peak.   usage: python profiles/dropin_ops_probe.py [uniform|lidar]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
synth = importlib.import_module(bench.PKG + ".synth")
dev = torch.device("cuda", 0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
make = synth.lidar_scenes if kind == "lidar" else synth.scenes
B = 8
print("| operator (B = 8, %s scenes) | algorithmic MB | us / call | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|" % kind)
def row(name, nbytes, us):
    print("| %s | %.1f | %.1f | %.0f | %.3f |" % (name, nbytes / 1e6, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000), flush=True)

xyz = torch.from_numpy(make(B, 16384, seed0=1000)).to(dev)
levels = [(16384, 4096), (4096, 1024), (1024, 256)]
cur = xyz
for (N, M), C, radii in zip(levels, (128, 96, 256), ((0.2, 32), (1.0, 32), (2.0, 32))):
    temp = torch.full((B, N), 1e10, device=dev); sel = torch.empty((B, M), dtype=torch.int32, device=dev)
    P.furthest_point_sampling_wrapper(B, N, M, cur, temp, sel)
    new_xyz = torch.gather(cur, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    r, ns = radii
    idx = torch.zeros((B, M, ns), dtype=torch.int32, device=dev)
    us = timed(lambda: P.ball_query_wrapper(B, N, M, r, ns, new_xyz, cur, idx))
    row("ball_query N = %d, M = %d, r = %g, nsample = %d" % (N, M, r, ns), B * (12 * N + 12 * M + 4 * M * ns), us)
    feats = torch.randn((B, C, N), device=dev)
    out = torch.empty((B, C, M, ns), device=dev)
    us = timed(lambda: P.group_points_wrapper(B, C, N, M, ns, feats, idx, out))
    row("group_points C = %d, N = %d -> (M = %d, nsample = %d)" % (C, N, M, ns), B * (4 * C * N + 4 * M * ns + 4 * C * M * ns), us)
    xt = cur.transpose(1, 2).contiguous(); gx = torch.empty((B, 3, M, ns), device=dev)
    us = timed(lambda: P.group_points_wrapper(B, 3, N, M, ns, xt, idx, gx))
    row("group_points C = 3 (coordinates), N = %d" % N, B * (12 * N + 4 * M * ns + 12 * M * ns), us)
    d2 = torch.empty((B, N, 3), device=dev); i3 = torch.empty((B, N, 3), dtype=torch.int32, device=dev)
    us = timed(lambda: P.three_nn_wrapper(B, N, M, cur, new_xyz, d2, i3))
    row("three_nn %d <- %d" % (N, M), B * (12 * N + 12 * M + 24 * N), us)
    Ck = {16384: 256, 4096: 512, 1024: 512}[N]
    kf = torch.randn((B, Ck, M), device=dev); w = torch.rand((B, N, 3), device=dev); oi = torch.empty((B, Ck, N), device=dev)
    us = timed(lambda: P.three_interpolate_wrapper(B, Ck, M, N, kf, i3, w, oi))
    row("three_interpolate C = %d, %d <- %d" % (Ck, N, M), B * (4 * Ck * M + 24 * N + 4 * Ck * N), us)
    cur = new_xyz
