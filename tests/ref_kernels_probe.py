"""The REFERENCE'S OWN kernels (oracle/_ref/*.so: its .cu files compiled for gfx950 from the sources in place, oracle/Makefile) timed on
the MI355X beside this build's kernels for the same operator on the same inputs (B = 8 scenes; uniform and LiDAR-shaped clouds).
Operator level only -- the reference's Python cannot travel to the GPU box.  usage: python tests/ref_kernels_probe.py"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P, iou3d_cuda as I, roipool3d_cuda as R
from oracle import ref_gpu as G
S = importlib.import_module("3d_adapt_auto_driving_amd.synth")
dev = torch.device("cuda", 0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("| operator (B = 8) | scene | reference kernel on MI355X (ms) | this build (ms) | ratio |\n|---|---|---|---|---|")
def row(name, kind, tr, tm):
    print("| %s | %s | %.3f | %.3f | %.1fx |" % (name, kind, tr, tm, tr / tm))

B, N, M = 8, 16384, 4096
for kind in ("uniform", "lidar"):
    make = S.lidar_scenes if kind == "lidar" else S.scenes
    xyz = torch.from_numpy(make(B, N, seed0=0)).to(dev)
    temp = torch.empty((B, N), device=dev); sel = torch.empty((B, M), dtype=torch.int32, device=dev)
    tm = timed(lambda: (temp.fill_(1e10), P.furthest_point_sampling_wrapper(B, N, M, xyz, temp, sel)), 3)
    tr = timed(lambda: G.furthest_point_sample(xyz, M), 3)
    row("furthest_point_sampling 16384 -> 4096", kind, tr, tm)
    new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    for r, ns in ((0.2, 32), (0.5, 32), (0.4, 64)):
        idx = torch.zeros((B, M, ns), dtype=torch.int32, device=dev)
        tm = timed(lambda: P.ball_query_wrapper(B, N, M, r, ns, new_xyz, xyz, idx))
        tr = timed(lambda: G.ball_query(r, ns, xyz, new_xyz))
        row("ball_query r = %.1f, nsample = %d" % (r, ns), kind, tr, tm)
    feats = torch.randn((B, 128, N), device=dev)
    idx = G.ball_query(0.2, 32, xyz, new_xyz)
    out = torch.empty((B, 128, M, 32), device=dev)
    tm = timed(lambda: P.group_points_wrapper(B, 128, N, M, 32, feats, idx, out))
    tr = timed(lambda: G.group_points(feats, idx))
    row("group_points C = 128, nsample = 32 (550 MB out)", kind, tr, tm)
    qout = torch.empty((B, 131, M, 32), device=dev); qidx = torch.empty((B, M, 32), dtype=torch.int32, device=dev)
    tq = timed(lambda: P.query_and_group_wrapper(B, N, M, 128, 0.2, 32, new_xyz, xyz, feats, qidx, qout))
    xt = xyz.transpose(1, 2).contiguous()
    def ref_qg():                                    # pointnet2_utils.py:241-264: ball query, two groupings, centre subtraction, cat
        i = G.ball_query(0.2, 32, xyz, new_xyz)
        gx = G.group_points(xt, i); gx -= new_xyz.transpose(1, 2).unsqueeze(-1)
        return torch.cat([gx, G.group_points(feats, i)], dim=1)
    row("QueryAndGroup (ball query + group xyz + group features + cat) vs prcnn_query_and_group", kind, timed(ref_qg), tq)
    d2 = torch.empty((B, N, 3), device=dev); i3 = torch.empty((B, N, 3), dtype=torch.int32, device=dev)
    tm = timed(lambda: P.three_nn_wrapper(B, N, M, xyz, new_xyz, d2, i3))
    tr = timed(lambda: G.three_nn(xyz, new_xyz))
    row("three_nn 16384 <- 4096", kind, tr, tm)
    kf = torch.randn((B, 256, M), device=dev); w = torch.rand((B, N, 3), device=dev); oi = torch.empty((B, 256, N), device=dev)
    tm = timed(lambda: P.three_interpolate_wrapper(B, 256, M, N, kf, i3, w, oi))
    tr = timed(lambda: G.three_interpolate(kf, i3, w))
    row("three_interpolate C = 256", kind, tr, tm)
    rng = np.random.default_rng(3)
    sys.path.insert(0, os.path.join(ROOT, "tests")); from helpers import boxes3d, bev_boxes
    boxes = torch.from_numpy(np.stack([boxes3d(rng, 100, xz_scope=((-12, 12), (6, 40))) for _ in range(B)])).to(dev)
    pf = torch.randn((B, N, 130), device=dev)
    pooled = torch.zeros((B, 100, 512, 133), device=dev); empty = torch.zeros((B, 100), dtype=torch.int32, device=dev)
    tm = timed(lambda: R.forward(xyz, boxes, pf, pooled, empty))
    tr = timed(lambda: G.roipool3d(xyz, boxes, pf, 512))
    row("roipool3d 100 boxes x 512 x 133", kind, tr, tm)
rng = np.random.default_rng(4)
for n, th, rot in ((6300, 0.8, False), (100, 0.1, True), (2000, 0.3, True)):
    bx = torch.from_numpy(bev_boxes(rng, n, spread=30.0 if n > 1000 else 6.0, rotated=rot)).to(dev)
    keep = torch.zeros(n, dtype=torch.int64)
    tm = timed(lambda: (I.nms_gpu if rot else I.nms_normal_gpu)(bx, keep, th))
    tr = timed(lambda: G.nms(bx, th, rot))
    row("nms%s n = %d, thresh %.1f (blocking API; reference: mask kernel + D2H + host reduce restated in numpy)" % ("" if rot else "_normal", n, th), "-", tr, tm)
