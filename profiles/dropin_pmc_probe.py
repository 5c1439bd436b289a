"""The two drop-in operators VERDICT r3 W8 named, alone, at the flagged shapes, for rocprofv3 --pmc passes (one shape per kernel name, so that
pmc_step_summarize.py's per-kernel medians are per-shape figures): group_points (B = 8, C = 128, N = 16384 -> M = 4096 x nsample = 32;
algorithmic 608.2 MB) and three_interpolate (B = 8, C = 256, m = 4096 -> n = 16384; algorithmic 170.9 MB)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
synth = importlib.import_module(bench.PKG + ".synth")
dev = torch.device("cuda", 0)
B, N, M, ns = 8, 16384, 4096, 32
xyz = torch.from_numpy(synth.scenes(B, N, seed0=1000)).to(dev)
temp = torch.full((B, N), 1e10, device=dev); sel = torch.empty((B, M), dtype=torch.int32, device=dev)
P.furthest_point_sampling_wrapper(B, N, M, xyz, temp, sel)
new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = torch.zeros((B, M, ns), dtype=torch.int32, device=dev)
P.ball_query_wrapper(B, N, M, 0.2, ns, new_xyz, xyz, idx)
feats = torch.randn((B, 128, N), device=dev); out = torch.empty((B, 128, M, ns), device=dev)
d2 = torch.empty((B, N, 3), device=dev); i3 = torch.empty((B, N, 3), dtype=torch.int32, device=dev)
P.three_nn_wrapper(B, N, M, xyz, new_xyz, d2, i3)
kf = torch.randn((B, 256, M), device=dev); w = torch.rand((B, N, 3), device=dev); oi = torch.empty((B, 256, N), device=dev)
for _ in range(6):
    P.group_points_wrapper(B, 128, N, M, ns, feats, idx, out)
    P.three_interpolate_wrapper(B, 256, M, N, kf, i3, w, oi)
torch.cuda.synchronize()
print("group_points algorithmic MB %.1f, three_interpolate algorithmic MB %.1f" % (
    B * (4 * 128 * N + 4 * M * ns + 4 * 128 * M * ns) / 1e6, B * (4 * 256 * M + 24 * N + 4 * 256 * N) / 1e6))
