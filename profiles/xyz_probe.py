"""RPN SA1 (coordinates-only level) over the packed rows of one bench batch: time of the two scales, standalone."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
synth = importlib.import_module("3d_adapt_auto_driving_amd.synth")
dev = torch.device("cuda", 0)
xyz = torch.from_numpy(synth.scenes(8, 16384, seed0=0)).to(dev)
sel = torch.empty((8, 4096), dtype=torch.int32, device=dev); tmp = torch.full((8, 16384), 1e10, device=dev)
P.furthest_point_sampling_wrapper(8, 16384, 4096, xyz, tmp, sel)
new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
g = torch.Generator(device=dev).manual_seed(0)
for (r, ns, c1, c2, c3) in ((0.1, 16, 16, 16, 32), (0.5, 32, 32, 32, 64)):
    idx = torch.zeros((8, 4096, ns), dtype=torch.int32, device=dev)
    P.ball_query_wrapper(8, 16384, 4096, r, ns, new_xyz, xyz, idx)
    pk = P.ball_pack_wrapper(idx, xyz, new_xyz)
    W = lambda *s: torch.randn(s, device=dev, generator=g) / s[0] ** 0.5
    w1, w2, w3 = W(3, c1), W(c1, c2), W(c2, c3)
    b1, b2, b3 = (torch.randn(c, device=dev, generator=g) * 0.1 for c in (c1, c2, c3))
    out = torch.zeros((8, 4096, c3), device=dev)
    run = lambda: P.sa_xyz_mlp_packed_wrapper(new_xyz, xyz, pk, w1, b1, w2, b2, w3, b3, out, 0, True)
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    hdr = pk.hdr.cpu().numpy()
    print("r=%.1f ns=%d %d-%d-%d: %.1f us  (%d live tiles of %d, %d rows)" % (r, ns, c1, c2, c3, (time.perf_counter() - t0) / 20 * 1e6, hdr[0], pk.max_tiles, hdr[1]))
