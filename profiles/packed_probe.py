"""Times of the fused SA-MLP kernels at the RCNN SA1 shape (800 clouds x 128 centres x 64 samples, 128-128-128):
unpacked (all rows), packed on full balls, packed on ball-query-like rows with a given mean number of distinct rows."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd")
sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as X
dev = torch.device("cuda:0")
b, n, m, ns, c3 = 800, 512, 128, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device=dev).manual_seed(0)
xyz = torch.randn((b, n, 3), device=dev, generator=g); new_xyz = xyz[:, :m].contiguous()
P = torch.randn((b, n, 128), device=dev, generator=g); wx = torch.randn((3, 128), device=dev, generator=g)
w2 = torch.randn((128, 128), device=dev, generator=g) / 11; w3 = torch.randn((128, c3), device=dev, generator=g) / 11
b2 = torch.randn(128, device=dev, generator=g); b3 = torch.randn(c3, device=dev, generator=g)
out = torch.empty((b, m, c3), device=dev)
full = torch.argsort(torch.rand((b, m, n), device=dev, generator=g), dim=2)[:, :, :ns].to(torch.int32).contiguous()

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in ev:
        a.record(); fn(); e.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(e) for a, e in ev]))

flops = 2.0 * b * m * ns * (128 * 128 + 128 * c3)
t = timeit(lambda: X.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, full, w2, b2, w3, b3, out, 0))
print("unpacked, all rows      : %.3f ms  %.1f TF/s" % (t, flops / t / 1e9))
pk = X.ball_pack_wrapper(full, xyz, new_xyz)
t = timeit(lambda: X.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, out, 0))
print("packed, full balls      : %.3f ms  %.1f TF/s (tiles %d)" % (t, flops / t / 1e9, int(pk.hdr[0])))
for mean in (32, 14, 4):
    cnt = torch.clamp((torch.rand((b, m, 1), device=dev, generator=g) * 2 * mean).long() + 1, max=ns)
    idx = torch.where(torch.arange(ns, device=dev).view(1, 1, ns) < cnt, torch.sort(full, dim=2).values, torch.sort(full, dim=2).values[:, :, :1]).contiguous()
    pk = X.ball_pack_wrapper(idx, xyz, new_xyz)
    rows = int(pk.hdr[1])
    t = timeit(lambda: X.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, out, 0))
    tp = timeit(lambda: X.ball_pack_wrapper(idx, xyz, new_xyz))
    print("packed, mean %2d distinct : %.3f ms  %.1f TF/s on the distinct rows (%d tiles), pack %.3f ms" % (mean, t, 2.0 * rows * (128 * 128 + 128 * c3) / t / 1e9, int(pk.hdr[0]), tp))
