"""Standalone times of the per-point layer kernel (csrc/packed_layer.hip, host-count mode) at the shapes the B = 8 step runs it
at: rows x K -> N, mean of 20 launches, with the f32 MFMA rate and the algorithmic HBM rate (A read once + out written once)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd")
sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as X
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [(131072, 256, 128, "FP0 layer 1"), (131072, 128, 128, "FP0 layer 2 / RPN heads"), (32768, 640, 256, "FP1 layer 1"),
          (32768, 256, 256, "FP1 layer 2"), (32768, 128, 128, "SA2 per-point part"), (8192, 768, 512, "FP2 layer 1"),
          (8192, 512, 512, "FP2 layer 2"), (8192, 256, 256, "SA3 per-point part"), (2048, 1536, 512, "FP3 layer 1"),
          (2048, 512, 512, "FP3 layer 2 / SA4 per-point"), (25600, 256, 512, "RCNN GroupAll layer"), (800, 512, 256, "RCNN head layer")]

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in ev:
        a.record(); fn(); e.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(e) for a, e in ev]))

print("| rows | K | N | where | us | TF/s (f32) | TB/s (A + out) |"); print("|---|---|---|---|---|---|---|")
for rows, K, N, where in SHAPES:
    a = torch.randn((rows, K), device=dev, generator=g); w = torch.randn((K, N), device=dev, generator=g) / 16
    bias = torch.randn(N, device=dev, generator=g); out = torch.empty((rows, N), device=dev)
    t = timeit(lambda: X.packed_layer_wrapper(a, w, bias, True, out))
    print("| %d | %d | %d | %s | %.1f | %.1f | %.2f |" % (rows, K, N, where, t * 1e3, 2.0 * rows * K * N / t / 1e9, (rows * (K + N) * 4) / t / 1e9))
