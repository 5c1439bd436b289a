"""EXPERIMENT: the per-point layer act(A @ W + b) as a three-way bf16 split on the bf16 matrix cores
(csrc/split_bf16.hip, prcnn_rows_layer_bf16x3) beside the product's f32 MFMA kernels (prcnn_packed_layer) at the step's layer shapes:
solo time (HIP events, back-to-back launches), and the error of both against a float64 product of the same f32 operands.
usage: python profiles/split_bf16_probe.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
import pointnet2_cuda as P
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
print("| rows | K | N | f32 MFMA (product) us | TFLOP/s | split-bf16 us | TFLOP/s | speed-up | max abs err f32 / split vs f64 | max rel-to-row-scale err f32 / split |")
print("|---|---|---|---|---|---|---|---|---|---|")
for rows, K, N in ((131072, 128, 128), (262144, 128, 128), (65536, 256, 256), (65536, 256, 128), (32768, 512, 512), (16384, 512, 256), (8192, 512, 512), (2048, 1024, 512)):
    a = torch.randn((rows, K), generator=g).relu_().to(dev)                 # activations behind a ReLU, as in the network
    w = (torch.randn((K, N), generator=g) / np.sqrt(K)).to(dev)
    b = torch.randn((N,), generator=g).to(dev)
    o1 = torch.empty((rows, N), device=dev); o2 = torch.empty((rows, N), device=dev)
    def t(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    us1 = t(lambda: P.packed_layer_wrapper(a, w, b, True, o1))
    us2 = t(lambda: P.rows_layer_bf16x3_wrapper(a, w, b, True, o2))
    n = min(rows, 4096)
    ref = torch.relu(a[:n].double() @ w.double() + b.double())
    scale = (a[:n].double().abs() @ w.double().abs() + b.double().abs())      # sum of |terms|: what the rounding errors scale with
    e1 = (o1[:n].double() - ref).abs(); e2 = (o2[:n].double() - ref).abs()
    fl = 2.0 * rows * K * N
    print("| %d | %d | %d | %.1f | %.1f | %.1f | %.1f | %.2fx | %.2e / %.2e | %.2e / %.2e |" % (
        rows, K, N, us1, fl / us1 / 1e6, us2, fl / us2 / 1e6, us1 / us2, float(e1.max()), float(e2.max()), float((e1 / scale).max()), float((e2 / scale).max())), flush=True)
# A = I check with an asymmetric W: the operand layouts of the instruction
K = N = 128
a = torch.eye(K, device=dev).repeat(2, 1).contiguous(); w = torch.arange(K * N, dtype=torch.float32, device=dev).view(K, N) / 7.0
o = torch.empty((2 * K, N), device=dev)
P.rows_layer_bf16x3_wrapper(a, w, torch.zeros(N, device=dev), False, o)
print("\nA = I, asymmetric W: max |out - W| = %.3e (f32 ulp of the largest entry: %.3e)" % (float((o[:K] - w).abs().max()), float(np.spacing(np.float32(w.max().item())))))
