"""One layer shape of the FP chain through prcnn_packed_layer at growing K: the slope is the cost of a 128-deep panel, the intercept what
a launch pays besides its MFMAs (prologue, epilogue, dispatch).  rows x N fixed; HIP-event medians over regions of SWEEP_REP back-to-back launches, alone on the GPU.
usage: python profiles/layer_k_sweep.py [rows N]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"
pu = importlib.import_module(PKG + ".pointnet2.pointnet2_utils"); ext = pu.pointnet2
dev = torch.device("cuda", 0)
shapes = [(131072, 256), (32768, 512), (8192, 512), (131072, 128)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
print("| rows | N | K | us | TFLOP/s | of 157.3 |\n|---|---|---|---|---|---|")
for rows, N in shapes:
    for K in (128, 256, 512, 1024, 2048):
        a = torch.randn((rows, K), device=dev); w = torch.randn((K, N), device=dev) * 0.05; b = torch.randn((N,), device=dev)
        out = torch.empty((rows, N), device=dev)
        ts = []
        REP = int(os.environ.get("SWEEP_REP", "10"))          # launches per timed region, back to back (one launch per region measures
        for it in range(12):                                   # the idle -> busy transition of the chip as well: +50 us)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP): ext.packed_layer_wrapper(a, w, b, True, out)
            e1.record(); torch.cuda.synchronize()
            if it >= 2: ts.append(e0.elapsed_time(e1) * 1e3 / REP)
        us = float(np.median(ts)); tf = 2.0 * rows * N * K / us / 1e6
        print("| %d | %d | %d | %.1f | %.1f | %.2f |" % (rows, N, K, us, tf, tf / 157.3), flush=True)
        del a, w, out
