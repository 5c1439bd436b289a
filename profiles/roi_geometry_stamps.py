"""Where does a RoI cloud's chain in rcnn_roi_geometry_kernel go?  s_memtime stamps at the phase boundaries of a text-instrumented COPY of
csrc/fps.hip (linked with the product's other objects into profiles/_exp/libprcnn_hip_rg_stamps.so; the product library is not touched),
on the RoI clouds of a real step: the inputs of prcnn_rcnn_roi_geometry are recorded from one engine pass over 16 synthetic scenes.

  python profiles/roi_geometry_stamps.py build                 (build container: hipcc)
  python profiles/roi_geometry_stamps.py run [uniform|lidar]   (GPU box): per phase, cycles per cloud (mean / median / p90 over the 1600
      clouds of a launch), and the distribution of `limit` (distinct pooled points per cloud)"""
import ctypes, importlib, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d_adapt_auto_driving_amd", "csrc")
EXP = os.path.join(ROOT, "profiles", "_exp")
LIB = os.path.join(EXP, "libprcnn_hip_rg_stamps.so")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()
PHASES = ["load 512 points", "FPS 512 -> 128", "centres out + ball query 1", "rep map 1", "rows of idx1 out (skipped by the engine)", "row list 1", "FPS 128 -> 32",
          "centres 2 + ball query 2", "rep map 2", "rows of idx2 out (skipped by the engine)", "row lists 2 and 3"]
NPH = len(PHASES)
MAXB = 4096


def instrument():
    s = open(os.path.join(CSRC, "fps.hip")).read()
    s = s.replace('#include "common.hpp"', '#include "%s/common.hpp"' % CSRC).replace('#include "../../include/prcnn_hip.h"', '#include "%s/include/prcnn_hip.h"' % ROOT)
    a = s.index("__global__ __launch_bounds__(64) void rcnn_roi_geometry_kernel(")
    b = s.index("/* RoI clouds xyz (b,512,3) whose points k >= limit[cloud]")
    k = s[a:b]

    def put(old, new, count=1):
        nonlocal k
        assert k.count(old) == count, (k.count(old), old[:70])
        k = k.replace(old, new)
    put("    __builtin_amdgcn_s_setprio(3);\n", "    __builtin_amdgcn_s_setprio(3);\n    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();\n    int ph_ = 0;\n"
        "#define PH { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && b < %d) g_rg_acc[b * %d + ph_] = n_ - t_prev_; t_prev_ = n_; ++ph_; }\n" % (MAXB, NPH))
    put("    const int nd1 = roi_fps_any<8, true>(RG_N, lim, RG_M1, kc1, px, py, pz, s_sel1, lane);\n", "    PH\n    const int nd1 = roi_fps_any<8, true>(RG_N, lim, RG_M1, kc1, px, py, pz, s_sel1, lane);\n    PH\n")
    put("    // representative map of the centres: the first centre sampled from the same source (prcnn_dup_rep)\n", "    PH\n")
    put("    if (idx1) roi_rows_out<RG_M1>(", "    PH\n    if (idx1) roi_rows_out<RG_M1>(")
    put("    if (pk.rowinfo1) {\n        __syncthreads();                                          // s_first is free: the list's offsets\n", "    PH\n    if (pk.rowinfo1) {\n        __syncthreads();\n")
    put("    roi_fps_any<2, false>(RG_M1, nd1, RG_M2, kc2, qx, qy, qz, s_sel2, lane);\n", "    PH\n    roi_fps_any<2, false>(RG_M1, nd1, RG_M2, kc2, qx, qy, qz, s_sel2, lane);\n    PH\n")
    put("    // representative map of level 2's centres through the map of level 1\n", "    PH\n")
    put("    if (idx2) roi_rows_out<RG_M2>(", "    PH\n    if (idx2) roi_rows_out<RG_M2>(")
    # the lists of level 2 and of the GroupAll level, the end of the kernel
    put("    if (pk.rowinfo1) {\n        __syncthreads();\n        roi_pack_out<RG_M2>(", "    PH\n    if (pk.rowinfo1) {\n        __syncthreads();\n        roi_pack_out<RG_M2>(")
    tail = "            pk.rowdxyz3[r] = make_float4(cx[0] - 0.f, cy[0] - 0.f, cz[0] - 0.f, 0.f);\n        }\n    }\n}\n"
    put(tail, tail[:-2] + "    PH\n}\n")
    k = "}\n__device__ unsigned long long g_rg_acc[%d * %d];\nnamespace prcnn {\n" % (MAXB, NPH) + k
    # the kernel sits inside namespace prcnn: close and reopen it around the symbol so that HIP_SYMBOL finds it at file scope
    s = s[:a] + k + s[b:]
    s += ('\nextern "C" int prcnn_debug_rg_acc(unsigned long long *dst)\n{\n    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rg_acc), '
          'sizeof(unsigned long long) * %d * %d);\n}\n' % (MAXB, NPH))
    return s


def build():
    os.makedirs(EXP, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC])
    src = os.path.join(EXP, "rg_stamps.hip")
    open(src, "w").write(instrument())
    obj = os.path.join(EXP, "rg_stamps.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", obj])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "fps.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj] + objs)
    print("built", LIB)


def run(kind):
    import torch
    sys.path.insert(0, ROOT)
    PKG = "3d_adapt_auto_driving_amd"
    L = importlib.import_module(PKG + "._lib")
    L.LIB_PATH = LIB
    C, E, S = (importlib.import_module(PKG + "." + m) for m in ("config", "eval_rcnn", "synth"))
    F = importlib.import_module(PKG + ".net.fast_infer")
    P = importlib.import_module(PKG + ".pointnet2.pointnet2_utils").pointnet2
    dev = "cuda:0"
    cfg = C.default_eval_cfg()
    model = E.build_model(cfg, dev, seed=0)
    make = S.lidar_scenes if kind == "lidar" else S.scenes
    pts = torch.from_numpy(make(16, 16384, seed0=1000)).to(dev)
    seen = []
    real = P.rcnn_roi_geometry_packs_wrapper

    def spy(xyz, limit, *a):
        seen.append((xyz.clone(), limit.clone(), a))
        return real(xyz, limit, *a)
    P.rcnn_roi_geometry_packs_wrapper = spy
    eng = F.FastPointRCNN(model, cfg)
    with torch.no_grad():
        eng.forward(pts)
    torch.cuda.synchronize()
    P.rcnn_roi_geometry_packs_wrapper = real
    xyz, limit, a = seen[0]
    b = xyz.shape[0]

    def again():                                   # the product's call with fresh (zero) list headers
        z = lambda: torch.zeros(4, dtype=torch.int32, device=xyz.device)
        return real(xyz, limit, *a[:6], z(), z(), a[8], a[9], z(), a[11])
    for _ in range(3):
        again()
    torch.cuda.synchronize()
    z3 = [torch.zeros(4, dtype=torch.int32, device=xyz.device) for _ in range(3)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); real(xyz, limit, *a[:6], z3[0], z3[1], a[8], a[9], z3[2], a[11]); ev1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (MAXB * NPH))()
    lib = ctypes.CDLL(LIB)
    assert lib.prcnn_debug_rg_acc(buf) == 0
    acc = np.array(buf, dtype=np.float64).reshape(MAXB, NPH)[:b]
    lim = limit.cpu().numpy()
    print("%s scenes: %d RoI clouds in one launch, %.1f us (instrumented build); distinct pooled points per cloud: median %d, p10 %d, p90 %d, max %d, "
          "clouds with 512: %d" % (kind, b, ev0.elapsed_time(ev1) * 1e3, np.median(lim), np.percentile(lim, 10), np.percentile(lim, 90), lim.max(), int((lim >= 512).sum())))
    print("| phase | mean | median | p90 | share of the mean total |\n|---|---|---|---|---|")
    tot = acc.sum(1)
    for i, name in enumerate(PHASES):
        v = acc[:, i]
        print("| %s | %.0f | %.0f | %.0f | %.2f |" % (name, v.mean(), np.median(v), np.percentile(v, 90), v.mean() / tot.mean()))
    print("| cloud total (s_memtime ticks) | %.0f | %.0f | %.0f | |" % (tot.mean(), np.median(tot), np.percentile(tot, 90)))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "uniform")
