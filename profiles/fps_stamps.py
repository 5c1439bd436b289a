"""Where does a round of the speculative FPS kernel go?  s_memtime stamps at the phase boundaries of csrc/fps.hip fps_spec_kernel,
accumulated per wave of workgroup 0 over the whole run (a text-instrumented COPY of the source, linked with the product's other objects
into profiles/_exp/libprcnn_hip_fps_stamps.so; the product library is not touched).

  python profiles/fps_stamps.py build               (build container: hipcc)
  python profiles/fps_stamps.py run [uniform|lidar] (GPU box): 16384 -> 4096, B = 8; per wave: rounds, and cycles per round spent in
      rebuild + publish | waiting at barrier A | the pair tests of the merge | waiting at barrier A2 | the verdict + the pivot permute
      (every wave since round 6: no barrier B) | - | distance updates"""
import ctypes, importlib, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d_adapt_auto_driving_amd", "csrc")
EXP = os.path.join(ROOT, "profiles", "_exp")
LIB = os.path.join(EXP, "libprcnn_hip_fps_stamps.so")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()
NPH = 8


def instrument():
    s = open(os.path.join(CSRC, "fps.hip")).read()
    s = s.replace('#include "common.hpp"', '#include "%s/common.hpp"' % CSRC).replace('#include "../../include/prcnn_hip.h"', '#include "%s/include/prcnn_hip.h"' % ROOT)
    a = s.index("template <int PPT>\n__global__ __launch_bounds__(1024) void fps_spec_kernel(")
    b = s.index("// The speculative kernel for 16384 < n <= 32768")
    k = s[a:b]

    def put(old, new):
        nonlocal k
        assert k.count(old) == 1, old[:70]
        k = k.replace(old, new)
    put("    while (j < m) {\n", "    unsigned long long acc_[%d] = {0}, t_prev_ = __builtin_amdgcn_s_memtime();\n#define PH(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_[i] += n_ - t_prev_; t_prev_ = n_; }\n    while (j < m) {\n        acc_[7] += 1;\n" % NPH)
    put("        lds_barrier();                                                // A: the table is complete\n", "        PH(0)\n        lds_barrier();                                                // A: the table is complete\n        PH(1)\n")
    put("        lds_barrier();                                                // A2: every entry has its rank\n", "        PH(2)\n        lds_barrier();                                                // A2: every entry has its rank\n        PH(3)\n")
    put("        // ---- 4. running minima against this round's pivots (the last pick of the whole run does not update them: sampling_gpu.cu)\n", "        PH(4)\n        PH(5)\n        // ---- 4. running minima against this round's pivots (the last pick of the whole run does not update them: sampling_gpu.cu)\n")
    put("    if (mind) {\n#pragma unroll\n        for (int i = 0; i < PPT; ++i)\n            if (pc[i] != 0xffffffffu) mind[kc.decode(pc[i] >> SB)] = pt[i];\n    }\n}",
        "    if (mind) {\n#pragma unroll\n        for (int i = 0; i < PPT; ++i)\n            if (pc[i] != 0xffffffffu) mind[kc.decode(pc[i] >> SB)] = pt[i];\n    }\n"
        "    if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < %d; ++i) g_fps_acc[w * %d + i] = acc_[i];\n}" % (NPH, NPH))
    # the update phase ends at the bottom of the while loop: stamp right before its closing brace = before 'if (mind)'
    put("                    update(ox, oy, oz, mk);\n                }\n            }\n        }\n    }\n", "                    update(ox, oy, oz, mk);\n                }\n            }\n        }\n        PH(6)\n    }\n")
    k = "__device__ unsigned long long g_fps_acc[16 * %d];\n" % NPH + k
    s = s[:a] + k + s[b:]
    s += ('\nextern "C" int prcnn_debug_fps_acc(unsigned long long *dst)\n{\n    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(prcnn::g_fps_acc), '
          'sizeof(unsigned long long) * 16 * %d);\n}\n' % NPH)
    return s


def build():
    os.makedirs(EXP, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC])
    src = os.path.join(EXP, "fps_stamps.hip")
    open(src, "w").write(instrument())
    obj = os.path.join(EXP, "fps_stamps.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", obj])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "fps.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj] + objs)
    print("built", LIB)


def run(kind):
    import torch
    sys.path.insert(0, ROOT)
    L = importlib.import_module("3d_adapt_auto_driving_amd._lib")
    L.LIB_PATH = LIB
    pkg = importlib.import_module("3d_adapt_auto_driving_amd"); sys.path.insert(0, pkg.DROPIN_DIR)
    import pointnet2_cuda as P
    synth = importlib.import_module("3d_adapt_auto_driving_amd.synth")
    dev = torch.device("cuda", 0)
    xyz = torch.from_numpy((synth.lidar_scenes if kind == "lidar" else synth.scenes)(8, 16384, seed0=0)).to(dev)
    n, m = 16384, 4096
    temp = torch.empty((8, n), device=dev); idx = torch.empty((8, m), dtype=torch.int32, device=dev)
    for _ in range(3):
        temp.fill_(1e10); P.furthest_point_sampling_wrapper(8, n, m, xyz, temp, idx)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (16 * NPH))()
    lib = ctypes.CDLL(LIB)
    assert lib.prcnn_debug_fps_acc(buf) == 0
    a = np.array(buf, dtype=np.float64).reshape(16, NPH)
    rounds = a[:, 7]
    print("%s scenes, cloud 0, 16384 -> 4096: %d rounds; s_memtime cycles per round by wave" % (kind, int(rounds[0])))
    print("wave | rebuild+publish | wait A | pairs | wait A2 | verdict + permute (every wave, round 6) | - | updates | sum")
    for w in range(16):
        v = a[w, :7] / rounds[w]
        print("%4d | " % w + " | ".join("%7.1f" % x for x in v) + " | %7.1f" % v.sum())
    v = (a[:, :7] / rounds[:, None]).mean(0)
    print("mean | " + " | ".join("%7.1f" % x for x in v) + " | %7.1f" % v.sum())


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "uniform")
