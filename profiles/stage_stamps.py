"""Poor man's thread trace: s_memtime stamps at the stage boundaries of a kernel, written per wave to a __device__ array.

The image has rocprofv3's --att front end but not its decoder library, so a real thread trace cannot be read back; this is the
substitute that answered "where does the time between the MFMAs go".  Two steps:

  python profiles/stage_stamps.py build          (in the build container: hipcc)
      text-instruments COPIES of csrc/rpn_tail.hip and csrc/packed_layer.hip (a STAMP(k) after every barrier / panel stage /
      epilogue; the product sources are not touched), compiles them and links profiles/_exp/libprcnn_hip_stamps.so from the
      product's other objects (profiles/_exp/ is git-ignored but travels with gpurun)
  python profiles/stage_stamps.py tail [grid]    (on the GPU box)   per-stage cycles of rpn_tail_kernel, B = 8 shape
  python profiles/stage_stamps.py layer R K N    (on the GPU box)   per-stage cycles of packed_layer_pipe_kernel
  python profiles/stage_stamps.py packed [mean]  (on the GPU box)   per-stage cycles of sa_packed_mlp128_kernel, RCNN SA1 shape

`grid` (tail): number of persistent workgroups (256 = one per CU, 512 = the product's two per CU).  The read-out steps load
the instrumented library INSTEAD of lib/libprcnn_hip.so (they point _lib.LIB_PATH at it), nothing else changes."""
import ctypes, importlib, os, re, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d_adapt_auto_driving_amd", "csrc")
EXP = os.path.join(ROOT, "profiles", "_exp")
LIB = os.path.join(EXP, "libprcnn_hip_stamps.so")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()


def _abs_includes(s):
    return (s.replace('#include "common.hpp"', '#include <cstdlib>\n#include "%s/common.hpp"' % CSRC)
             .replace('#include "segmax.hpp"', '#include "%s/segmax.hpp"' % CSRC)
             .replace('#include "mfma_stream.hpp"', '#include "%s/mfma_stream.hpp"' % CSRC)
             .replace('#include "../../include/prcnn_hip.h"', '#include "%s/include/prcnn_hip.h"' % ROOT))


def instrument_tail(lin=False):
    """lin: the product form (rpn_tail_lin_kernel, the second tile loop of the file) instead of rpn_tail_kernel"""
    s = _abs_includes(open(os.path.join(CSRC, "rpn_tail.hip")).read())
    s = s.replace("struct RpnTailArgs {", "__device__ unsigned long long g_trace[512 * 4 * 32];\n"
                  "#define STAMP(k) if (lane == 0 && served == 2) g_trace[(blockIdx.x * 4 + w) * 32 + (k)] = __builtin_amdgcn_s_memtime();\n"
                  "struct RpnTailArgs {")
    lo = s.index("    for (unsigned int served = 0; t < tiles; ++served) {")
    hi = s.index("        tp = t;")
    if lin:
        lo = s.index("    for (unsigned int served = 0; t < tiles; ++served) {", lo + 1)
        hi = s.index("        tp = t;", hi + 1)
    lines = s[lo:hi].split("\n")
    out, names, k = [lines[0], "        STAMP(0)"], ["start"], 1
    for ln in lines[1:]:
        out.append(ln)
        st = ln.strip()
        if st.startswith(("lds_barrier();", "RT_STAGE(", "RT_STAGE_HOOK(", "RT_EPILOGUE(")):
            out.append("        STAMP(%d)" % k); names.append(st.split(";")[0][:44]); k += 1
    s = s[:lo] + "\n".join(out) + s[hi:]
    # (the launch size: PRCNN_MFMA_GRID, capped at one workgroup per CU since the kernel keeps four LDS tiles)
    s += ('\nextern "C" int prcnn_debug_trace(unsigned long long *dst)\n{\n    return (int)hipMemcpyFromSymbol(dst, '
          'HIP_SYMBOL(prcnn::g_trace), sizeof(unsigned long long) * 512 * 4 * 32);\n}\n')
    return s, names


def instrument_layer():
    s = _abs_includes(open(os.path.join(CSRC, "packed_layer.hip")).read())
    s = s.replace("constexpr int PL_ROWS = 64;", "__device__ unsigned long long g_pl_trace[1024 * 4 * 64];\n"
                  "#define STAMP(k) if (!SEGMAX && lane == 0 && (blockIdx.x + gridDim.x * blockIdx.y) < 1024) "
                  "g_pl_trace[((blockIdx.x + gridDim.x * blockIdx.y) * 4 + w) * 64 + (k)] = __builtin_amdgcn_s_memtime();\n"
                  "constexpr int PL_ROWS = 64;")
    # (first occurrence only: the 32-row kernel below repeats these lines)
    s = s.replace("    float wa[64], wb[64];\n    {\n        PL_LOAD_W(wa, 0)", "    float wa[64], wb[64];\n    STAMP(0)\n    {\n        PL_LOAD_W(wa, 0)", 1)
    s = s.replace("    const int np = K >> 7;\n    for (int p = 0; p < np; ++p) {\n        PL_VM_DRAIN",
                  "    const int np = K >> 7;\n    int sk = 1;\n    for (int p = 0; p < np; ++p) {\n        STAMP(sk++)\n        PL_VM_DRAIN", 1)
    s = s.replace("        lds_barrier();                                     // ... and published; the other tile is free\n",
                  "        lds_barrier();                                     // ... and published; the other tile is free\n        STAMP(sk++)\n")
    s = s.replace("            PL_STAGE_PREFETCH(T, TN, wa, wb, (p + 1) * 128)\n", "            PL_STAGE_PREFETCH(T, TN, wa, wb, (p + 1) * 128)\n            STAMP(sk++)\n")
    s = s.replace("            PL_STAGE(T, wa)\n        }\n    }\n", "            PL_STAGE(T, wa)\n            STAMP(sk++)\n        }\n    }\n    STAMP(62)\n")
    tail = "    pl_epilogue<SEGMAX>(acc0, acc1, tiles, ctr, t, rows, n0, bias, do_relu, out, ldo, rowinfo, tilecloud, m, out_col, n_store);\n}"
    assert tail in s
    s = s.replace(tail, tail[:-1] + "    STAMP(63)\n}")
    s += ('\nextern "C" int prcnn_debug_pl_trace(unsigned long long *dst)\n{\n    return (int)hipMemcpyFromSymbol(dst, '
          'HIP_SYMBOL(prcnn::g_pl_trace), sizeof(unsigned long long) * 1024 * 4 * 64);\n}\n')
    assert s.count("STAMP(") >= 8
    return s


def instrument_packed():
    s = _abs_includes(open(os.path.join(CSRC, "sa_packed.hip")).read())
    a = s.index("void sa_packed_mlp128_kernel(")
    b = s.index("// ------------------------------------------------------------------------------------------------ C3 = 256")
    k = s[a:b]
    def put(old, new):
        nonlocal k
        assert old in k, old[:60]
        k = k.replace(old, new, 1)
    put("    for (int served = 0; served < tiles_per_wg && t < tiles; ++served) {\n", "    for (int served = 0; served < tiles_per_wg && t < tiles; ++served) {\n        STAMP(0)\n")
    put("        __syncthreads();\n        const long t_next = slot[(served + 1) & 1];", "        STAMP(1)\n        __syncthreads();\n        STAMP(2)\n        const long t_next = slot[(served + 1) & 1];")
    put("#pragma unroll\n            for (int r = 0; r < 16; ++r) {\n                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;\n                Y1[row * PK_LD + 32 * w + j] = fmaxf(acc0[r] + bias2, 0.f);",
        "            STAMP(3)\n#pragma unroll\n            for (int r = 0; r < 16; ++r) {\n                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;\n                Y1[row * PK_LD + 32 * w + j] = fmaxf(acc0[r] + bias2, 0.f);")
    put("        if (tid < PK_ROWS && t_next < tiles) dxyz_s[(served + 1) & 1][tid] = dnext;   // ordered before", "        STAMP(4)\n        if (tid < PK_ROWS && t_next < tiles) dxyz_s[(served + 1) & 1][tid] = dnext;   // ordered before")
    put("   // in flight during layer 3\n        }\n        __syncthreads();\n", "   // in flight during layer 3\n        }\n        __syncthreads();\n        STAMP(5)\n")
    put("            const int myc = cc[lane], prevc = cc[lane ? lane - 1 : 0];\n            const unsigned long long start = __ballot(lane == 0 || myc != prevc);\n            pk_segmented_max(acc0, acc1, cc, start, h, out, out_stride, out_col + 32 * w + j, bias3);\n        }",
        "            STAMP(6)\n            const int myc = cc[lane], prevc = cc[lane ? lane - 1 : 0];\n            const unsigned long long start = __ballot(lane == 0 || myc != prevc);\n            pk_segmented_max(acc0, acc1, cc, start, h, out, out_stride, out_col + 32 * w + j, bias3);\n            STAMP(7)\n        }")
    s = s[:a] + k + s[b:]
    s = s.replace("constexpr int PK_ROWS = 64;", "__device__ unsigned long long g_sp_trace[1024 * 4 * 8];\n"
                  "#define STAMP(k) if ((threadIdx.x & 63) == 0 && served == 1 && blockIdx.x < 1024) g_sp_trace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime();\n"
                  "constexpr int PK_ROWS = 64;", 1)
    s += ('\nextern "C" int prcnn_debug_sp_trace(unsigned long long *dst)\n{\n    return (int)hipMemcpyFromSymbol(dst, '
          'HIP_SYMBOL(prcnn::g_sp_trace), sizeof(unsigned long long) * 1024 * 4 * 8);\n}\n')
    return s


def build():
    os.makedirs(EXP, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC])
    tail, names = instrument_tail(os.environ.get("STAMP_TAIL_LIN") == "1")       # STAMP_TAIL_LIN=1: stamps in rpn_tail_lin_kernel
    open(os.path.join(EXP, "trace_names.txt"), "w").write("\n".join(names))
    objs, done = [], []
    sources = [("rpn_tail", tail)]
    for name, fn in (("packed_layer", instrument_layer), ("sa_packed", instrument_packed)):
        try:
            sources.append((name, fn()))
        except AssertionError:
            print("(%s: the source no longer has the round-2 anchors of this tool -- linked uninstrumented)" % name)
    for name, src in sources:
        done.append(name + ".o")
        path = os.path.join(EXP, name + "_stamps.hip")
        open(path, "w").write(src)
        obj = os.path.join(EXP, name + "_stamps.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", path, "-o", obj])
        objs.append(obj)
    others = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
              if f.endswith(".o") and f not in done]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + others + objs)
    print("built", LIB)


def _load():
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("3d_adapt_auto_driving_amd")
    lib = importlib.import_module("3d_adapt_auto_driving_amd._lib")
    lib.LIB_PATH = LIB                                  # before the first load(): this process binds the instrumented build
    sys.path.insert(0, pkg.DROPIN_DIR)
    import pointnet2_cuda as X
    return X, lib.load()


def _row(label, v):
    print("| %-44s | %7.0f | %7.0f | %7.0f |" % (label, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))


def read_tail(grid):
    os.environ["PRCNN_MFMA_GRID"] = str(grid)
    import torch
    X, L = _load()
    dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
    b, n, m, n_reg = 8, 16384, 4096, 76
    known = torch.randn((b, m, 256), device=dev, generator=g)
    base = (torch.arange(n, device=dev) * m // n).view(1, n, 1)
    idx = ((base + torch.randint(0, 8, (b, n, 3), device=dev, generator=g)) % m).to(torch.int32).contiguous()
    w = torch.rand((b, n, 3), device=dev, generator=g) + 0.05; w = (w / w.sum(2, keepdim=True)).contiguous()
    wcat = (torch.randn((768, 128), device=dev, generator=g) / 11).contiguous(); bcat = torch.randn((5, 128), device=dev, generator=g) * 0.1
    wc2 = torch.randn(128, device=dev, generator=g) / 11; bc2 = torch.randn(1, device=dev, generator=g)
    feats = torch.empty((b, n, 128), device=dev); cls = torch.empty((b, n, 1), device=dev); reg = torch.empty((b, n, n_reg), device=dev)
    if os.environ.get("STAMP_TAIL_LIN") == "1":
        G = torch.randn((b, m, 128), device=dev, generator=g)
        X.rpn_tail_lin_wrapper(G, idx, w, wcat[256:].contiguous(), bcat, wc2, bc2, feats, cls, reg); torch.cuda.synchronize()
    else:
        X.rpn_tail_wrapper(known, idx, w, wcat, bcat, wc2, bc2, feats, cls, reg); torch.cuda.synchronize()
    names = open(os.path.join(EXP, "trace_names.txt")).read().split("\n")
    K = len(names)
    buf = np.zeros(512 * 4 * 32, np.uint64)
    L.prcnn_debug_trace.argtypes = [ctypes.c_void_p]; assert L.prcnn_debug_trace(buf.ctypes.data) == 0
    tr = buf.reshape(512, 4, 32).astype(np.int64)[:, :, :K]
    ok = tr[:, :, 0] > 0
    d = np.diff(tr, axis=2)
    print("rpn_tail_kernel, %d persistent workgroups, third tile of every workgroup (%d of them), s_memtime cycles per wave\n" % (grid, int(ok[:, 0].sum())))
    print("| up to | median | p10 | p90 |\n|---|---|---|---|")
    for k in range(K - 1):
        _row(names[k + 1], d[:, :, k][ok])
    _row("tile total", (tr[:, :, K - 1] - tr[:, :, 0])[ok])


def read_layer(rows, K, N):
    import torch
    X, L = _load()
    dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn((rows, K), device=dev, generator=g); w = torch.randn((K, N), device=dev, generator=g) / 16
    bias = torch.randn(N, device=dev, generator=g); out = torch.empty((rows, N), device=dev)
    for _ in range(3): X.packed_layer_wrapper(a, w, bias, True, out)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 4 * 64, np.uint64)
    L.prcnn_debug_pl_trace.argtypes = [ctypes.c_void_p]; assert L.prcnn_debug_pl_trace(buf.ctypes.data) == 0
    nwg = min(1024, ((rows + 63) // 64) * (N // 128)); npan = K // 128
    tr = buf.reshape(1024, 4, 64).astype(np.int64)[:nwg]
    labels = ["prologue (panel 0: weights, rows -> LDS)"]
    for p in range(npan):
        labels += ["panel %d drain + barrier" % p, "panel %d stage (128 MFMAs + next panel)" % p, "panel %d weight copy / loop" % p]
    labels = labels[:-1]
    d = np.diff(tr[:, :, :1 + len(labels)], axis=2)
    print("packed_layer_pipe_kernel %d x %d -> %d: %d workgroups, s_memtime cycles per wave\n" % (rows, K, N, nwg))
    print("| span | median | p10 | p90 |\n|---|---|---|---|")
    for i, lb in enumerate(labels):
        _row(lb, d[:, :, i].ravel())
    _row("epilogue", (tr[:, :, 63] - tr[:, :, 62]).ravel())
    _row("workgroup total", (tr[:, :, 63] - tr[:, :, 0]).ravel())


def read_packed(mean):
    import torch
    X, L = _load()
    dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
    b, n, m, ns = 800, 512, 128, 64
    xyz = torch.randn((b, n, 3), device=dev, generator=g); new_xyz = xyz[:, :m].contiguous()
    P = torch.randn((b, n, 128), device=dev, generator=g); wx = torch.randn((3, 128), device=dev, generator=g)
    w2 = torch.randn((128, 128), device=dev, generator=g) / 11; w3 = torch.randn((128, 128), device=dev, generator=g) / 11
    b2 = torch.randn(128, device=dev, generator=g); b3 = torch.randn(128, device=dev, generator=g)
    out = torch.empty((b, m, 128), device=dev)
    full = torch.sort(torch.argsort(torch.rand((b, m, n), device=dev, generator=g), dim=2)[:, :, :ns], dim=2).values.to(torch.int32)
    cnt = torch.clamp((torch.rand((b, m, 1), device=dev, generator=g) * 2 * mean).long() + 1, max=ns)
    idx = torch.where(torch.arange(ns, device=dev).view(1, 1, ns) < cnt, full, full[:, :, :1]).contiguous()
    pk = X.ball_pack_wrapper(idx, xyz, new_xyz)
    for _ in range(3): X.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, out, 0)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 4 * 8, np.uint64)
    L.prcnn_debug_sp_trace.argtypes = [ctypes.c_void_p]; assert L.prcnn_debug_sp_trace(buf.ctypes.data) == 0
    tr = buf.reshape(1024, 4, 8).astype(np.int64); ok = tr[:, :, 0] > 0
    d = np.diff(tr, axis=2)
    print("sa_packed_mlp128_kernel, 800 clouds x 128 centres, ~%.1f distinct rows per centre: %d tiles; second tile of every workgroup, s_memtime cycles per wave\n" % (mean + 1, int(pk.hdr[0])))
    print("| span | median | p10 | p90 |\n|---|---|---|---|")
    for k, nm in enumerate(["builder (P rows + affine -> LDS)", "barrier", "next tile's row list + layer 2 MFMAs", "epilogue -> Y1", "prefetch of the next P rows + barrier",
                            "layer 3 MFMAs", "segmented max + atomics"]):
        _row(nm, d[:, :, k][ok])
    _row("tile total", (tr[:, :, 7] - tr[:, :, 0])[ok])


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "build"
    if cmd == "build":
        build()
    elif cmd == "tail":
        read_tail(int(sys.argv[2]) if len(sys.argv) > 2 else 512)
    elif cmd == "layer":
        read_layer(*[int(v) for v in sys.argv[2:5]])
    elif cmd == "packed":
        read_packed(float(sys.argv[2]) if len(sys.argv) > 2 else 0.6)
    else:
        raise SystemExit(__doc__)
