"""ctypes/numpy front-end of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the shipped package (3d_adapt_auto_driving_amd/) never
imports this module.  Every function takes/returns C-contiguous numpy arrays
(float32 / int32 / int64) and mirrors one entry of oracle/prcnn_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None
_variant_libs = {}

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)
_l = C.POINTER(C.c_longlong)


def build(force=False):
    """(Re)build liboracle.so with gcc; also builds oracle/_ref when /root/reference exists."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("prcnn_oracle.c", "mlp_oracle.c", "prcnn_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _declare(h):
    h.orc_opt_n_threads.restype = C.c_int
    h.orc_nms.restype = C.c_int
    h.orc_nms_normal.restype = C.c_int
    h.orc_num_threads.restype = C.c_int
    h.orc_variant.restype = C.c_int
    return h


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _declare(C.CDLL(_LIB_PATH))
        assert _lib.orc_variant() == 0
    return _lib


def variant(name):
    """Context manager: every front-end function of this module runs on an FMA-CONTRACTED build of the same
    source ('fma' = liboracle_fma.so, 'fma2' = liboracle_fma2.so; oracle/Makefile) instead of the contract-off
    oracle.  Only for counting how many decisions move under contraction -- never the parity reference."""
    import contextlib

    @contextlib.contextmanager
    def _cm():
        global _lib
        if name not in _variant_libs:
            path = os.path.join(_HERE, "liboracle_%s.so" % name)
            src = os.path.join(_HERE, "prcnn_oracle.c")
            if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", _HERE, os.path.basename(path)], stdout=subprocess.DEVNULL)
            _variant_libs[name] = _declare(C.CDLL(path))
            assert _variant_libs[name].orc_variant() == {"fma": 1, "fma2": 2}[name]
        saved = lib()
        _lib = _variant_libs[name]
        try:
            yield _lib
        finally:
            _lib = saved
    return _cm()


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(t):
    lib().orc_set_num_threads(int(t))


def opt_n_threads(n):
    return lib().orc_opt_n_threads(int(n))


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)  # caller zero-fills (pointnet2_utils.py:218)
    lib().orc_ball_query(b, n, m, C.c_float(radius), nsample, _p(new_xyz, _f), _p(xyz, _f), _p(idx, _i))
    return idx


def ball_query_into(radius, nsample, xyz, new_xyz, idx):
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    lib().orc_ball_query(b, n, m, C.c_float(radius), nsample, _p(new_xyz, _f), _p(xyz, _f), _p(idx, _i))


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), np.float32)
    lib().orc_group_points(b, c, n, m, ns, _p(points, _f), _p(idx, _i), _p(out, _f))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m, ns = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().orc_group_points_grad(b, c, n, m, ns, _p(grad_out, _f), _p(idx, _i), _p(g, _f))
    return g


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), np.float32)
    lib().orc_gather_points(b, c, n, m, _p(points, _f), _p(idx, _i), _p(out, _f))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().orc_gather_points_grad(b, c, n, m, _p(grad_out, _f), _p(idx, _i), _p(g, _f))
    return g


def furthest_point_sample(xyz, npoint, block_size=None, return_temp=False, hipcc_arithmetic=False):
    """hipcc_arithmetic: the distance as the reference's kernel binary computes it when hipcc builds sampling_gpu.cu for gfx950
    ((fma(dy, dy, dx*dx)) + dz*dz, mlp_oracle.c) instead of the source's one-rounding-per-operation contract"""
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)  # pointnet2_utils.py:26
    idx = np.empty((b, npoint), np.int32)
    if hipcc_arithmetic:
        bs = int(block_size) if block_size is not None else int(lib().orc_opt_n_threads(n))
        lib().orc_furthest_point_sampling_hipcc_bs(b, n, npoint, bs, _p(xyz, _f), _p(temp, _f), _p(idx, _i))
    elif block_size is None:
        lib().orc_furthest_point_sampling(b, n, npoint, _p(xyz, _f), _p(temp, _f), _p(idx, _i))
    else:
        lib().orc_furthest_point_sampling_bs(b, n, npoint, int(block_size), _p(xyz, _f), _p(temp, _f), _p(idx, _i))
    return (idx, temp) if return_temp else idx


def three_nn(unknown, known):
    """Returns (dist2, idx): squared distances as the kernel writes them (sqrt is the caller's)."""
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().orc_three_nn(b, n, m, _p(unknown, _f), _p(known, _f), _p(d2, _f), _p(idx, _i))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), np.float32)
    lib().orc_three_interpolate(b, c, m, n, _p(points, _f), _p(idx, _i), _p(weight, _f), _p(out, _f))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    b, c, n = grad_out.shape
    g = np.zeros((b, c, m), np.float32)
    lib().orc_three_interpolate_grad(b, c, n, m, _p(grad_out, _f), _p(idx, _i), _p(weight, _f), _p(g, _f))
    return g


def query_and_group(radius, nsample, xyz, new_xyz, features=None):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    if features is not None:
        features = _f32(features)
        c = features.shape[1]
        fp = _p(features, _f)
    else:
        c, fp = 0, None
    idx = np.empty((b, m, nsample), np.int32)
    out = np.empty((b, 3 + c, m, nsample), np.float32)
    lib().orc_query_and_group(b, n, m, c, C.c_float(radius), nsample, _p(new_xyz, _f), _p(xyz, _f), fp,
                              _p(idx, _i), _p(out, _f))
    return out, idx


def boxes_overlap_bev(boxes_a, boxes_b):
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_overlap_bev(a.shape[0], _p(a, _f), b.shape[0], _p(b, _f), _p(out, _f))
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_iou_bev(a.shape[0], _p(a, _f), b.shape[0], _p(b, _f), _p(out, _f))
    return out


def nms(boxes, thresh):
    """Rotated NMS on score-sorted (n,5) BEV boxes; returns kept row indices (int64)."""
    boxes = _f32(boxes)
    keep = np.zeros((boxes.shape[0],), np.int64)
    k = lib().orc_nms(boxes.shape[0], _p(boxes, _f), _p(keep, _l), C.c_float(thresh))
    return keep[:k]


def nms_normal(boxes, thresh):
    boxes = _f32(boxes)
    keep = np.zeros((boxes.shape[0],), np.int64)
    k = lib().orc_nms_normal(boxes.shape[0], _p(boxes, _f), _p(keep, _l), C.c_float(thresh))
    return keep[:k]


def roipool3d(xyz, boxes3d, pts_feature, sampled_pts_num):
    """Batched GPU-semantics pooling on ALREADY ENLARGED boxes (roipool3d_utils.py:19-26)."""
    xyz, boxes3d, pts_feature = _f32(xyz), _f32(boxes3d), _f32(pts_feature)
    bsz, n, _ = xyz.shape
    m = boxes3d.shape[1]
    cf = pts_feature.shape[2]
    pooled = np.zeros((bsz, m, sampled_pts_num, 3 + cf), np.float32)
    empty = np.zeros((bsz, m), np.int32)
    lib().orc_roipool3d(bsz, n, m, cf, sampled_pts_num, _p(xyz, _f), _p(boxes3d, _f), _p(pts_feature, _f),
                        _p(pooled, _f), _p(empty, _i))
    return pooled, empty


def pts_in_boxes3d(pts, boxes3d):
    pts, boxes3d = _f32(pts), _f32(boxes3d)
    flag = np.zeros((boxes3d.shape[0], pts.shape[0]), np.int64)
    lib().orc_pts_in_boxes3d(boxes3d.shape[0], pts.shape[0], _p(pts, _f), _p(boxes3d, _f), _p(flag, _l))
    return flag


def roipool3d_cpu(pts, boxes3d, pts_feature, sampled_pts_num):
    pts, boxes3d, pts_feature = _f32(pts), _f32(boxes3d), _f32(pts_feature)
    m, n, cf = boxes3d.shape[0], pts.shape[0], pts_feature.shape[1]
    pp = np.zeros((m, sampled_pts_num, 3), np.float32)
    pf = np.zeros((m, sampled_pts_num, cf), np.float32)
    pe = np.zeros((m,), np.int64)
    lib().orc_roipool3d_cpu(m, n, cf, sampled_pts_num, _p(pts, _f), _p(boxes3d, _f), _p(pts_feature, _f),
                            _p(pp, _f), _p(pf, _f), _p(pe, _l))
    return pp, pf, pe


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    boxes, query_boxes = _f32(boxes), _f32(query_boxes)
    n, k = boxes.shape[0], query_boxes.shape[0]
    iou = np.zeros((n, k), np.float32)
    if n and k:
        lib().orc_rotate_iou_eval(n, k, _p(boxes, _f), _p(query_boxes, _f), _p(iou, _f), int(criterion))
    return iou


# --------------------------------------------------------------------------- oracle/_ref

def load_reference_roipool():
    """Import oracle/_ref/roipool3d_ref.so = the reference's own roipool3d.cpp compiled by
    oracle/Makefile.  Returns the module or None when it was never built (e.g. no /root/reference).
    Its CUDA entry points reference launchers that do not exist here; they stay unresolved
    (RTLD_LAZY) and must not be called."""
    path = os.path.join(_HERE, "_ref", "roipool3d_ref.so")
    if not os.path.exists(path):
        return None
    import importlib.util
    import sys
    import torch  # noqa: F401  (libtorch symbols)
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location("roipool3d_ref", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod
