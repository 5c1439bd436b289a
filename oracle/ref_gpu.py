"""The REFERENCE'S OWN device kernels on the MI355X (test infrastructure; oracle/Makefile builds them from /root/reference in place
into oracle/_ref/, the .so files travel to the GPU box):

  iou3d_kernel_ref.so      boxesoverlapLauncher, boxesioubevLauncher, nmsLauncher, nmsNormalLauncher   (iou3d_kernel.cu:350-388)
  roipool3d_kernel_ref.so  roipool3dLauncher, roipool3dLauncher_slow                                    (roipool3d_kernel.cu:197-237)
  pointnet2_kernels_ref.so the nine *_kernel_launcher(_fast) functions of ball_query_gpu.cu, group_points_gpu.cu, sampling_gpu.cu,
                           interpolate_gpu.cu (K1-K9), launched on the NULL stream here

The launchers are C++ functions (mangled names) that launch on the NULL stream and do not synchronise; the wrappers below take torch
tensors on cuda:0, synchronise before and after, and add the two pieces of HOST code the reference keeps outside these files --
restated, not compiled: the greedy reduce over the suppression mask (iou3d.cpp:100-119) and the allocation / zero-fill conventions of
iou3d_utils.py / roipool3d_utils.py.  Never imported by the package: only tests/ use it."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}
_P, _I, _F = C.c_void_p, C.c_int, C.c_float


def available():
    return all(os.path.exists(os.path.join(_HERE, "_ref", f)) for f in ("iou3d_kernel_ref.so", "roipool3d_kernel_ref.so", "pointnet2_kernels_ref.so"))


def _load(name):
    if name not in _libs:
        import torch  # the HIP runtime of the process is torch's copy: map it first (see 3d_adapt_auto_driving_amd/_lib.py)
        hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip):
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        _libs[name] = C.CDLL(os.path.join(_HERE, "_ref", name))
    return _libs[name]


def _fn(lib, mangled, argtypes):
    f = getattr(_load(lib), mangled)
    f.argtypes, f.restype = argtypes, None
    return f


def _sync():
    import torch
    torch.cuda.synchronize()


def boxes_overlap_bev(a, b):
    """iou3d_utils.py:21-34 around boxesoverlapLauncher: (N,5), (M,5) -> (N,M) overlap areas"""
    import torch
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _sync()
    _fn("iou3d_kernel_ref.so", "_Z20boxesoverlapLauncheriPKfiS0_Pf", [_I, _P, _I, _P, _P])(a.shape[0], a.data_ptr(), b.shape[0], b.data_ptr(), out.data_ptr())
    _sync()
    return out


def boxes_iou_bev(a, b):
    import torch
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _sync()
    _fn("iou3d_kernel_ref.so", "_Z19boxesioubevLauncheriPKfiS0_Pf", [_I, _P, _I, _P, _P])(a.shape[0], a.data_ptr(), b.shape[0], b.data_ptr(), out.data_ptr())
    _sync()
    return out


def nms_mask(boxes, thresh, rotated=True):
    """the device half of nms_gpu / nms_normal_gpu (iou3d.cpp:73-99 / :123-149): the (n, ceil(n / 64)) u64 suppression mask"""
    import torch
    n = boxes.shape[0]
    cols = (n + 63) // 64
    mask = torch.zeros((n, cols), dtype=torch.int64, device=boxes.device)
    _sync()
    name = "_Z11nmsLauncherPKfPyif" if rotated else "_Z17nmsNormalLauncherPKfPyif"
    _fn("iou3d_kernel_ref.so", name, [_P, _P, _I, _F])(boxes.data_ptr(), mask.data_ptr(), n, float(thresh))
    _sync()
    return mask.cpu().numpy().view(np.uint64)


def nms_keep_from_mask(mask, n):
    """the host half, iou3d.cpp:100-119 restated: walk the rows in order, keep a row unless an earlier kept row suppresses it"""
    cols = mask.shape[1]
    remv = np.zeros(cols, dtype=np.uint64)
    keep = []
    for i in range(n):
        nb, ib = i // 64, i % 64
        if not (int(remv[nb]) >> ib) & 1:
            keep.append(i)
            remv |= mask[i]
    return np.array(keep, dtype=np.int64)


def nms(boxes, thresh, rotated=True):
    return nms_keep_from_mask(nms_mask(boxes, thresh, rotated), boxes.shape[0])


def roipool3d(xyz, boxes3d, pts_feature, sampled, slow=False):
    """roipool3d.cpp:48-79 / :15-46 around roipool3dLauncher(_slow): zero-filled outputs (roipool3d_utils.py:21-23), boxes as given
    (the caller enlarges them).  -> pooled (B,M,S,3+C), empty (B,M) i32"""
    import torch
    B, N, _ = xyz.shape
    M, Cf = boxes3d.shape[1], pts_feature.shape[2]
    pooled = torch.zeros((B, M, sampled, 3 + Cf), dtype=torch.float32, device=xyz.device)
    empty = torch.zeros((B, M), dtype=torch.int32, device=xyz.device)
    _sync()
    name = "_Z22roipool3dLauncher_slowiiiiiPKfS0_S0_PfPi" if slow else "_Z17roipool3dLauncheriiiiiPKfS0_S0_PfPi"
    _fn("roipool3d_kernel_ref.so", name, [_I] * 5 + [_P] * 5)(B, N, M, Cf, sampled, xyz.data_ptr(), boxes3d.data_ptr(), pts_feature.data_ptr(),
                                                             pooled.data_ptr(), empty.data_ptr())
    _sync()
    return pooled, empty


# ---- pointnet2 (K1-K9): outputs allocated as pointnet2_utils.py allocates them
_PN2 = "pointnet2_kernels_ref.so"


def ball_query(radius, nsample, xyz, new_xyz):
    """pointnet2_utils.py:203-221 around ball_query_kernel_launcher_fast: zero-filled idx (b,m,nsample) i32"""
    import torch
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz.device)
    _sync()
    _fn(_PN2, "_Z31ball_query_kernel_launcher_fastiiifiPKfS0_PiP12ihipStream_t", [_I, _I, _I, _F, _I, _P, _P, _P, _P])(
        b, n, m, float(radius), nsample, new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(), None)
    _sync()
    return idx


def furthest_point_sample(xyz, npoint):
    """pointnet2_utils.py:12-29 around furthest_point_sampling_kernel_launcher: temp filled with 1e10 -> idx (b,npoint) i32"""
    import torch
    b, n, _ = xyz.shape
    temp = torch.full((b, n), 1e10, dtype=torch.float32, device=xyz.device)
    idx = torch.empty((b, npoint), dtype=torch.int32, device=xyz.device)
    _sync()
    _fn(_PN2, "_Z39furthest_point_sampling_kernel_launcheriiiPKfPfPiP12ihipStream_t", [_I, _I, _I, _P, _P, _P, _P])(
        b, n, npoint, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr(), None)
    _sync()
    return idx, temp


def three_nn(unknown, known):
    """-> dist2 (b,n,3) f32 (squared, as the kernel writes them), idx (b,n,3) i32"""
    import torch
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknown.device)
    _sync()
    _fn(_PN2, "_Z29three_nn_kernel_launcher_fastiiiPKfS0_PfPiP12ihipStream_t", [_I, _I, _I, _P, _P, _P, _P, _P])(
        b, n, m, unknown.data_ptr(), known.data_ptr(), d2.data_ptr(), idx.data_ptr(), None)
    _sync()
    return d2, idx


def three_interpolate(points, idx, weight):
    import torch
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    _sync()
    _fn(_PN2, "_Z38three_interpolate_kernel_launcher_fastiiiiPKfPKiS0_PfP12ihipStream_t", [_I, _I, _I, _I, _P, _P, _P, _P, _P])(
        b, c, m, n, points.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr(), None)
    _sync()
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    import torch
    b, c, n = grad_out.shape
    gp = torch.zeros((b, c, m), dtype=torch.float32, device=grad_out.device)
    _sync()
    _fn(_PN2, "_Z43three_interpolate_grad_kernel_launcher_fastiiiiPKfPKiS0_PfP12ihipStream_t", [_I, _I, _I, _I, _P, _P, _P, _P, _P])(
        b, c, n, m, grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), gp.data_ptr(), None)
    _sync()
    return gp


def group_points(points, idx):
    import torch
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    _sync()
    _fn(_PN2, "_Z33group_points_kernel_launcher_fastiiiiiPKfPKiPfP12ihipStream_t", [_I] * 5 + [_P] * 4)(
        b, c, n, npoints, nsample, points.data_ptr(), idx.data_ptr(), out.data_ptr(), None)
    _sync()
    return out


def group_points_grad(grad_out, idx, n):
    import torch
    b, c, npoints, nsample = grad_out.shape
    gp = torch.zeros((b, c, n), dtype=torch.float32, device=grad_out.device)
    _sync()
    _fn(_PN2, "_Z38group_points_grad_kernel_launcher_fastiiiiiPKfPKiPfP12ihipStream_t", [_I] * 5 + [_P] * 4)(
        b, c, n, npoints, nsample, grad_out.data_ptr(), idx.data_ptr(), gp.data_ptr(), None)
    _sync()
    return gp


def gather_points(points, idx):
    import torch
    b, c, n = points.shape
    npoints = idx.shape[1]
    out = torch.empty((b, c, npoints), dtype=torch.float32, device=points.device)
    _sync()
    _fn(_PN2, "_Z34gather_points_kernel_launcher_fastiiiiPKfPKiPfP12ihipStream_t", [_I] * 4 + [_P] * 4)(
        b, c, n, npoints, points.data_ptr(), idx.data_ptr(), out.data_ptr(), None)
    _sync()
    return out


def gather_points_grad(grad_out, idx, n):
    import torch
    b, c, npoints = grad_out.shape
    gp = torch.zeros((b, c, n), dtype=torch.float32, device=grad_out.device)
    _sync()
    _fn(_PN2, "_Z39gather_points_grad_kernel_launcher_fastiiiiPKfPKiPfP12ihipStream_t", [_I] * 4 + [_P] * 4)(
        b, c, n, npoints, grad_out.data_ptr(), idx.data_ptr(), gp.data_ptr(), None)
    _sync()
    return gp
