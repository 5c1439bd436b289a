"""Drop-in for the reference's ``roipool3d_cuda`` pybind module
(pointrcnn/lib/utils/roipool3d/src/roipool3d.cpp:198-203) over libprcnn_hip.so.

``forward`` / ``forward_slow`` are the device path (both map to the same kernel: the "slow"
overload, roipool3d_kernel.cu:31-94, computes the identical result).  ``pts_in_boxes3d_cpu`` and
``roipool3d_cpu`` are the module's HOST utilities (CPU tensors, unbatched) used by the reference's
dataset / GT-database code (kitti_rcnn_dataset.py:507, generate_gt_database.py:75): host functions of
the same library (csrc/roipool_host.hip), pinned to the outputs of the reference's own compiled
roipool3d.cpp.  They are not a fallback: the device entry points refuse CPU tensors and these refuse CUDA ones.
"""
import importlib
import os
import sys

import torch

_pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg_dir) not in sys.path:
    sys.path.insert(0, os.path.dirname(_pkg_dir))
_lib = importlib.import_module(os.path.basename(_pkg_dir) + "._lib")
IS_HIP_EXTENSION = True     # marks the real extension (the test suite's CPU stand-ins do not carry it)


def _chk(dtype, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("roipool3d_cuda: tensor must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError("roipool3d_cuda: tensor must be contiguous")
        if t.dtype != dtype:
            raise RuntimeError("roipool3d_cuda: expected %s, got %s" % (dtype, t.dtype))


def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    _chk(torch.float32, xyz, boxes3d, pts_feature, pooled_features)
    _chk(torch.int32, pooled_empty_flag)
    _lib.call("prcnn_roipool3d", xyz.size(0), xyz.size(1), boxes3d.size(1), pts_feature.size(2),
              pooled_features.size(2), xyz.data_ptr(), boxes3d.data_ptr(), pts_feature.data_ptr(),
              pooled_features.data_ptr(), pooled_empty_flag.data_ptr(), _lib.current_stream(xyz))
    return 1


forward_slow = forward


def point_groups(xyz):
    """xyz (B,N,3), N % 64 == 0, N <= 65536 -> (pxyz (B,N,4), aabb (B,N/64,2,4)): the clouds in Morton order with the original
    index in lane 3, and the bounding box of every 64-point group (csrc/fps.hip prcnn_point_groups) -- forward_canonical culls
    by group when handed these."""
    _chk(torch.float32, xyz)
    B, N, _ = xyz.shape
    pxyz = torch.empty((B, N, 4), dtype=torch.float32, device=xyz.device)
    aabb = torch.empty((B, N // 64, 2, 4), dtype=torch.float32, device=xyz.device)
    _lib.call("prcnn_point_groups", B, N, xyz.data_ptr(), pxyz.data_ptr(), aabb.data_ptr(), _lib.current_stream(xyz))
    return pxyz, aabb


def forward_canonical(xyz, rois, feats, seg_mask, depth, pool_extra_width, pooled, pooled_empty_flag, pooled_cnt=None, groups=None, xyz_out=None):
    """Extension beyond the reference ABI (csrc/roipool.hip roipool3d_canonical_kernel): enlarge + pool + canonical
    transform + RCNN row layout [x',y',z',mask,depth,0,0,0 | C feats] in one pass.  pooled (B,M,S,8+C).
    pooled_cnt (B,M) i32, optional: distinct rows per box (rows beyond are wrap-around copies); when given, feature columns
    are written for rows < round_up(cnt, 64) only.  groups = point_groups(xyz), optional: same result, found by culling.
    xyz_out (B,M,S,3), optional: receives pooled[..., 0:3] as dense clouds in the same pass."""
    _chk(torch.float32, xyz, rois, feats, seg_mask, depth, pooled)
    if groups is not None:
        _chk(torch.float32, *groups)
    _chk(torch.int32, pooled_empty_flag)
    if pooled_cnt is not None:
        _chk(torch.int32, pooled_cnt)
    if xyz_out is not None:
        _chk(torch.float32, xyz_out)
        if tuple(xyz_out.shape) != (xyz.size(0), rois.size(1), pooled.size(2), 3):
            raise RuntimeError("roipool3d_cuda.forward_canonical: xyz_out must be (B, M, S, 3)")
    _lib.call("prcnn_roipool3d_canonical_xyz", xyz.size(0), xyz.size(1), rois.size(1), feats.size(2), pooled.size(2),
              float(pool_extra_width), xyz.data_ptr(), rois.data_ptr(), feats.data_ptr(), seg_mask.data_ptr(),
              depth.data_ptr(), pooled.data_ptr(), pooled_empty_flag.data_ptr(), _lib.ptr(pooled_cnt),
              None if groups is None else groups[0].data_ptr(), None if groups is None else groups[1].data_ptr(),
              _lib.ptr(xyz_out), _lib.current_stream(xyz))
    return 1


def _chk_host(dtype, *tensors):
    for t in tensors:
        if t.is_cuda:
            raise RuntimeError("roipool3d_cuda: host utility called with a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError("roipool3d_cuda: tensor must be contiguous")
        if t.dtype != dtype:
            raise RuntimeError("roipool3d_cuda: expected %s, got %s" % (dtype, t.dtype))


def pts_in_boxes3d_cpu(pts_flag, pts, boxes3d):
    """Host utility of the reference module (roipool3d.cpp:97-125): pts_flag (M,N) int64 <- point j inside box i."""
    _chk_host(torch.float32, pts, boxes3d)
    _chk_host(torch.int64, pts_flag)
    _lib.call("prcnn_host_pts_in_boxes3d", boxes3d.size(0), pts.size(0), pts.data_ptr(), boxes3d.data_ptr(), pts_flag.data_ptr())
    return 1


def roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag):
    """Host utility of the reference module (roipool3d.cpp:127-195), unbatched, CPU tensors."""
    _chk_host(torch.float32, pts, boxes3d, pts_feature, pooled_pts, pooled_features)
    _chk_host(torch.int64, pooled_empty_flag)
    _lib.call("prcnn_host_roipool3d", boxes3d.size(0), pts.size(0), pts_feature.size(1), pooled_pts.size(1), pts.data_ptr(),
              boxes3d.data_ptr(), pts_feature.data_ptr(), pooled_pts.data_ptr(), pooled_features.data_ptr(),
              pooled_empty_flag.data_ptr())
    return 1
