"""Drop-in for the reference's ``iou3d_cuda`` pybind module
(pointrcnn/lib/utils/iou3d/src/iou3d.cpp:174-179) over libprcnn_hip.so.

boxes are (n,5) [x1,y1,x2,y2,ry] CUDA f32 contiguous (CHECK_INPUT, iou3d.cpp:7-9); ``keep`` of
the two NMS calls is a CPU int64 tensor and the call blocks, as in the reference.
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


def _chk(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("iou3d_cuda: tensor must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError("iou3d_cuda: tensor must be contiguous")
        if t.dtype != torch.float32:
            raise RuntimeError("iou3d_cuda: expected float32")


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _chk(boxes_a, boxes_b, ans_overlap)
    _lib.call("prcnn_boxes_overlap_bev", boxes_a.size(0), boxes_a.data_ptr(), boxes_b.size(0),
              boxes_b.data_ptr(), ans_overlap.data_ptr(), _lib.current_stream(boxes_a))
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    _chk(boxes_a, boxes_b, ans_iou)
    _lib.call("prcnn_boxes_iou_bev", boxes_a.size(0), boxes_a.data_ptr(), boxes_b.size(0),
              boxes_b.data_ptr(), ans_iou.data_ptr(), _lib.current_stream(boxes_a))
    return 1


def _nms(name, boxes, keep, thresh):
    _chk(boxes)
    if keep.is_cuda or keep.dtype != torch.int64 or not keep.is_contiguous():
        raise RuntimeError("iou3d_cuda: keep must be a contiguous CPU int64 tensor")
    if keep.numel() < boxes.size(0):
        raise RuntimeError("iou3d_cuda: keep is shorter than boxes")
    return _lib.call(name, boxes.size(0), boxes.data_ptr(), keep.data_ptr(), float(thresh),
                     _lib.current_stream(boxes))


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms("prcnn_nms", boxes, keep, nms_overlap_thresh)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms("prcnn_nms_normal", boxes, keep, nms_overlap_thresh)


# -- extension beyond the reference module: batched, device-resident greedy NMS -------------
def nms_device(boxes, counts, thresh, rotated, max_keep, keep, num_keep):
    """boxes (P, n_max, 5) f32, counts (P) i32 or None, keep (P, max_keep) i32, num_keep (P) i32."""
    _chk(boxes)
    _lib.call("prcnn_nms_device", boxes.size(0), boxes.size(1), _lib.ptr(counts), boxes.data_ptr(),
              float(thresh), int(bool(rotated)), int(max_keep), keep.data_ptr(), num_keep.data_ptr(),
              _lib.current_stream(boxes))
    return 1


def rpn_proposals(xyz, scores, reg, anchor_size, loc_scope, loc_bin_size, num_head_bin, xz_fine,
                  pre_nms_top_n, post_nms_top_n, nms_thresh, rotated, rois, roi_scores):
    """Fused RPN proposal layer (csrc/proposal.hip).  xyz (B,N,3), scores (B,N), reg (B,N,C) ->
    rois (B,post,7), roi_scores (B,post); anchor_size: 3 python floats (h, w, l)."""
    import ctypes
    _chk(xyz, scores, reg, rois, roi_scores)
    anchor = (ctypes.c_float * 3)(*[float(v) for v in anchor_size])
    _lib.call("prcnn_rpn_proposals", xyz.size(0), xyz.size(1), reg.size(2), float(loc_scope), float(loc_bin_size),
              int(num_head_bin), int(bool(xz_fine)), ctypes.cast(anchor, ctypes.c_void_p), int(pre_nms_top_n),
              int(post_nms_top_n), float(nms_thresh), int(bool(rotated)), xyz.data_ptr(), scores.data_ptr(),
              reg.data_ptr(), rois.data_ptr(), roi_scores.data_ptr(), _lib.current_stream(xyz))
    return rois, roi_scores


def rpn_proposals_boxes(scores, boxes, pre_nms_top_n, post_nms_top_n, nms_thresh, rotated, rois, roi_scores):
    """The proposal layer over boxes decoded already (pointnet2_cuda.rpn_tail_lin_boxes_wrapper): scores (B,N), boxes (B,N,7) ->
    rois (B,post,7), roi_scores (B,post)."""
    _chk(scores, boxes, rois, roi_scores)
    _lib.call("prcnn_rpn_proposals_boxes", boxes.size(0), boxes.size(1), int(pre_nms_top_n), int(post_nms_top_n), float(nms_thresh),
              int(bool(rotated)), scores.data_ptr(), boxes.data_ptr(), rois.data_ptr(), roi_scores.data_ptr(), _lib.current_stream(boxes))
    return rois, roi_scores


def rcnn_postprocess_blobs(rois, rcnn_reg, rcnn_cls, anchor_size, loc_scope, loc_bin_size, num_head_bin, y_by_bin,
                           loc_y_scope, loc_y_bin_size, score_thresh, nms_thresh, pred_boxes3d, blobs, scenes_per_blob):
    """rcnn_postprocess with the results as one blob per batch of scenes_per_blob scenes: blobs (B / spb, spb (8 M + 1)) f32, each
    [boxes | scores | num] of its scenes (eval_rcnn.split_detections' layout) -- one D2H copy per batch out of a launch over several."""
    import ctypes
    _chk(rois, rcnn_reg, rcnn_cls, pred_boxes3d, blobs)
    B, M = rois.size(0), rois.size(1)
    if B % scenes_per_blob or blobs.numel() != B * (8 * M + 1):
        raise RuntimeError("iou3d_cuda: %d scenes do not fill blobs of %d (or blobs has the wrong size)" % (B, scenes_per_blob))
    anchor = (ctypes.c_float * 3)(*[float(v) for v in anchor_size])
    _lib.call("prcnn_rcnn_postprocess_blobs", B, M, rcnn_reg.size(2), float(loc_scope), float(loc_bin_size), int(num_head_bin),
              int(bool(y_by_bin)), float(loc_y_scope), float(loc_y_bin_size), ctypes.cast(anchor, ctypes.c_void_p), float(score_thresh),
              float(nms_thresh), rois.data_ptr(), rcnn_reg.data_ptr(), rcnn_cls.data_ptr(), pred_boxes3d.data_ptr(), blobs.data_ptr(),
              int(scenes_per_blob), _lib.current_stream(rois))
    return blobs


def rcnn_postprocess(rois, rcnn_reg, rcnn_cls, anchor_size, loc_scope, loc_bin_size, num_head_bin, y_by_bin,
                     loc_y_scope, loc_y_bin_size, score_thresh, nms_thresh, pred_boxes3d, boxes, scores, num):
    """Fused final stage (csrc/proposal.hip): decode against the RoIs, score threshold, rotated NMS.
    rois (B,M,7), rcnn_reg (B,M,C), rcnn_cls (B,M) -> pred_boxes3d (B,M,7), boxes (B,M,7), scores (B,M), num (B) i32."""
    import ctypes
    _chk(rois, rcnn_reg, rcnn_cls, pred_boxes3d, boxes, scores)
    if num.dtype != torch.int32 or not num.is_cuda:
        raise RuntimeError("iou3d_cuda: num must be a CUDA int32 tensor")
    anchor = (ctypes.c_float * 3)(*[float(v) for v in anchor_size])
    _lib.call("prcnn_rcnn_postprocess", rois.size(0), rois.size(1), rcnn_reg.size(2), float(loc_scope),
              float(loc_bin_size), int(num_head_bin), int(bool(y_by_bin)), float(loc_y_scope), float(loc_y_bin_size),
              ctypes.cast(anchor, ctypes.c_void_p), float(score_thresh), float(nms_thresh), rois.data_ptr(),
              rcnn_reg.data_ptr(), rcnn_cls.data_ptr(), pred_boxes3d.data_ptr(), boxes.data_ptr(), scores.data_ptr(),
              num.data_ptr(), _lib.current_stream(rois))
    return boxes, scores, num
