"""Python-level API of the iou3d extension with the reference's names
(pointrcnn/lib/utils/iou3d/iou3d_utils.py:6-87): boxes_iou_bev, boxes_iou3d_gpu, nms_gpu,
nms_normal_gpu -- plus nms_device_batched, the host-sync-free form the batched proposal layer uses.
"""
import torch

from .dropin import iou3d_cuda
from . import kitti_utils


def boxes_iou_bev(boxes_a, boxes_b):
    """(M,5) x (N,5) BEV boxes -> rotated IoU (M,N)."""
    ans = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans)
    return ans


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7) x (M,7) [x,y,z,h,w,l,ry] -> 3D IoU (N,M): BEV overlap x height overlap / union volume."""
    a_bev = kitti_utils.boxes3d_to_bev_torch(boxes_a)
    b_bev = kitti_utils.boxes3d_to_bev_torch(boxes_b)
    overlaps_bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_overlap_bev_gpu(a_bev.contiguous(), b_bev.contiguous(), overlaps_bev)

    a_min, a_max = (boxes_a[:, 1] - boxes_a[:, 3]).view(-1, 1), boxes_a[:, 1].view(-1, 1)
    b_min, b_max = (boxes_b[:, 1] - boxes_b[:, 3]).view(1, -1), boxes_b[:, 1].view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)


def _nms(fn, boxes, scores, thresh):
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep = torch.zeros(boxes.size(0), dtype=torch.int64)  # CPU, as the extension expects
    num_out = fn(boxes, keep, thresh)
    return order[keep[:num_out].to(boxes.device)].contiguous()


def nms_gpu(boxes, scores, thresh):
    """Rotated-IoU greedy NMS; returns indices into ``boxes`` in descending score order."""
    return _nms(iou3d_cuda.nms_gpu, boxes, scores, thresh)


def nms_normal_gpu(boxes, scores, thresh):
    """Axis-aligned-IoU greedy NMS (the rotation column is ignored)."""
    return _nms(iou3d_cuda.nms_normal_gpu, boxes, scores, thresh)


def nms_device_batched(boxes_sorted, counts, thresh, rotated, max_keep):
    """boxes_sorted (P, n_max, 5) already in descending score order, counts (P) i32 device or None.
    Returns keep (P, max_keep) i32 padded with -1 and num_keep (P) i32, all on the device, with
    no host synchronisation."""
    P = boxes_sorted.size(0)
    keep = torch.empty((P, max_keep), dtype=torch.int32, device=boxes_sorted.device)
    num = torch.empty((P,), dtype=torch.int32, device=boxes_sorted.device)
    iou3d_cuda.nms_device(boxes_sorted.contiguous(), counts, thresh, rotated, max_keep, keep, num)
    return keep, num
