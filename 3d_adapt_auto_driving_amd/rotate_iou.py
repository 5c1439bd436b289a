"""Rotated-IoU entry points of the KITTI AP evaluator (counterparts of evaluate/rotate_iou.py:294-329
``rotate_iou_gpu_eval`` and of its two callers evaluate/eval2.py:131-168 ``bev_box_overlap`` /
``d3_box_overlap``).  The pairwise rotated intersection runs in csrc/rotate_iou.hip; numpy in,
numpy out, like the reference (which uploads, launches its numba.cuda kernel and downloads)."""
import numpy as np
import torch

from . import _lib


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes (N,5), query_boxes (K,5) as [cx, cy, w, h, angle] -> (N,K) in boxes.dtype.
    criterion: -1 IoU, 0 intersection/area(query), 1 intersection/area(box), 2 intersection
    (the kernel hands (query, box) to the device function, rotate_iou.py:287-291)."""
    boxes = np.asarray(boxes)
    query_boxes = np.asarray(query_boxes)
    out_dtype = boxes.dtype
    n, k = boxes.shape[0], query_boxes.shape[0]
    iou = np.zeros((n, k), dtype=np.float32)
    if n == 0 or k == 0:
        return iou.astype(out_dtype)
    dev = torch.device("cuda", device_id)
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(query_boxes, dtype=np.float32)).to(dev)
    out = torch.zeros((n, k), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("prcnn_rotate_iou_eval", n, k, b.data_ptr(), q.data_ptr(), out.data_ptr(), int(criterion),
                  _lib.current_stream(out))
    return out.cpu().numpy().astype(out_dtype)


def bev_box_overlap(boxes, qboxes, criterion=-1):
    return rotate_iou_gpu_eval(boxes, qboxes, criterion)


def d3_box_overlap(boxes, qboxes, criterion=-1):
    """Camera-frame 3D overlap: BEV intersection (criterion 2) x height overlap, normalised as the
    criterion says.  boxes (N,7) = [x, y, z, l, h, w, ry] with y the bottom (eval2.py:136-161;
    that part is host code in the reference too -- a numba CPU jit -- vectorised here)."""
    boxes = np.asarray(boxes)
    qboxes = np.asarray(qboxes)
    rinc = rotate_iou_gpu_eval(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2)
    iw = (np.minimum(boxes[:, None, 1], qboxes[None, :, 1]) -
          np.maximum(boxes[:, None, 1] - boxes[:, None, 4], qboxes[None, :, 1] - qboxes[None, :, 4]))
    area1 = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
    area2 = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
    inc = iw * rinc
    if criterion == -1:
        ua = area1 + area2 - inc
    elif criterion == 0:
        ua = np.broadcast_to(area1, inc.shape)
    elif criterion == 1:
        ua = np.broadcast_to(area2, inc.shape)
    else:
        ua = inc
    valid = (rinc > 0) & (iw > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        res = np.where(valid, inc / ua, np.where(rinc > 0, 0.0, rinc))
    return res.astype(rinc.dtype)
