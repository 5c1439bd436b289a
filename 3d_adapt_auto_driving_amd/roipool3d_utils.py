"""Python-level API of the roipool3d extension with the reference's names
(pointrcnn/lib/utils/roipool3d/roipool3d_utils.py:7-112).  Only ``roipool3d_gpu`` is on the
inference path; the three *_cpu helpers serve the reference's dataset / GT-database code
(out of scope) and raise NotImplementedError."""
import torch

from .dropin import roipool3d_cuda
from . import kitti_utils


def roipool3d_gpu(pts, pts_feature, boxes3d, pool_extra_width, sampled_pt_num=512):
    """pts (B,N,3), pts_feature (B,N,C), boxes3d (B,M,7) ->
    pooled_features (B,M,sampled_pt_num,3+C), pooled_empty_flag (B,M) i32."""
    batch_size, boxes_num, feature_len = pts.shape[0], boxes3d.shape[1], pts_feature.shape[2]
    pooled_boxes3d = kitti_utils.enlarge_box3d(boxes3d.view(-1, 7), pool_extra_width).view(batch_size, -1, 7)
    pooled_features = torch.zeros((batch_size, boxes_num, sampled_pt_num, 3 + feature_len),
                                  dtype=torch.float32, device=pts.device)
    pooled_empty_flag = torch.zeros((batch_size, boxes_num), dtype=torch.int32, device=pts.device)
    roipool3d_cuda.forward(pts.contiguous(), pooled_boxes3d.contiguous(), pts_feature.contiguous(),
                           pooled_features, pooled_empty_flag)
    return pooled_features, pooled_empty_flag


def pts_in_boxes3d_cpu(pts, boxes3d):
    raise NotImplementedError("pts_in_boxes3d_cpu: host-side dataset utility, not part of the MI355X hot path")


def roipool_pc_cpu(pts, pts_feature, boxes3d, sampled_pt_num):
    raise NotImplementedError("roipool_pc_cpu: host-side dataset utility, not part of the MI355X hot path")


def roipool3d_cpu(boxes3d, pts, pts_feature, pts_extra_input, pool_extra_width, sampled_pt_num=512,
                  canonical_transform=True):
    raise NotImplementedError("roipool3d_cpu: host-side dataset utility, not part of the MI355X hot path")
