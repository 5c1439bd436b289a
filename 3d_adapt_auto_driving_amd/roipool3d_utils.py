"""Python-level API of the roipool3d extension with the reference's names
(pointrcnn/lib/utils/roipool3d/roipool3d_utils.py:7-112).  Only ``roipool3d_gpu`` is on the
inference path; the three *_cpu helpers are the host utilities of the reference's dataset / GT-database code
(CPU tensors / numpy; host functions of the same library)."""
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


def _host_f32(t):
    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


def pts_in_boxes3d_cpu(pts, boxes3d):
    """Host utility (roipool3d_utils.py:31-49): for each of the M boxes a boolean mask over the N points."""
    if pts.is_cuda:
        raise NotImplementedError("pts_in_boxes3d_cpu works on CPU tensors (the device path is roipool3d_gpu)")
    flags = torch.zeros((boxes3d.shape[0], pts.shape[0]), dtype=torch.int64)
    roipool3d_cuda.pts_in_boxes3d_cpu(flags, _host_f32(pts), _host_f32(boxes3d))
    return list(flags.bool().unbind(0))


def roipool_pc_cpu(pts, pts_feature, boxes3d, sampled_pt_num):
    """Host utility (roipool3d_utils.py:52-70): first ``sampled_pt_num`` in-box points per box, wrap-around filled.
    -> coordinates (M,S,3), features (M,S,C), empty flags (M) int64."""
    cloud, feats, boxes = _host_f32(pts), _host_f32(pts_feature), _host_f32(boxes3d)
    if cloud.dim() != 2 or cloud.shape[1] != 3 or feats.shape[0] != cloud.shape[0]:
        raise AssertionError("pts %s / pts_feature %s" % (tuple(cloud.shape), tuple(feats.shape)))
    n_box = boxes.shape[0]
    out_xyz = torch.zeros((n_box, sampled_pt_num, 3))
    out_feat = torch.zeros((n_box, sampled_pt_num, feats.shape[1]))
    empty = torch.zeros(n_box, dtype=torch.int64)
    roipool3d_cuda.roipool3d_cpu(cloud, boxes, feats, out_xyz, out_feat, empty)
    return out_xyz, out_feat, empty


def roipool3d_cpu(boxes3d, pts, pts_feature, pts_extra_input, pool_extra_width, sampled_pt_num=512,
                  canonical_transform=True):
    """numpy in / numpy out (roipool3d_utils.py:73-108): boxes enlarged by ``pool_extra_width``, pooled on the host;
    the pooled input = [xyz | extra input], optionally moved into each RoI's canonical frame."""
    import numpy as np
    n_extra = pts_extra_input.shape[1]
    xyz, feat, empty = roipool_pc_cpu(torch.from_numpy(pts),
                                      torch.from_numpy(np.concatenate((pts_extra_input, pts_feature), axis=1)),
                                      torch.from_numpy(kitti_utils.enlarge_box3d(boxes3d, pool_extra_width)),
                                      sampled_pt_num)
    pooled_input = np.concatenate((xyz.numpy(), feat.numpy()[:, :, :n_extra]), axis=2)
    pooled_feature = feat.numpy()[:, :, n_extra:]
    if not canonical_transform:
        return pooled_input, pooled_feature, empty.numpy()
    pooled_input[:, :, 0:3] -= boxes3d[:, None, 0:3]                 # RoI centre to the origin
    headings = np.mod(boxes3d[:, 6], 2 * np.pi)
    for roi in range(pooled_input.shape[0]):                         # rotate about y by the RoI heading
        pooled_input[roi] = kitti_utils.rotate_pc_along_y(pooled_input[roi], headings[roi])
    return pooled_input, pooled_feature
