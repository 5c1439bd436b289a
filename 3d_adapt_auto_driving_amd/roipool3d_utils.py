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


def pts_in_boxes3d_cpu(pts, boxes3d):
    """pts (N,3), boxes3d (M,7) CPU tensors -> list of M boolean masks (N) (roipool3d_utils.py:31-49)."""
    if pts.is_cuda:
        raise NotImplementedError
    pts = pts.float().contiguous()
    boxes3d = boxes3d.float().contiguous()
    pts_flag = torch.zeros((boxes3d.size(0), pts.size(0)), dtype=torch.int64)
    roipool3d_cuda.pts_in_boxes3d_cpu(pts_flag, pts, boxes3d)
    return [pts_flag[k] > 0 for k in range(boxes3d.shape[0])]


def roipool_pc_cpu(pts, pts_feature, boxes3d, sampled_pt_num):
    """pts (N,3), pts_feature (N,C), boxes3d (M,7) -> pooled_pts (M,S,3), pooled_features (M,S,C), empty (M) i64
    (roipool3d_utils.py:52-70)."""
    pts = pts.cpu().float().contiguous()
    pts_feature = pts_feature.cpu().float().contiguous()
    boxes3d = boxes3d.cpu().float().contiguous()
    assert pts.shape[0] == pts_feature.shape[0] and pts.shape[1] == 3, "%s %s" % (pts.shape, pts_feature.shape)
    pooled_pts = torch.zeros((boxes3d.shape[0], sampled_pt_num, 3), dtype=torch.float32)
    pooled_features = torch.zeros((boxes3d.shape[0], sampled_pt_num, pts_feature.shape[1]), dtype=torch.float32)
    pooled_empty_flag = torch.zeros(boxes3d.shape[0], dtype=torch.int64)
    roipool3d_cuda.roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag)
    return pooled_pts, pooled_features, pooled_empty_flag


def roipool3d_cpu(boxes3d, pts, pts_feature, pts_extra_input, pool_extra_width, sampled_pt_num=512,
                  canonical_transform=True):
    """numpy in, numpy out (roipool3d_utils.py:73-108): enlarge, pool on the host, optional canonical transform."""
    import numpy as np
    pooled_boxes3d = kitti_utils.enlarge_box3d(boxes3d, pool_extra_width)
    pts_feature_all = np.concatenate((pts_extra_input, pts_feature), axis=1)
    pooled_pts, pooled_features, pooled_empty_flag = roipool_pc_cpu(
        torch.from_numpy(pts), torch.from_numpy(pts_feature_all), torch.from_numpy(pooled_boxes3d), sampled_pt_num)
    extra_input_len = pts_extra_input.shape[1]
    sampled_pts_input = torch.cat((pooled_pts, pooled_features[:, :, 0:extra_input_len]), dim=2).numpy()
    sampled_pts_feature = pooled_features[:, :, extra_input_len:].numpy()
    if canonical_transform:
        roi_ry = boxes3d[:, 6] % (2 * np.pi)
        roi_center = boxes3d[:, 0:3]
        sampled_pts_input[:, :, 0:3] = sampled_pts_input[:, :, 0:3] - roi_center[:, np.newaxis, :]
        for k in range(sampled_pts_input.shape[0]):
            sampled_pts_input[k] = kitti_utils.rotate_pc_along_y(sampled_pts_input[k], roi_ry[k])
        return sampled_pts_input, sampled_pts_feature
    return sampled_pts_input, sampled_pts_feature, pooled_empty_flag.numpy()
