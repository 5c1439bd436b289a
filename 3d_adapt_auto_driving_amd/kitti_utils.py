"""Box-geometry helpers of the inference path (counterparts of pointrcnn/lib/utils/kitti_utils.py
:45-63 rotate_pc_along_y_torch, :66-101 boxes3d_to_corners3d, :134-147 boxes3d_to_bev_torch,
:150-160 enlarge_box3d).  Boxes are [x, y, z, h, w, l, ry] in rect-camera coordinates with y the
bottom centre."""
import numpy as np
import torch


def rotate_pc_along_y(pc, rot_angle):
    """numpy: pc (N, 3+C) in the rect camera frame, rot_angle scalar; x and z are rotated in place
    (kitti_utils.py:32-42)."""
    import numpy as np
    cosval, sinval = np.cos(rot_angle), np.sin(rot_angle)
    rotmat = np.array([[cosval, -sinval], [sinval, cosval]])
    pc[:, [0, 2]] = np.dot(pc[:, [0, 2]], np.transpose(rotmat))
    return pc


def rotate_pc_along_y_torch(pc, rot_angle):
    """pc (N, P, 3+C) rotated in place about y by rot_angle (N): [x z] <- [x z] @ R^T with
    R = [[cos, -sin], [sin, cos]] (a batched 2x2 matmul, as the reference does it)."""
    cosa = torch.cos(rot_angle).view(-1, 1)
    sina = torch.sin(rot_angle).view(-1, 1)
    R = torch.stack([torch.cat([cosa, -sina], dim=1), torch.cat([sina, cosa], dim=1)], dim=1)  # (N,2,2)
    xz = torch.stack((pc[:, :, 0], pc[:, :, 2]), dim=2)              # (N,P,2), no index tensor (no host sync)
    out = torch.matmul(xz, R.permute(0, 2, 1))
    pc[:, :, 0] = out[:, :, 0]
    pc[:, :, 2] = out[:, :, 1]
    return pc


def boxes3d_to_bev_torch(boxes3d):
    """(N,7) -> (N,5) [x1, y1, x2, y2, ry] with x along l and y(z) along w."""
    cu, cv = boxes3d[:, 0], boxes3d[:, 2]
    half_l, half_w = boxes3d[:, 5] / 2, boxes3d[:, 4] / 2
    return torch.stack([cu - half_l, cv - half_w, cu + half_l, cv + half_w, boxes3d[:, 6]], dim=1)


def enlarge_box3d(boxes3d, extra_width):
    """h, w, l grow by 2*extra_width and the bottom centre drops by extra_width."""
    large = boxes3d.copy() if isinstance(boxes3d, np.ndarray) else boxes3d.clone()
    large[:, 3:6] += extra_width * 2
    large[:, 1] += extra_width
    return large


def boxes3d_to_corners3d(boxes3d, rotate=True):
    """numpy (N,7) -> (N,8,3) corners: 0-3 bottom face (y = box y), 4-7 top face (y - h)."""
    n = boxes3d.shape[0]
    h, w, l = boxes3d[:, 3], boxes3d[:, 4], boxes3d[:, 5]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1], dtype=np.float32)
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1], dtype=np.float32)
    xc = (l[:, None] / 2.).astype(np.float32) * sx[None]
    zc = (w[:, None] / 2.).astype(np.float32) * sz[None]
    yc = np.zeros((n, 8), dtype=np.float32)
    yc[:, 4:8] = -h.reshape(n, 1)
    if rotate:
        ry = boxes3d[:, 6]
        zeros, ones = np.zeros(ry.size, dtype=np.float32), np.ones(ry.size, dtype=np.float32)
        rot = np.array([[np.cos(ry), zeros, -np.sin(ry)], [zeros, ones, zeros], [np.sin(ry), zeros, np.cos(ry)]])
        rot = np.transpose(rot, (2, 0, 1))  # (N,3,3)
        local = np.stack([xc, yc, zc], axis=2)  # (N,8,3)
        turned = np.matmul(local, rot)
        xc, yc, zc = turned[:, :, 0], turned[:, :, 1], turned[:, :, 2]
    x = boxes3d[:, 0].reshape(-1, 1) + xc.reshape(-1, 8)
    y = boxes3d[:, 1].reshape(-1, 1) + yc.reshape(-1, 8)
    z = boxes3d[:, 2].reshape(-1, 1) + zc.reshape(-1, 8)
    return np.stack([x, y, z], axis=2).astype(np.float32)
