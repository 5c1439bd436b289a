"""Bin-based box decoding (counterpart of pointrcnn/lib/utils/bbox_transform.py:5-121).

Regression layout along the channel axis (per_loc_bin_num = 2*int(loc_scope/loc_bin_size)):
  [x bins | z bins | (x residuals | z residuals if get_xz_fine) | y offset or (y bins | y residuals)
   | ry bins | ry residuals | h,w,l residuals]
The arithmetic is written in the reference's operation order so f32 results agree bit for bit on
the same backend.
"""
import numpy as np
import torch


def rotate_pc_along_y_torch(pc, rot_angle):
    """pc (N, 3+C), rot_angle (N): rotate the (x, z) pair of every row by its own angle."""
    cosa = torch.cos(rot_angle).view(-1, 1)
    sina = torch.sin(rot_angle).view(-1, 1)
    R = torch.stack([torch.cat([cosa, -sina], dim=1), torch.cat([sina, cosa], dim=1)], dim=1)  # (N,2,2)
    # same batched 2x2 matmul as the reference; the columns are picked / written back without an
    # index tensor (a Python-list index uploads a tensor and synchronises the stream)
    xz = torch.stack((pc[:, 0], pc[:, 2]), dim=1).unsqueeze(dim=1)  # (N,1,2)
    out = torch.matmul(xz, R.permute(0, 2, 1)).squeeze(dim=1)
    pc[:, 0] = out[:, 0]
    pc[:, 2] = out[:, 1]
    return pc


def _pick(reg, lo, bins, idx):
    """reg[:, lo:lo+bins] gathered at idx (N,) -> (N,)."""
    return torch.gather(reg[:, lo:lo + bins], 1, idx.unsqueeze(1)).squeeze(1)


def decode_bbox_target(roi_box3d, pred_reg, loc_scope, loc_bin_size, num_head_bin, anchor_size,
                       get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25,
                       get_ry_fine=False):
    """roi_box3d (N,3) points or (N,7) RoIs, pred_reg (N,C) -> boxes (N,7) in the input frame."""
    anchor_size = anchor_size.to(pred_reg.device)
    nbin = int(loc_scope / loc_bin_size) * 2
    nbin_y = int(loc_y_scope / loc_y_bin_size) * 2

    x_bin = torch.argmax(pred_reg[:, 0:nbin], dim=1)
    z_bin = torch.argmax(pred_reg[:, nbin:2 * nbin], dim=1)
    pos_x = x_bin.float() * loc_bin_size + loc_bin_size / 2 - loc_scope
    pos_z = z_bin.float() * loc_bin_size + loc_bin_size / 2 - loc_scope
    cursor = 2 * nbin
    if get_xz_fine:
        pos_x += _pick(pred_reg, 2 * nbin, nbin, x_bin) * loc_bin_size
        pos_z += _pick(pred_reg, 3 * nbin, nbin, z_bin) * loc_bin_size
        cursor = 4 * nbin

    if get_y_by_bin:
        y_bin = torch.argmax(pred_reg[:, cursor:cursor + nbin_y], dim=1)
        y_res = _pick(pred_reg, cursor + nbin_y, nbin_y, y_bin) * loc_y_bin_size
        pos_y = y_bin.float() * loc_y_bin_size + loc_y_bin_size / 2 - loc_y_scope + y_res
        pos_y = pos_y + roi_box3d[:, 1]
        cursor += 2 * nbin_y
    else:
        pos_y = roi_box3d[:, 1] + pred_reg[:, cursor]
        cursor += 1

    ry_bin = torch.argmax(pred_reg[:, cursor:cursor + num_head_bin], dim=1)
    ry_res_norm = _pick(pred_reg, cursor + num_head_bin, num_head_bin, ry_bin)
    if get_ry_fine:   # heading within +-pi/4 of the RoI heading
        angle_per_class = (np.pi / 2) / num_head_bin
        ry_res = ry_res_norm * (angle_per_class / 2)
        ry = (ry_bin.float() * angle_per_class + angle_per_class / 2) + ry_res - np.pi / 4
    else:             # full circle, bin centres at 0, 30, ... degrees
        angle_per_class = (2 * np.pi) / num_head_bin
        ry_res = ry_res_norm * (angle_per_class / 2)
        ry = (ry_bin.float() * angle_per_class + ry_res) % (2 * np.pi)
        ry = torch.where(ry > np.pi, ry - 2 * np.pi, ry)          # == ry[ry > pi] -= 2 pi, without a host sync
    cursor += 2 * num_head_bin

    assert cursor + 3 == pred_reg.shape[1], "regression width %d != layout %d" % (pred_reg.shape[1], cursor + 3)
    size_res_norm = pred_reg[:, cursor:cursor + 3]
    hwl = size_res_norm * anchor_size + anchor_size

    ret = torch.cat((pos_x.view(-1, 1), pos_y.view(-1, 1), pos_z.view(-1, 1), hwl, ry.view(-1, 1)), dim=1)
    if roi_box3d.shape[1] == 7:  # decoded in the RoI's canonical frame: rotate back
        roi_ry = roi_box3d[:, 6]
        ret = rotate_pc_along_y_torch(ret, -roi_ry)
        ret[:, 6] += roi_ry
    ret[:, 0] += roi_box3d[:, 0]
    ret[:, 2] += roi_box3d[:, 2]
    return ret
