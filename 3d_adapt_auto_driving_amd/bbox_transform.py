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


def _bin_centre(bins, bin_size, scope):
    """Centre of the chosen bin: index * size + size / 2 - scope, evaluated in that order (f32)."""
    return bins.float() * bin_size + bin_size / 2 - scope


def _decode_planar(reg, nbin, loc_bin_size, loc_scope, with_residual):
    """x / z from the first 2 (or 4) blocks of ``nbin`` channels -> (pos_x, pos_z, channels consumed)."""
    x_bin = torch.argmax(reg[:, 0:nbin], dim=1)
    z_bin = torch.argmax(reg[:, nbin:2 * nbin], dim=1)
    pos_x = _bin_centre(x_bin, loc_bin_size, loc_scope)
    pos_z = _bin_centre(z_bin, loc_bin_size, loc_scope)
    if not with_residual:
        return pos_x, pos_z, 2 * nbin
    pos_x += _pick(reg, 2 * nbin, nbin, x_bin) * loc_bin_size
    pos_z += _pick(reg, 3 * nbin, nbin, z_bin) * loc_bin_size
    return pos_x, pos_z, 4 * nbin


def _decode_heading(reg, at, num_head_bin, fine):
    """Heading from ``num_head_bin`` class scores + as many normalised residuals starting at channel ``at``."""
    ry_bin = torch.argmax(reg[:, at:at + num_head_bin], dim=1)
    res_norm = _pick(reg, at + num_head_bin, num_head_bin, ry_bin)
    if fine:          # refinement stage: bins span +-pi/4 around the RoI heading
        per_class = (np.pi / 2) / num_head_bin
        return (ry_bin.float() * per_class + per_class / 2) + res_norm * (per_class / 2) - np.pi / 4
    per_class = (2 * np.pi) / num_head_bin          # proposal stage: full circle, bin centres at 0, 30, ... degrees
    ry = (ry_bin.float() * per_class + res_norm * (per_class / 2)) % (2 * np.pi)
    return torch.where(ry > np.pi, ry - 2 * np.pi, ry)          # wrap into (-pi, pi] without a host sync


def decode_bbox_target(roi_box3d, pred_reg, loc_scope, loc_bin_size, num_head_bin, anchor_size,
                       get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25,
                       get_ry_fine=False):
    """roi_box3d (N,3) points or (N,7) RoIs, pred_reg (N,C) -> boxes (N,7) in the input frame."""
    anchor_size = anchor_size.to(pred_reg.device)
    nbin = int(loc_scope / loc_bin_size) * 2
    pos_x, pos_z, at = _decode_planar(pred_reg, nbin, loc_bin_size, loc_scope, get_xz_fine)

    if get_y_by_bin:
        nbin_y = int(loc_y_scope / loc_y_bin_size) * 2
        y_bin = torch.argmax(pred_reg[:, at:at + nbin_y], dim=1)
        y_res = _pick(pred_reg, at + nbin_y, nbin_y, y_bin) * loc_y_bin_size
        pos_y = _bin_centre(y_bin, loc_y_bin_size, loc_y_scope) + y_res
        pos_y = pos_y + roi_box3d[:, 1]
        at += 2 * nbin_y
    else:
        pos_y = roi_box3d[:, 1] + pred_reg[:, at]
        at += 1

    ry = _decode_heading(pred_reg, at, num_head_bin, get_ry_fine)
    at += 2 * num_head_bin
    if at + 3 != pred_reg.shape[1]:
        raise AssertionError("regression width %d != layout %d" % (pred_reg.shape[1], at + 3))
    hwl = pred_reg[:, at:at + 3] * anchor_size + anchor_size

    box = torch.cat((pos_x.view(-1, 1), pos_y.view(-1, 1), pos_z.view(-1, 1), hwl, ry.view(-1, 1)), dim=1)
    if roi_box3d.shape[1] == 7:          # decoded in the RoI's canonical frame: rotate back, add the RoI heading
        roi_ry = roi_box3d[:, 6]
        box = rotate_pc_along_y_torch(box, -roi_ry)
        box[:, 6] += roi_ry
    box[:, 0] += roi_box3d[:, 0]
    box[:, 2] += roi_box3d[:, 2]
    return box
