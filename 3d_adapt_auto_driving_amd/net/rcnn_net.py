"""Box refinement network (inference branch of pointrcnn/lib/net/rcnn_net.py:14-190): RoI point
pooling, canonical transform, xyz-up / merge MLPs, three SA levels, cls / reg heads."""
import torch
import torch.nn as nn

from ..pointnet2.pointnet2_modules import PointnetSAModule
from ..pointnet2 import pytorch_utils as pt_utils
from ..pointnet2 import fused_mlp
from .. import kitti_utils
from .. import roipool3d_utils
from .rpn import _head


class RCNNNet(nn.Module):
    def __init__(self, cfg, num_classes, input_channels=0, use_xyz=True):
        super().__init__()
        self.cfg = cfg
        R = cfg.RCNN
        if R.USE_RPN_FEATURES:
            self.rcnn_input_channel = 3 + int(R.USE_INTENSITY) + int(R.USE_MASK) + int(R.USE_DEPTH)
            self.xyz_up_layer = pt_utils.SharedMLP([self.rcnn_input_channel] + list(R.XYZ_UP_LAYER), bn=R.USE_BN)
            c_out = R.XYZ_UP_LAYER[-1]
            self.merge_down_layer = pt_utils.SharedMLP([c_out * 2, c_out], bn=R.USE_BN)

        self.SA_modules = nn.ModuleList()
        channel_in = input_channels
        for k in range(len(R.SA_CONFIG.NPOINTS)):
            spec = [channel_in] + list(R.SA_CONFIG.MLPS[k])
            npoint = R.SA_CONFIG.NPOINTS[k] if R.SA_CONFIG.NPOINTS[k] != -1 else None
            self.SA_modules.append(PointnetSAModule(npoint=npoint, radius=R.SA_CONFIG.RADIUS[k],
                                                    nsample=R.SA_CONFIG.NSAMPLE[k], mlp=spec, use_xyz=use_xyz,
                                                    bn=R.USE_BN))
            channel_in = spec[-1]

        cls_channel = 1 if num_classes == 2 else num_classes
        self.cls_layer = _head(channel_in, R.CLS_FC, cls_channel, R.USE_BN, R.DP_RATIO)
        per_loc_bin_num = int(R.LOC_SCOPE / R.LOC_BIN_SIZE) * 2
        loc_y_bin_num = int(R.LOC_Y_SCOPE / R.LOC_Y_BIN_SIZE) * 2
        reg_channel = per_loc_bin_num * 4 + R.NUM_HEAD_BIN * 2 + 3
        reg_channel += (1 if not R.LOC_Y_BY_BIN else loc_y_bin_num * 2)
        self.reg_layer = _head(channel_in, R.REG_FC, reg_channel, R.USE_BN, R.DP_RATIO)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():                      # xavier, as rcnn_net.py:85,100-112
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layer[-1].conv.weight, mean=0, std=0.001)

    def pool_rois(self, input_data):
        """RoI pooling + canonical transform -> pts_input (B*M, NUM_POINTS, 3+extra+C)."""
        R = self.cfg.RCNN
        rpn_xyz, rpn_features = input_data["rpn_xyz"], input_data["rpn_features"]
        batch_rois = input_data["roi_boxes3d"]
        extra = [input_data["rpn_intensity"].unsqueeze(dim=2)] if R.USE_INTENSITY else []
        extra.append(input_data["seg_mask"].unsqueeze(dim=2))
        if R.USE_DEPTH:
            extra.append((input_data["pts_depth"] / 70.0 - 0.5).unsqueeze(dim=2))
        pts_feature = torch.cat(extra + [rpn_features], dim=2)
        pooled, _empty = roipool3d_utils.roipool3d_gpu(rpn_xyz, pts_feature, batch_rois, R.POOL_EXTRA_WIDTH,
                                                       sampled_pt_num=R.NUM_POINTS)
        B, M = batch_rois.shape[0], batch_rois.shape[1]
        pooled[:, :, :, 0:3] -= batch_rois[:, :, 0:3].unsqueeze(dim=2)
        flat = pooled.view(B * M, pooled.shape[2], pooled.shape[3])
        # every RoI is rotated by its own heading; one batched call == the reference's per-scene loop
        flat[:, :, 0:3] = kitti_utils.rotate_pc_along_y_torch(flat[:, :, 0:3], batch_rois.reshape(-1, 7)[:, 6])
        return flat

    def forward(self, input_data):
        R = self.cfg.RCNN
        if self.training:
            raise NotImplementedError("RCNNNet: training branch (proposal target layer) is out of scope")
        pts_input = self.pool_rois(input_data) if R.ROI_SAMPLE_JIT else input_data["pts_input"]

        xyz = pts_input[..., 0:3].contiguous()
        if R.USE_RPN_FEATURES:
            xyz_input = pts_input[..., 0:self.rcnn_input_channel].transpose(1, 2).unsqueeze(dim=3)
            xyz_feature = fused_mlp.run(self.xyz_up_layer, xyz_input.contiguous(), pool=False)
            rpn_feature = pts_input[..., self.rcnn_input_channel:].transpose(1, 2).unsqueeze(dim=3)
            merged = fused_mlp.run(self.merge_down_layer, torch.cat((xyz_feature, rpn_feature), dim=1), pool=False)
            l_xyz, l_features = [xyz], [merged.squeeze(dim=3).contiguous()]
        else:
            feats = pts_input[..., 3:].transpose(1, 2).contiguous() if pts_input.size(-1) > 3 else None
            l_xyz, l_features = [xyz], [feats]

        for sa in self.SA_modules:
            nx, nf = sa(l_xyz[-1], l_features[-1])
            l_xyz.append(nx)
            l_features.append(nf)

        rcnn_cls = self.cls_layer(l_features[-1]).transpose(1, 2).contiguous().squeeze(dim=1)
        rcnn_reg = self.reg_layer(l_features[-1]).transpose(1, 2).contiguous().squeeze(dim=1)
        return {"rcnn_cls": rcnn_cls, "rcnn_reg": rcnn_reg}
