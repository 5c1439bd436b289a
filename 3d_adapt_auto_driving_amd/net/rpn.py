"""Region proposal network: backbone + per-point cls / reg heads + proposal layer (inference part
of pointrcnn/lib/net/rpn.py:11-83; losses are training-only and out of scope)."""
import numpy as np
import torch.nn as nn

from ..pointnet2 import pytorch_utils as pt_utils
from . import pointnet2_msg
from .proposal_layer import ProposalLayer


def _head(in_ch, hidden, out_ch, bn, dp_ratio):
    layers, pre = [], in_ch
    for h in hidden:
        layers.append(pt_utils.Conv1d(pre, h, bn=bn))
        pre = h
    layers.append(pt_utils.Conv1d(pre, out_ch, activation=None))
    if dp_ratio >= 0:
        layers.insert(1, nn.Dropout(dp_ratio))   # index 1 -> conv keys are 0 and 2.. in checkpoints
    return nn.Sequential(*layers)


class RPN(nn.Module):
    def __init__(self, cfg, use_xyz=True, mode="TRAIN"):
        super().__init__()
        self.cfg = cfg
        self.training_mode = (mode == "TRAIN")
        if cfg.RPN.BACKBONE != "pointnet2_msg":
            raise NotImplementedError("RPN backbone %r" % cfg.RPN.BACKBONE)
        self.backbone_net = pointnet2_msg.get_model(cfg, input_channels=int(cfg.RPN.USE_INTENSITY), use_xyz=use_xyz)

        feat = cfg.RPN.FP_MLPS[0][-1]
        self.rpn_cls_layer = _head(feat, cfg.RPN.CLS_FC, 1, cfg.RPN.USE_BN, cfg.RPN.DP_RATIO)
        per_loc_bin_num = int(cfg.RPN.LOC_SCOPE / cfg.RPN.LOC_BIN_SIZE) * 2
        reg_channel = per_loc_bin_num * (4 if cfg.RPN.LOC_XZ_FINE else 2) + cfg.RPN.NUM_HEAD_BIN * 2 + 3 + 1
        self.rpn_reg_layer = _head(feat, cfg.RPN.REG_FC, reg_channel, cfg.RPN.USE_BN, cfg.RPN.DP_RATIO)
        self.proposal_layer = ProposalLayer(cfg, mode=mode)
        self.init_weights()

    def init_weights(self):
        if self.cfg.RPN.LOSS_CLS in ["SigmoidFocalLoss"]:
            pi = 0.01
            nn.init.constant_(self.rpn_cls_layer[2].conv.bias, -np.log((1 - pi) / pi))
        nn.init.normal_(self.rpn_reg_layer[-1].conv.weight, mean=0, std=0.001)

    def forward(self, input_data):
        pts_input = input_data["pts_input"]
        backbone_xyz, backbone_features = self.backbone_net(pts_input)            # (B,N,3), (B,C,N)
        rpn_cls = self.rpn_cls_layer(backbone_features).transpose(1, 2).contiguous()  # (B,N,1)
        rpn_reg = self.rpn_reg_layer(backbone_features).transpose(1, 2).contiguous()  # (B,N,C)
        return {"rpn_cls": rpn_cls, "rpn_reg": rpn_reg,
                "backbone_xyz": backbone_xyz, "backbone_features": backbone_features}
