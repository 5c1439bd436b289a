"""PointRCNN = RPN -> proposal layer -> RCNN (inference flow of pointrcnn/lib/net/point_rcnn.py:8-70).
Top-level children are named ``rpn`` and ``rcnn_net`` (checkpoint keys)."""
import torch
import torch.nn as nn

from .rpn import RPN
from .rcnn_net import RCNNNet


class PointRCNN(nn.Module):
    def __init__(self, cfg, num_classes, use_xyz=True, mode="TRAIN"):
        super().__init__()
        self.cfg = cfg
        assert cfg.RPN.ENABLED or cfg.RCNN.ENABLED
        if cfg.RPN.ENABLED:
            self.rpn = RPN(cfg, use_xyz=use_xyz, mode=mode)
        if cfg.RCNN.ENABLED:
            if cfg.RCNN.BACKBONE != "pointnet":
                raise NotImplementedError("RCNN backbone %r" % cfg.RCNN.BACKBONE)
            self.rcnn_net = RCNNNet(cfg, num_classes=num_classes, input_channels=128, use_xyz=use_xyz)

    def forward(self, input_data):
        cfg = self.cfg
        if not cfg.RPN.ENABLED:
            if cfg.RCNN.ENABLED:
                return self.rcnn_net(input_data)
            raise NotImplementedError
        output = {}
        with torch.set_grad_enabled((not cfg.RPN.FIXED) and self.training):
            if cfg.RPN.FIXED:
                self.rpn.eval()
            rpn_output = self.rpn(input_data)
            output.update(rpn_output)
        if cfg.RCNN.ENABLED:
            with torch.no_grad():
                rpn_cls, rpn_reg = rpn_output["rpn_cls"], rpn_output["rpn_reg"]
                backbone_xyz, backbone_features = rpn_output["backbone_xyz"], rpn_output["backbone_features"]
                rpn_scores_raw = rpn_cls[:, :, 0]
                seg_mask = (torch.sigmoid(rpn_scores_raw) > cfg.RPN.SCORE_THRESH).float()
                pts_depth = torch.norm(backbone_xyz, p=2, dim=2)
                rois, roi_scores_raw = self.rpn.proposal_layer(rpn_scores_raw, rpn_reg, backbone_xyz)
                output["rois"] = rois
                output["roi_scores_raw"] = roi_scores_raw
                output["seg_result"] = seg_mask
            rcnn_input = {"rpn_xyz": backbone_xyz, "rpn_features": backbone_features.permute((0, 2, 1)),
                          "seg_mask": seg_mask, "roi_boxes3d": rois, "pts_depth": pts_depth}
            if self.training:
                rcnn_input["gt_boxes3d"] = input_data["gt_boxes3d"]
            output.update(self.rcnn_net(rcnn_input))
        return output
