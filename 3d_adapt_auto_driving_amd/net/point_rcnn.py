"""Two-stage detector assembled from an RPN, its proposal layer and the RCNN refinement head
(inference flow of pointrcnn/lib/net/point_rcnn.py:8-70).  The two children must be called ``rpn`` and
``rcnn_net``: those names are the prefixes of the reference's checkpoint keys."""
import torch
import torch.nn as nn

from .rcnn_net import RCNNNet
from .rpn import RPN


class PointRCNN(nn.Module):
    def __init__(self, cfg, num_classes, use_xyz=True, mode="TRAIN"):
        super().__init__()
        self.cfg = cfg
        if not (cfg.RPN.ENABLED or cfg.RCNN.ENABLED):
            raise AssertionError("at least one of RPN / RCNN must be enabled")
        if cfg.RPN.ENABLED:
            self.rpn = RPN(cfg, use_xyz=use_xyz, mode=mode)
        if cfg.RCNN.ENABLED:
            if cfg.RCNN.BACKBONE != "pointnet":
                raise NotImplementedError("RCNN backbone %r" % cfg.RCNN.BACKBONE)
            # 128 = width of the RPN backbone features handed to the second stage
            self.rcnn_net = RCNNNet(cfg, num_classes=num_classes, input_channels=128, use_xyz=use_xyz)

    # ---- stage 1: point-wise foreground score + box regression, with the RPN frozen when cfg.RPN.FIXED
    def _first_stage(self, batch):
        frozen = self.cfg.RPN.FIXED
        if frozen:
            self.rpn.eval()
        with torch.set_grad_enabled(self.training and not frozen):
            return self.rpn(batch)

    # ---- hand-over: scores -> segmentation mask, depth, proposals
    @torch.no_grad()
    def _second_stage_inputs(self, first):
        xyz = first["backbone_xyz"]
        raw = first["rpn_cls"][:, :, 0]
        mask = (torch.sigmoid(raw) > self.cfg.RPN.SCORE_THRESH).float()
        rois, roi_raw = self.rpn.proposal_layer(raw, first["rpn_reg"], xyz)
        extras = {"rois": rois, "roi_scores_raw": roi_raw, "seg_result": mask}
        feed = {"rpn_xyz": xyz, "rpn_features": first["backbone_features"].permute((0, 2, 1)), "seg_mask": mask,
                "roi_boxes3d": rois, "pts_depth": torch.norm(xyz, p=2, dim=2)}
        return extras, feed

    def forward(self, input_data):
        cfg = self.cfg
        if not cfg.RPN.ENABLED:                      # RCNN-only: the caller supplies pooled inputs
            if not cfg.RCNN.ENABLED:
                raise NotImplementedError
            return self.rcnn_net(input_data)
        result = dict(self._first_stage(input_data))
        if cfg.RCNN.ENABLED:
            extras, feed = self._second_stage_inputs(result)
            result.update(extras)
            if self.training:
                feed["gt_boxes3d"] = input_data["gt_boxes3d"]
            result.update(self.rcnn_net(feed))
        return result
