"""RPN proposal generation (counterpart of pointrcnn/lib/rpn/proposal_layer.py:9-145).

Same result as the reference -- decode every point's box, sort by score, split into the
(0,40] m and (40,80] m depth bands, take the top 70 % / 30 % of RPN_PRE_NMS_TOP_N per band,
greedy NMS, keep the first 70 % / 30 % of RPN_POST_NMS_TOP_N, zero-pad -- but BATCHED and without
a single host synchronisation: the reference loops over scenes in Python with two device syncs
(`dist_mask.sum() != 0`, :83) and two NMS round trips per scene; here the band selection is a
masked cumulative rank + scatter into padded per-(scene, band) tables, all 2*B NMS problems run
in one launch of the device-resident NMS, and the 100 RoIs are gathered with index arithmetic.
"""
import torch
import torch.nn as nn

from ..bbox_transform import decode_bbox_target
from .. import kitti_utils
from .. import iou3d_utils
from .._lib import has_entry


class ProposalLayer(nn.Module):
    def __init__(self, cfg, mode="TRAIN"):
        super().__init__()
        self.cfg = cfg
        self.mode = mode
        self.register_buffer("MEAN_SIZE", torch.from_numpy(cfg.CLS_MEAN_SIZE[0]).float().clone(), persistent=False)
        self._anchor = [float(v) for v in torch.from_numpy(cfg.CLS_MEAN_SIZE[0]).float()]   # f32-rounded h, w, l
        self.fused = True      # use the fused HIP proposal path when the extension offers it

    def forward(self, rpn_scores, rpn_reg, xyz):
        """rpn_scores (B,N), rpn_reg (B,N,C), xyz (B,N,3) -> rois (B,M,7), roi_scores_raw (B,M)."""
        cfg = self.cfg
        B, N = rpn_scores.shape
        ext = iou3d_utils.iou3d_cuda
        # (the reference reads the switch from cfg.TEST whatever the mode, proposal_layer.py:40)
        M = cfg[self.mode].RPN_POST_NMS_TOP_N
        if (self.fused and cfg.TEST.RPN_DISTANCE_BASED_PROPOSE and N <= 65536 and M <= 128
                and cfg.RPN.NMS_TYPE in ("normal", "rotate") and has_entry(ext, "rpn_proposals")):
            # one extension call: decode, sort, band selection, NMS and assembly as HIP kernels
            rois = torch.empty((B, M, 7), dtype=torch.float32, device=xyz.device)
            roi_scores = torch.empty((B, M), dtype=torch.float32, device=xyz.device)
            ext.rpn_proposals(xyz.contiguous(), rpn_scores.contiguous(), rpn_reg.contiguous(), self._anchor,
                              cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE, cfg.RPN.NUM_HEAD_BIN, cfg.RPN.LOC_XZ_FINE,
                              cfg[self.mode].RPN_PRE_NMS_TOP_N, M, cfg[self.mode].RPN_NMS_THRESH,
                              cfg.RPN.NMS_TYPE == "rotate", rois, roi_scores)
            return rois, roi_scores
        proposals = decode_bbox_target(xyz.view(-1, 3), rpn_reg.view(-1, rpn_reg.shape[-1]),
                                       anchor_size=self.MEAN_SIZE, loc_scope=cfg.RPN.LOC_SCOPE,
                                       loc_bin_size=cfg.RPN.LOC_BIN_SIZE, num_head_bin=cfg.RPN.NUM_HEAD_BIN,
                                       get_xz_fine=cfg.RPN.LOC_XZ_FINE, get_y_by_bin=False, get_ry_fine=False)
        proposals[:, 1] += proposals[:, 3] / 2          # y becomes the bottom centre
        proposals = proposals.view(B, N, 7)
        if not hasattr(ext, "nms_device"):
            # an extension module with the reference's four entry points only (the compiled dropin_native/iou3d_cuda): the NMS is
            # the blocking nms_gpu / nms_normal_gpu, one call per scene and distance band, as in proposal_layer.py:40-119
            return self._per_scene_blocking(rpn_scores, proposals)
        if not cfg.TEST.RPN_DISTANCE_BASED_PROPOSE:
            return self._score_based(rpn_scores, proposals)
        return self._distance_based(rpn_scores, proposals)

    def _per_scene_blocking(self, scores, proposals):
        """The proposal layer over the reference's blocking NMS API: scene by scene, scores sorted, (distance-based) the near
        band (0, 40] m and the far band (40, 80] m cut to 70 % / 30 % of RPN_PRE_NMS_TOP_N, NMS per band, 70 % / 30 % of
        RPN_POST_NMS_TOP_N kept; a scene without far points takes the next near candidates instead.  Two host round trips per
        band (the keep list comes back through host memory) -- the cost the batched device path above avoids."""
        cfg, mode = self.cfg, self.cfg[self.mode]
        B = scores.shape[0]
        post_tot = mode.RPN_POST_NMS_TOP_N
        rois = scores.new_zeros((B, post_tot, 7))
        roi_scores = scores.new_zeros((B, post_tot))
        nms = iou3d_utils.nms_gpu if cfg.RPN.NMS_TYPE == "rotate" else iou3d_utils.nms_normal_gpu
        if cfg.RPN.NMS_TYPE not in ("rotate", "normal"):
            raise NotImplementedError(cfg.RPN.NMS_TYPE)
        for b in range(B):
            s_sorted, order = torch.sort(scores[b], descending=True)
            p_sorted = proposals[b][order]
            kept_s, kept_p = [], []
            if cfg.TEST.RPN_DISTANCE_BASED_PROPOSE:
                pre = [int(mode.RPN_PRE_NMS_TOP_N * 0.7)]
                pre.append(mode.RPN_PRE_NMS_TOP_N - pre[0])
                post = [int(post_tot * 0.7)]
                post.append(post_tot - post[0])
                dist = p_sorted[:, 2]
                near = (dist > 0.0) & (dist <= 40.0)
                for band, lo, hi in ((0, 0.0, 40.0), (1, 40.0, 80.0)):
                    m = (dist > lo) & (dist <= hi)
                    if int(m.sum()) != 0:
                        cs, cp = s_sorted[m][:pre[band]], p_sorted[m][:pre[band]]
                    else:                                       # no point that far: the near candidates behind the first cut
                        cs, cp = s_sorted[near][pre[0]:][:pre[band]], p_sorted[near][pre[0]:][:pre[band]]
                    keep = nms(kitti_utils.boxes3d_to_bev_torch(cp), cs, mode.RPN_NMS_THRESH)[:post[band]]
                    kept_s.append(cs[keep]); kept_p.append(cp[keep])
            else:
                cs, cp = s_sorted[:mode.RPN_PRE_NMS_TOP_N], p_sorted[:mode.RPN_PRE_NMS_TOP_N]
                keep = iou3d_utils.nms_gpu(kitti_utils.boxes3d_to_bev_torch(cp), cs, mode.RPN_NMS_THRESH)[:post_tot]
                kept_s.append(cs[keep]); kept_p.append(cp[keep])
            ks, kp = torch.cat(kept_s), torch.cat(kept_p)
            rois[b, :kp.shape[0]] = kp
            roi_scores[b, :ks.shape[0]] = ks
        return rois, roi_scores

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _compact(mask, rank_lo, rank_hi, src, table_rows):
        """Rows of src (B,N,C) with mask set and rank in [rank_lo, rank_hi) (rank = order of
        appearance among masked rows) are written, in order, to a (B, table_rows, C) table."""
        rank = torch.cumsum(mask.to(torch.int32), dim=1) - 1
        sel = mask & (rank >= rank_lo) & (rank < rank_hi)
        slot = torch.where(sel, rank - rank_lo, torch.full_like(rank, table_rows)).long()
        table = src.new_zeros((src.shape[0], table_rows + 1, src.shape[2]))   # last row = discard bin
        table.scatter_(1, slot.unsqueeze(-1).expand(-1, -1, src.shape[2]), src)
        count = sel.sum(dim=1).to(torch.int32)
        return table[:, :table_rows], count

    def _distance_based(self, scores, proposals):
        cfg = self.cfg[self.mode]
        B, N = scores.shape
        pre_tot, post_tot = cfg.RPN_PRE_NMS_TOP_N, cfg.RPN_POST_NMS_TOP_N
        pre = [int(pre_tot * 0.7), pre_tot - int(pre_tot * 0.7)]
        post = [int(post_tot * 0.7), post_tot - int(post_tot * 0.7)]

        sorted_scores, order = torch.sort(scores, dim=1, descending=True)
        ordered = torch.gather(proposals, 1, order.unsqueeze(-1).expand(-1, -1, 7))
        payload = torch.cat([ordered, sorted_scores.unsqueeze(-1)], dim=2)      # (B,N,8)
        dist = ordered[:, :, 2]
        near = (dist > 0) & (dist <= 40.0)
        far = (dist > 40.0) & (dist <= 80.0)
        far_empty = (far.sum(dim=1, keepdim=True) == 0)

        rows = pre[0]
        near_tab, near_cnt = self._compact(near, 0, pre[0], payload, rows)
        far_tab, far_cnt = self._compact(far, 0, pre[1], payload, rows)
        # a scene with no far points re-uses the NEXT pre[1] near proposals (proposal_layer.py:92-99)
        spill_tab, spill_cnt = self._compact(near, pre[0], pre[0] + pre[1], payload, rows)
        far_tab = torch.where(far_empty.unsqueeze(-1), spill_tab, far_tab)
        far_cnt = torch.where(far_empty.squeeze(1), spill_cnt, far_cnt)

        tabs = torch.stack([near_tab, far_tab], dim=1).view(2 * B, rows, 8)     # problem p = 2*b + band
        counts = torch.stack([near_cnt, far_cnt], dim=1).view(2 * B).contiguous()
        bev = kitti_utils.boxes3d_to_bev_torch(tabs.view(-1, 8)[:, :7]).view(2 * B, rows, 5)
        rotated = self.cfg.RPN.NMS_TYPE == "rotate"
        if not rotated and self.cfg.RPN.NMS_TYPE != "normal":
            raise NotImplementedError(self.cfg.RPN.NMS_TYPE)
        keep, num = iou3d_utils.nms_device_batched(bev, counts, cfg.RPN_NMS_THRESH, rotated, post[0])
        keep = keep.view(B, 2, post[0]).long()
        num = num.view(B, 2).long()
        k_near = num[:, 0].clamp(max=post[0])
        k_far = num[:, 1].clamp(max=post[1])

        # slot j of a scene: near keep j for j < k_near, then far keep j - k_near, then zeros
        j = torch.arange(post_tot, device=scores.device).unsqueeze(0).expand(B, -1)
        from_near = j < k_near.unsqueeze(1)
        from_far = (~from_near) & (j < (k_near + k_far).unsqueeze(1))
        jn = j.clamp(max=post[0] - 1)
        jf = (j - k_near.unsqueeze(1)).clamp(min=0, max=post[0] - 1)
        row_near = torch.gather(keep[:, 0], 1, jn).clamp(min=0)
        row_far = torch.gather(keep[:, 1], 1, jf).clamp(min=0)
        tabs = tabs.view(B, 2, rows, 8)
        pick_near = torch.gather(tabs[:, 0], 1, row_near.unsqueeze(-1).expand(-1, -1, 8))
        pick_far = torch.gather(tabs[:, 1], 1, row_far.unsqueeze(-1).expand(-1, -1, 8))
        out = torch.where(from_near.unsqueeze(-1), pick_near,
                          torch.where(from_far.unsqueeze(-1), pick_far, torch.zeros_like(pick_near)))
        return out[:, :, :7].contiguous(), out[:, :, 7].contiguous()

    def _score_based(self, scores, proposals):
        cfg = self.cfg[self.mode]
        B, N = scores.shape
        pre, post = min(cfg.RPN_PRE_NMS_TOP_N, N), cfg.RPN_POST_NMS_TOP_N
        sorted_scores, order = torch.sort(scores, dim=1, descending=True)
        ordered = torch.gather(proposals, 1, order.unsqueeze(-1).expand(-1, -1, 7))[:, :pre]
        bev = kitti_utils.boxes3d_to_bev_torch(ordered.reshape(-1, 7)).view(B, pre, 5)
        keep, num = iou3d_utils.nms_device_batched(bev, None, cfg.RPN_NMS_THRESH, True, post)  # nms_gpu (:134)
        valid = torch.arange(post, device=scores.device).unsqueeze(0) < num.long().unsqueeze(1)
        rows = keep.long().clamp(min=0)
        boxes = torch.gather(ordered, 1, rows.unsqueeze(-1).expand(-1, -1, 7)) * valid.unsqueeze(-1)
        sc = torch.gather(sorted_scores[:, :pre], 1, rows) * valid
        return boxes.contiguous(), sc.contiguous()
