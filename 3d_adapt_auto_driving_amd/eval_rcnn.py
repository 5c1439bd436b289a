"""Joint RPN+RCNN evaluation harness: the counterpart of the hot loop of
pointrcnn/tools/eval_rcnn.py (eval_one_epoch_joint :466-690, save_kitti_format :76-101,
checkpoint loading via train_utils.load_checkpoint :78-92) for synthetic KITTI-shaped scenes.

Differences that are deliberate:
  * the per-scene tail (score threshold -> sort -> rotated NMS -> D2H, :611-635) is batched on
    the device: one masked sort, one launch of the device-resident NMS for all scenes, ONE D2H
    copy per batch of fixed-shape (B,M,7)/(B,M)/(B) tensors;
  * scenes shard across ranks (rank r takes scenes r, r+W, ...) and detections are combined with
    one all_gather at the end (the reference is single-process);
  * the reference's dataloader bug (`far_points=` keyword, eval_rcnn.py:862) is not reproduced.

CLI (subset of the reference's flags):  python -m ... --cfg_file X --eval_mode rcnn --ckpt Y
  --batch_size 8 --output_dir out [--set K V ...] [--scenes 64]
"""
import argparse
import collections
import contextlib
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import config as config_mod
from . import kitti_utils
from . import iou3d_utils
from . import synth
from .bbox_transform import decode_bbox_target
from .net.point_rcnn import PointRCNN
from .net.fast_infer import FastPointRCNN
from ._lib import has_entry

# PRCNN_NO_RCNN_SPLIT=1: the RCNN stage's RoI pooling / sampling / grouping geometry stays on the feature stream (A/B switch)
SPLIT_RCNN = os.environ.get("PRCNN_NO_RCNN_SPLIT") != "1"
# (removed in round 4, results in DESIGN.md section 7: PRCNN_GEO_THREAD -- geometry chains enqueued by a helper thread, slower: the GIL;
#  PRCNN_RCNN_GEO_STREAM -- the RCNN's geometry on a stream of its own, slower: a 5th busy stream shares a hardware queue; PRCNN_GATE --
#  chains only started at the end of an RPN stage, slower since the stages are our own ticketed kernels; PRCNN_STREAM_SKEW)


def build_model(cfg, device, seed=0):
    """Random-init PointRCNN in TEST mode (eval_rcnn.py:758), deterministic in ``seed``."""
    torch.manual_seed(seed)
    model = PointRCNN(cfg, num_classes=2, use_xyz=True, mode="TEST")
    return model.to(device).eval()


def load_checkpoint(model, filename, logger=None):
    """Accepts the reference's .pth layout {'model_state','optimizer_state','epoch','it'}
    (train_utils.py:60-92); also a bare state_dict."""
    ckpt = torch.load(filename, map_location="cpu")
    state = ckpt["model_state"] if isinstance(ckpt, dict) and "model_state" in ckpt else ckpt
    model.load_state_dict(state)
    return ckpt.get("epoch", -1) if isinstance(ckpt, dict) else -1


_ANCHORS = {}
FUSED_POSTPROCESS = True     # final stage through the fused HIP entry when the extension offers it
# Round 4: the final stage (decode + threshold + rotated NMS, 0.2 ms) runs BEHIND the RCNN features on the feature stream instead of
# on the proposal stream: with the FP modules on the geometry streams (fast_infer.EARLY_FP = 3) the proposal stream -- proposal layer,
# RoI pooling, the RoI clouds' geometry, final stage -- was the longest of the four, and the hop feature -> proposal -> host costs two
# event waits.  Round 5, second session: back on the proposal stream by default -- with the pack launches, half of the RoI geometry and the
# padded tiles gone that stream is busy 0.35-0.40 ms of a 0.97-ms step and the feature stream is the fullest (0.78): K = 20 7253 / 7238 /
# 7226 -> 7323 / 7261 / 7256 scenes/s, LiDAR-shaped 5109 / 5134 / 5143 -> 5183 / 5175 / 5180 (three alternating runs of five windows each);
# K = 100 unchanged.  PRCNN_FINAL_ON_FEATURE=1: behind the RCNN features on the feature stream (round 4).
FINAL_ON_FEATURE = os.environ.get("PRCNN_FINAL_ON_FEATURE", "0") != "0"
# batches of a geometry group that share the launches of the stages behind the geometry in the graphed runner (GraphedRunner.pair):
# 2 (measured: 6040-6150 / 6980-7050 scenes/s at K = 20 / 100 against 5800-5860 / 6730-6750 with 1, LiDAR-shaped 4270-4350 / 4930-4970
# against 4140-4160 / 4830; 4: 6060 / 6780 and 4330 / 4750); 1: every batch its own launches
RCNN_PAIR = int(os.environ.get("PRCNN_PAIR", "2"))


def _anchor_host(cfg):
    """CLS_MEAN_SIZE rounded to f32, as python floats (h, w, l)."""
    import numpy as np
    return [float(v) for v in np.asarray(cfg.CLS_MEAN_SIZE[0], dtype=np.float32)]


def _anchor_on(cfg, device):
    """CLS_MEAN_SIZE as a device tensor, uploaded once (a per-call H2D copy synchronises the stream)."""
    key = (str(device), tuple(float(v) for v in cfg.CLS_MEAN_SIZE[0]))
    if key not in _ANCHORS:
        _ANCHORS[key] = torch.tensor(key[1], dtype=torch.float32, device=device)
    return _ANCHORS[key]


def split_detections(blob, batch_size, M):
    """blob (batch_size * (8 M + 1)) f32, device or host -> views boxes (B,M,7) f32, scores (B,M) f32, num (B) i32 (the same bytes)."""
    nb, ns = batch_size * M * 7, batch_size * M
    return (blob[:nb].view(batch_size, M, 7), blob[nb:nb + ns].view(batch_size, M), blob[nb + ns:nb + ns + batch_size].view(torch.int32))


@torch.no_grad()
def postprocess(cfg, ret_dict, batch_size, blob_scenes=None):
    """Final box decoding + score threshold + rotated NMS, batched (eval_rcnn.py:506-530,611-629).
    Returns boxes (B,M,7), raw scores (B,M) and num (B) i32 on the device; rows >= num are zero.
    blob_scenes (the graphed runner's launches over several batches): the results as one blob per `blob_scenes` scenes --
    "blobs" (B / blob_scenes, blob_scenes (8 M + 1)); boxes / scores / num are then LISTS of per-blob views."""
    R = cfg.RCNN
    rois = ret_dict["rois"]
    M = rois.shape[1]
    rcnn_cls = ret_dict["rcnn_cls"].view(batch_size, M, -1)
    rcnn_reg = ret_dict["rcnn_reg"].view(batch_size, M, -1)
    if rcnn_cls.shape[2] != 1:
        raise NotImplementedError("multi-class RCNN head")
    raw = rcnn_cls[:, :, 0]
    ext = iou3d_utils.iou3d_cuda
    if FUSED_POSTPROCESS and M <= 128 and has_entry(ext, "rcnn_postprocess"):
        # one extension call (three launches): decode, threshold, score sort, rotated NMS, assembly
        dev = rois.device
        pred = torch.empty((batch_size, M, 7), dtype=torch.float32, device=dev)
        # the three results in ONE allocation ("blob": boxes | scores | num, see split_detections): the consumer brings them to the host
        # with one copy instead of three
        raw = raw.contiguous()
        if blob_scenes and blob_scenes < batch_size and batch_size % blob_scenes == 0 and has_entry(ext, "rcnn_postprocess_blobs") and R.NMS_THRESH >= 0:
            nb = batch_size // blob_scenes
            blobs = torch.empty((nb, blob_scenes * (M * 8 + 1)), dtype=torch.float32, device=dev)
            ext.rcnn_postprocess_blobs(rois.contiguous(), rcnn_reg.contiguous(), raw, _anchor_host(cfg), R.LOC_SCOPE, R.LOC_BIN_SIZE,
                                       R.NUM_HEAD_BIN, R.LOC_Y_BY_BIN, R.LOC_Y_SCOPE, R.LOC_Y_BIN_SIZE, R.SCORE_THRESH, R.NMS_THRESH,
                                       pred, blobs, blob_scenes)
            parts = [split_detections(blobs[i], blob_scenes, M) for i in range(nb)]
            return {"boxes": [p[0] for p in parts], "scores": [p[1] for p in parts], "num": [p[2] for p in parts], "pred_boxes3d": pred,
                    "raw_scores": raw, "blobs": blobs}
        blob = torch.empty((batch_size * (M * 8 + 1),), dtype=torch.float32, device=dev)
        boxes, scores, num = split_detections(blob, batch_size, M)
        ext.rcnn_postprocess(rois.contiguous(), rcnn_reg.contiguous(), raw, _anchor_host(cfg), R.LOC_SCOPE,
                             R.LOC_BIN_SIZE, R.NUM_HEAD_BIN, R.LOC_Y_BY_BIN, R.LOC_Y_SCOPE, R.LOC_Y_BIN_SIZE,
                             R.SCORE_THRESH, R.NMS_THRESH, pred, boxes, scores, num)
        return {"boxes": boxes, "scores": scores, "num": num, "pred_boxes3d": pred, "raw_scores": raw, "blob": blob}
    anchor = _anchor_on(cfg, rois.device)
    pred = decode_bbox_target(rois.view(-1, 7), rcnn_reg.view(-1, rcnn_reg.shape[-1]), anchor_size=anchor,
                              loc_scope=R.LOC_SCOPE, loc_bin_size=R.LOC_BIN_SIZE, num_head_bin=R.NUM_HEAD_BIN,
                              get_xz_fine=True, get_y_by_bin=R.LOC_Y_BY_BIN, loc_y_scope=R.LOC_Y_SCOPE,
                              loc_y_bin_size=R.LOC_Y_BIN_SIZE, get_ry_fine=True).view(batch_size, M, 7)
    selected = torch.sigmoid(raw) > R.SCORE_THRESH
    if not hasattr(ext, "nms_device"):
        # the reference's four entry points only (compiled dropin_native/iou3d_cuda): scene by scene over the blocking nms_gpu,
        # as eval_rcnn.py:611-629
        boxes, scores = torch.zeros_like(pred), torch.zeros_like(raw)
        num = torch.zeros((batch_size,), dtype=torch.int32, device=raw.device)
        for k in range(batch_size):
            cur = selected[k]
            if int(cur.sum()) == 0:
                continue
            b_sel, s_sel = pred[k][cur], raw[k][cur]
            keep = iou3d_utils.nms_gpu(kitti_utils.boxes3d_to_bev_torch(b_sel), s_sel, R.NMS_THRESH)
            boxes[k, :keep.numel()] = b_sel[keep]
            scores[k, :keep.numel()] = s_sel[keep]
            num[k] = keep.numel()
        return {"boxes": boxes, "scores": scores, "num": num, "pred_boxes3d": pred, "raw_scores": raw}
    key = torch.where(selected, raw, torch.full_like(raw, float("-inf")))
    _, order = torch.sort(key, dim=1, descending=True)          # selected boxes first, by raw score
    counts = selected.sum(dim=1).to(torch.int32)
    boxes_sorted = torch.gather(pred, 1, order.unsqueeze(-1).expand(-1, -1, 7))
    scores_sorted = torch.gather(raw, 1, order)
    bev = kitti_utils.boxes3d_to_bev_torch(boxes_sorted.reshape(-1, 7)).view(batch_size, M, 5)
    keep, num = iou3d_utils.nms_device_batched(bev, counts, R.NMS_THRESH, True, M)
    valid = torch.arange(M, device=raw.device).unsqueeze(0) < num.long().unsqueeze(1)
    rows = keep.long().clamp(min=0)
    boxes = torch.gather(boxes_sorted, 1, rows.unsqueeze(-1).expand(-1, -1, 7)) * valid.unsqueeze(-1)
    scores = torch.gather(scores_sorted, 1, rows) * valid
    return {"boxes": boxes, "scores": scores, "num": num, "pred_boxes3d": pred, "raw_scores": raw}


@torch.no_grad()
def infer_batch(model, cfg, pts_input, engine=None, geo=None):
    """pts_input (B,N,3) device f32 -> detections dict (all device tensors, fixed shapes).
    ``engine`` = a FastPointRCNN built from ``model`` (point-major fused path); without it the
    nn.Module graph (reference operation order) runs."""
    ret = engine(pts_input, geo, want_reg=False) if engine is not None else model({"pts_input": pts_input})
    det = postprocess(cfg, ret, pts_input.shape[0])
    det["rois"] = ret["rois"]
    det["rcnn_reg"] = ret["rcnn_reg"]
    det["rcnn_cls"] = ret["rcnn_cls"]
    return det


_RUNNER_STREAMS = {}


def _runner_streams(device, n_sides, prio):
    """The side / tail streams of the pipelined runner, created ONCE per (device, priority) and shared by every runner of the
    process.  HIP multiplexes its streams onto a few hardware queues (4 by default); two streams that land on the same queue
    serialise -- a feature-stream kernel then waits behind a 6 ms FPS kernel of a geometry chain.  A fresh set of streams
    per runner drew a new mapping every time (same process: 77 ms or 89 ms for the same 20 steps); with one fixed set the
    first-created streams keep the queues they were given at start-up."""
    key = (str(device), prio)
    have = _RUNNER_STREAMS.setdefault(key, {"tail": None, "sides": [], "feat": None})
    if have["tail"] is None:
        have["tail"] = torch.cuda.Stream(device)
    while len(have["sides"]) < n_sides:
        have["sides"].append(torch.cuda.Stream(device, priority=prio))
    return have["tail"], have["sides"][:n_sides]


class PipelinedRunner:
    """Software pipeline over batches on HIP streams: the xyz-only GEOMETRY of upcoming batches (FPS,
    ball query, three-NN: a latency-bound chain that keeps only B CUs busy) runs on side streams while
    the FEATURE pass of the current batch (MFMA kernels, GEMMs, pooling, NMS) fills the rest of the chip
    on the main stream.  ``depth`` batches of geometry are kept in flight (one side stream each).
    Call ``step(cur, upcoming)`` once per batch; ``upcoming`` = the next batch, or a list of the next
    ``depth`` batches, or None at the end."""

    def __init__(self, model, cfg, device, depth=None):
        self.model, self.cfg = model, cfg
        self.engine = FastPointRCNN(model, cfg)
        self.device = torch.device(device)
        # geometry runs for GROUPS of `group` batches in one chain (FastPointRCNN.geometry_group): a scene's FPS is serial
        # (~6.8 ms for 16384 -> 4096 on one CU) whatever the batch, so one chain over 3 batches costs the latency of one and
        # a single chain in flight keeps up with the feature stream.  `depth` = how many batches ahead the caller should
        # hand over (3 * group: a chain is launched 2 * `group` steps before its first batch is due; with 2 * group the chain -- 7 ms alone,
        # 11 ms beside the feature stream -- was just late for the first batch of every group: +0.5 ms once per group).
        self.group = max(1, int(os.environ.get("PRCNN_GEO_GROUP", "4")))
        self.depth = self.default_depth() if depth is None else depth
        # high priority: the geometry kernels are few, short-lived workgroups on a latency-bound chain;
        # when CU slots free up they should be placed before the feature pass's next workgroups
        # default priority: with the SA levels on the packed MFMA kernels the feature pass is short, and high-priority side
        # streams (three of them at depth 3) starve it -- measured 971 vs 1375 scenes/s
        prio = 0           # (default priority: high-priority side streams starve the feature pass; the switch PRCNN_SIDE_PRIORITY is gone, round 6)
        self._shared_tail, self.sides = _runner_streams(self.device, int(os.environ.get("PRCNN_SIDE_STREAMS", "2")) if self.group > 1 else max(1, self.depth), prio)
        self._next_side = 0
        self._pending = []        # [(batch tensor, geometry dict, ready event)] in launch order

    @staticmethod
    def default_depth():
        """Batches of look-ahead a caller should hand over (PRCNN_GEO_DEPTH; 3 x PRCNN_GEO_GROUP by default)."""
        group = max(1, int(os.environ.get("PRCNN_GEO_GROUP", "4")))
        return int(os.environ.get("PRCNN_GEO_DEPTH", str(3 * group if group > 1 else 3)))

    @property
    def side(self):
        return self.sides[0]

    def _launch_geometry(self, pts):
        main = torch.cuda.current_stream(self.device)
        side = self.sides[self._next_side % len(self.sides)]
        self._next_side += 1
        side.wait_stream(main)                        # pts (and the allocator's frees) are ordered before us
        with torch.cuda.stream(side):
            geo = self.engine.geometry(pts)
            ev = torch.cuda.Event()
            ev.record(side)
        for t in _tensors(geo):                       # consumed on the main stream: tell the caching allocator
            t.record_stream(main)
        self._pending.append((pts, geo, ev))

    def _take(self, pts):
        for i, (p, geo, ev) in enumerate(self._pending):
            if p is pts:
                del self._pending[i]
                return geo, ev
        self._launch_geometry(pts)
        return self._take(pts)

    def _prefetch_geometry(self, upcoming):
        if upcoming is not None:
            todo = list(upcoming) if isinstance(upcoming, (list, tuple)) else [upcoming]
            for nxt in todo[:max(1, self.depth)]:
                if nxt is not None and all(p is not nxt for p, _, _ in self._pending):
                    self._launch_geometry(nxt)        # enqueue first: it overlaps the feature pass below

    @torch.no_grad()
    def step(self, cur, upcoming=None):
        geo, ev = self._take(cur)
        self._prefetch_geometry(upcoming)
        torch.cuda.current_stream(self.device).wait_event(ev)
        return infer_batch(self.model, self.cfg, cur, engine=self.engine, geo=geo)

    # ---- three-stream form: the latency-bound tails leave the GEMM stream -------------------------
    # The proposal layer (per-scene score sort, band selection, NMS: ONE workgroup per scene) and the final
    # stage (decode + NMS) keep only B CUs busy.  On the feature stream they would stall the chip-filling
    # kernels behind them, so they run on a third stream and the feature stream is software-pipelined:
    #     feature stream :  RPN(i)   RCNN(i-1)   RPN(i+1)   RCNN(i)   ...
    #     tail stream    :        proposals(i)  final(i-1)      proposals(i+1)  final(i) ...
    #     geometry stream:  geometry(i+1)                geometry(i+2)
    # Exactly one stream carries library GEMMs (two GEMM streams deadlock).
    # ``submit`` returns the detections of the PREVIOUS batch (None on the first call), ``flush`` the last one;
    # the returned tensors are produced on ``self.tail``: read them under that stream or after ``det["ready"]``.
    @torch.no_grad()
    def submit(self, cur, upcoming=None):
        if getattr(self, "tail", None) is None:
            self.tail = self._shared_tail
            self._inflight = None
            self._chains = []                 # geometry chains in flight: dicts pts / side / state / geo / ev
            self._retired = []                # (side stream, RPN-done event, geometry) of batches whose geometry is still kept
        main = torch.cuda.current_stream(self.device)
        todo = [] if upcoming is None else (list(upcoming) if isinstance(upcoming, (list, tuple)) else [upcoming])
        todo = [p for p in todo if p is not None][:max(1, self.depth)]
        if self.group > 1:
            return self._submit_grouped(cur, todo, main)
        ch = self._chain(cur)
        if ch is None:                        # cold start: nothing was prefetched for this batch
            ch = self._chain_begin(cur, None)
        if ch["geo"] is None:
            self._chain_finish(ch, None)
        self._chains = [c for c in self._chains if c is not ch]      # by identity (dict equality would compare tensors)
        self._advance_chains(todo, None)      # geometry of the upcoming batches starts right away
        main.wait_event(ch["ev"])
        st = self.engine.rpn_stage(cur, ch["geo"])
        ev_rpn = torch.cuda.Event()
        ev_rpn.record(main)
        for t in (st["rpn_scores_raw"], st["rpn_reg"], st.get("rpn_boxes"), st["backbone_xyz"]):
            if t is not None:
                t.record_stream(self.tail)
        rois, roi_scores, ev_prop, rg = self._propose_on_tail(st, ev_rpn, main)
        done = self._finish_inflight()
        self._inflight = (cur, st, rois, roi_scores, ev_prop, None, None, rg)
        return done

    def _propose_on_tail(self, st, ev_rpn, main):
        """Proposal layer of the batch whose RPN stage ends at `ev_rpn`, on the tail stream -- and right behind it the part of
        its RCNN stage that needs the RoIs and coordinates only (RoI pooling, FPS, ball queries, row lists: ten latency-bound
        launches, 0.3 ms of every step while they sat on the feature stream in front of the RCNN's MFMA kernels).
        -> rois, scores, event (RoIs and, if split, the RCNN geometry are ready), RCNN geometry state or None."""
        rg = None
        with torch.cuda.stream(self.tail):
            self.tail.wait_event(ev_rpn)
            rois, roi_scores = self.engine.propose(st)
            if SPLIT_RCNN:
                # reads st["rpn_features"] (feature-stream memory: st stays referenced until the feature stream has run this
                # batch's RCNN stage, which waits for ev_prop) and makes ~15 tensors on the tail stream that the feature stream
                # reads: they are kept in self._inflight until the tail stream has waited for that RCNN stage (_finish_inflight)
                rg = self.engine.rcnn_geometry(st, rois)
            ev_prop = torch.cuda.Event()
            ev_prop.record(self.tail)
        if rg is None:
            for t in (rois, st["seg_result"], st["pts_depth"], st["depth_norm"]):   # made on the tail stream, read by the RCNN stage on the feature stream
                t.record_stream(main)
        return rois, roi_scores, ev_prop, rg

    # ---- grouped geometry: ONE chain per `group` batches -------------------------------------------------------
    def _launch_group(self, batch_list, urgent=False):
        main = torch.cuda.current_stream(self.device)
        side = self.sides[self._next_side % len(self.sides)]
        self._next_side += 1
        side.wait_stream(main)                        # the batches (and the allocator's frees) are ordered before the chain
        # Geometry tensors are allocated on a side stream and read by the RPN stage on the feature stream.  record_stream()
        # would make that safe, but the caching allocator then records one event ON THE FEATURE STREAM per tensor when the
        # batch's ~40 tensors are freed: 40 marker packets = 0.2 ms of feature-stream time per step (profiles/gap_probe.py,
        # HIP API trace).  Instead a batch's tensors are KEPT (self._retired) until the side stream that owns their memory has
        # been made to wait for the RPN stage that read them -- here, before that stream allocates again.
        mine = [r for r in self._retired if r[0] is side]
        self._retired = [r for r in self._retired if r[0] is not side]
        for _, ev_read, _ in mine:
            side.wait_event(ev_read)
        del mine
        entries = [{"pts": pts, "geo": None, "ev": None, "side": side, "future": None} for pts in batch_list]

        def enqueue():
            evs = []

            def mark(_):                              # one event per batch: its RPN stage need not wait for the rest of the group
                e = torch.cuda.Event()
                e.record(side)
                evs.append(e)
            with torch.cuda.device(self.device), torch.cuda.stream(side):
                # urgent (cold start: the first batch of this group is waited for right now): SA levels batch by batch, so that batch 0
                # is ready before the other three are computed; otherwise over the group's clouds at once (a quarter of the launches)
                geos = self.engine.geometry_group(batch_list, on_batch_done=mark, group_sa=False if urgent else None)
                if len(evs) != len(geos):
                    ev = torch.cuda.Event()
                    ev.record(side)
                    evs = [ev] * len(geos)
            for c, geo, ev in zip(entries, geos, evs):
                c["geo"], c["ev"] = geo, ev

        enqueue()
        self._chains.extend(entries)

    @staticmethod
    def _chain_ready(ch):
        return ch

    def _submit_grouped(self, cur, todo, main):
        ch = self._chain(cur)
        if ch is None:                                # cold start (or a caller that looks less far ahead): chain for what is known
            self._launch_group([cur] + [p for p in todo if self._chain(p) is None][:self.group - 1], urgent=True)
            ch = self._chain(cur)
        self._chains = [c for c in self._chains if c is not ch]
        self._chain_ready(ch)
        # start the next group as soon as a whole group of upcoming batches has no chain yet (with a look-ahead of
        # 2 * group that is `group` steps before its first batch is due), or when the look-ahead is about to run dry
        missing = [p for p in todo if self._chain(p) is None]
        have = len(todo) - len(missing)
        if missing and (len(missing) >= self.group or have <= 1):
            self._launch_group(missing[:self.group])
        main.wait_event(ch["ev"])
        st = self.engine.rpn_stage(cur, ch["geo"])
        ev_rpn = torch.cuda.Event()
        ev_rpn.record(main)
        for t in (st["rpn_scores_raw"], st["rpn_reg"], st.get("rpn_boxes"), st["backbone_xyz"]):
            if t is not None:
                t.record_stream(self.tail)
        rois, roi_scores, ev_prop, rg = self._propose_on_tail(st, ev_rpn, main)
        done = self._finish_inflight()
        self._inflight = (cur, st, rois, roi_scores, ev_prop, ch["side"], ch["geo"], rg)
        return done

    def _advance_chains(self, todo, gate):
        for k, nxt in enumerate(todo):
            c = self._chain(nxt)
            if c is None:
                c = self._chain_begin(nxt, gate)
                if k == 0:
                    self._chain_finish(c, None)           # needed by the very next submit: same gate, same stream
            elif c["geo"] is None and k == 0:
                self._chain_finish(c, gate)

    def _chain(self, pts):
        for c in self._chains:
            if c["pts"] is pts:
                return c
        return None

    def _chain_begin(self, pts, gate):
        side = self.sides[self._next_side % len(self.sides)]
        self._next_side += 1
        if gate is None:
            side.wait_stream(torch.cuda.current_stream(self.device))     # pts is ready on the calling stream
        else:
            side.wait_event(gate)
        with torch.cuda.stream(side):
            state = self.engine.geometry_begin(pts)
        c = {"pts": pts, "side": side, "state": state, "geo": None, "ev": None}
        self._chains.append(c)
        return c

    def _chain_finish(self, c, gate):
        side = c["side"]
        if gate is not None:
            side.wait_event(gate)
        with torch.cuda.stream(side):
            c["geo"] = self.engine.geometry_finish(c["state"])
            c["ev"] = torch.cuda.Event()
            c["ev"].record(side)
        c["state"] = None
        main = torch.cuda.current_stream(self.device)
        for t in _tensors(c["geo"]):                  # consumed on the feature stream: tell the caching allocator
            t.record_stream(main)

    def _finish_inflight(self):
        if self._inflight is None:
            return None
        main = torch.cuda.current_stream(self.device)
        cur, st, rois, roi_scores, ev_prop, side, geo, rg = self._inflight
        self._inflight = None
        main.wait_event(ev_prop)
        out = self.engine.rcnn_stage(st, rois) if rg is None else self.engine.rcnn_features(rg)
        ev_rcnn = torch.cuda.Event()
        ev_rcnn.record(main)
        if side is not None:
            # the batch's geometry (read by its RPN stage, its spatial groups by this RCNN stage) retires: kept until the side
            # stream that owns the memory has been made to wait for this point (see _launch_group)
            self._retired.append((side, ev_rcnn, geo))
        post = main if FINAL_ON_FEATURE else self.tail
        if not FINAL_ON_FEATURE:
            for t in (out["rcnn_cls"], out["rcnn_reg"]):
                t.record_stream(self.tail)
        with torch.cuda.stream(post):
            if not FINAL_ON_FEATURE:
                self.tail.wait_event(ev_rcnn)
            ret = {"rois": rois, "rcnn_cls": out["rcnn_cls"], "rcnn_reg": out["rcnn_reg"]}
            det = postprocess(self.cfg, ret, cur.shape[0])
            det.update(ret)
            ready = torch.cuda.Event()
            ready.record(post)
        if FINAL_ON_FEATURE:
            rois.record_stream(main)                  # produced on the proposal stream, read by the final stage here
        del rg                                        # tail-stream memory, read on the feature stream up to ev_rcnn: the tail stream waits for it above
        det["ready"] = ready
        det["stream"] = post
        return det

    @torch.no_grad()
    def flush(self):
        """Finish the batch still in flight (RCNN + final stage) and return its detections (or None)."""
        if getattr(self, "tail", None) is None:
            return None
        det = self._finish_inflight()
        for side, ev_read, _ in self._retired:        # the kept geometry goes back to its streams' pools, ordered after its readers
            side.wait_event(ev_read)
        self._retired = []
        return det

    def drain(self):
        """flush() until the pipeline is empty: the detections of every batch not handed back yet, oldest first (probes and
        timing loops call this in front of a synchronize -- ONE flush() leaves up to two batches' RCNN + final stages unlaunched)."""
        out = []
        while True:
            det = self.flush()
            if det is None:
                return out
            out.append(det)


USE_GRAPHS = os.environ.get("PRCNN_GRAPHS", "1") != "0"                   # hipGraph replay of the stages (GraphedRunner); 0: eager enqueue (PipelinedRunner)


_GRAPH_DEBUG = int(os.environ.get("PRCNN_GRAPH_DEBUG", "0"))   # 1: device sync before a geometry graph, 2: after it (bisecting overlaps)


def engine_covers(cfg):
    """Does the point-major engine (net/fast_infer.py) cover this configuration?  Round 4: cfg.RPN.USE_INTENSITY (a 4-channel
    pts_input, rpn.py:17 / kitti_rcnn_dataset.py:321-338; the reference's code default, lib/config.py:40) runs on the engine too --
    on its general kernels, serially (EngineRunner); the stream-pipelined / graph-replayed runners are written for the (B, N, 3)
    clouds of the shipped configurations.  cfg.RCNN.USE_INTENSITY stays on the nn.Module graph (ModuleRunner)."""
    return not bool(cfg.RCNN.ENABLED and cfg.RCNN.USE_INTENSITY)


_NATIVE_MODULES = None


def native_modules():
    """The three COMPILED extension modules (dropin_native/: pybind11 over the C ABI, the reference's 9 + 4 + 4 entry points and
    nothing else), loaded once from their directory without shadowing the ctypes modules of the same names."""
    global _NATIVE_MODULES
    if _NATIVE_MODULES is None:
        import importlib.util
        from . import NATIVE_DROPIN_DIR
        mods = []
        for name in ("pointnet2_cuda", "iou3d_cuda", "roipool3d_cuda"):
            hits = [f for f in os.listdir(NATIVE_DROPIN_DIR) if f.startswith(name + ".") and f.endswith(".so")]
            if not hits:
                raise RuntimeError("%s: compiled module %s not built (python __graft_entry__.py)" % (NATIVE_DROPIN_DIR, name))
            spec = importlib.util.spec_from_file_location(name, os.path.join(NATIVE_DROPIN_DIR, hits[0]))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mods.append(mod)
        _NATIVE_MODULES = tuple(mods)
    return _NATIVE_MODULES


@contextlib.contextmanager
def reference_api_only(native=True):
    """What a user of the reference's Python gets from the drop-in modules: inside this context the nn.Module graph runs in the
    reference's operation order over the reference's 17 entry points ONLY -- FPS -> gather -> ball_query -> group (x2) ->
    subtract -> cat -> Conv2d / BatchNorm / ReLU modules -> max_pool2d (pointnet2_modules.py:19-55), three_nn -> three_interpolate
    -> cat -> Conv (:139-151), the per-scene proposal layer and final stage over the blocking nms_gpu / nms_normal_gpu, roipool3d
    forward -- no fused entry, no folded MLP, no engine.  ``native``: through the compiled dropin_native modules (default) or the
    ctypes ones."""
    from .pointnet2 import pointnet2_utils as pu, fused_mlp
    from . import roipool3d_utils as ru
    saved = (pu.pointnet2, iou3d_utils.iou3d_cuda, ru.roipool3d_cuda, pu.REFERENCE_ORDER, fused_mlp.ENABLED)
    if native:
        pu.pointnet2, iou3d_utils.iou3d_cuda, ru.roipool3d_cuda = native_modules()
    else:
        class _Only:
            def __init__(self, mod, names):
                for n in names:
                    setattr(self, n, getattr(mod, n))
        pu.pointnet2 = _Only(pu.pointnet2, ("ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper",
                                            "gather_points_wrapper", "gather_points_grad_wrapper", "furthest_point_sampling_wrapper",
                                            "three_nn_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper"))
        iou3d_utils.iou3d_cuda = _Only(iou3d_utils.iou3d_cuda, ("boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu"))
        ru.roipool3d_cuda = _Only(ru.roipool3d_cuda, ("forward", "forward_slow", "pts_in_boxes3d_cpu", "roipool3d_cpu"))
    pu.REFERENCE_ORDER, fused_mlp.ENABLED = True, False
    try:
        yield
    finally:
        pu.pointnet2, iou3d_utils.iou3d_cuda, ru.roipool3d_cuda, pu.REFERENCE_ORDER, fused_mlp.ENABLED = saved


class ModuleRunner:
    """submit() / flush() of the pipelined runners over the nn.Module graph (the reference's operation order, HIP operators through the
    drop-in modules, no side streams): for configurations the point-major engine does not cover.  Same one-batch-late protocol."""
    depth = 1

    def __init__(self, model, cfg, device, depth=None):
        self.model, self.cfg, self.device = model, cfg, torch.device(device)
        self._pending = None

    @torch.no_grad()
    def submit(self, cur, upcoming=None):
        done = self._pending
        det = self._infer(cur)
        if self.device.type == "cuda":
            stream = torch.cuda.current_stream(self.device)
            det["ready"] = torch.cuda.Event()
            det["ready"].record(stream)
            det["stream"] = stream
        self._pending = det
        return done

    def flush(self):
        done, self._pending = self._pending, None
        return done

    def drain(self):
        """flush() until the pipeline is empty: the detections of every batch not handed back yet, oldest first (probes and
        timing loops call this in front of a synchronize -- ONE flush() leaves up to two batches' RCNN + final stages unlaunched)."""
        out = []
        while True:
            det = self.flush()
            if det is None:
                return out
            out.append(det)

    def _infer(self, cur):
        return infer_batch(self.model, self.cfg, cur)


class EngineRunner(ModuleRunner):
    """The same protocol over the point-major engine, one batch after the other on the caller's stream: configurations the
    engine covers on its general kernels only (cfg.RPN.USE_INTENSITY)."""

    def __init__(self, model, cfg, device, depth=None):
        super().__init__(model, cfg, device, depth)
        self.engine = FastPointRCNN(model, cfg)

    def _infer(self, cur):
        return infer_batch(self.model, self.cfg, cur, engine=self.engine)


def make_runner(model, cfg, device, depth=None):
    """The runner of the product path: hipGraph replay unless PRCNN_GRAPHS=0 (same streams, same kernels, same results)."""
    if not engine_covers(cfg):
        return ModuleRunner(model, cfg, device, depth)
    if cfg.RPN.USE_INTENSITY:
        return EngineRunner(model, cfg, device, depth)
    from . import GRAPH_REPLAY_SAFE
    if USE_GRAPHS and not GRAPH_REPLAY_SAFE:
        import warnings
        warnings.warn("hipGraph replay disabled: the HIP runtime was initialised before DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 could be set "
                      "(import the package, or export the variable, before the first torch.cuda call); using the eager runner")
    return (GraphedRunner if USE_GRAPHS and GRAPH_REPLAY_SAFE else PipelinedRunner)(model, cfg, device, depth)


class GraphedRunner:
    """PipelinedRunner with every stage captured ONCE into a hipGraph and replayed: the same kernels with the same arguments on the
    same streams in the same order -- the host's part of a step falls from ~70 extension calls + their torch glue (0.8 ms of Python
    per step, more than half of the step's period: the host thread was co-limiting, profiles/r03_microbench.md) to four graph
    launches and a dozen event operations.

    A graph replays fixed addresses, so the pipeline runs over SLOTS instead of freshly allocated tensors:
      * a group slot holds the coordinates of `group` batches (copied in when their chain is launched: 1.5 MB per batch), the
        geometry graph of the group (FastPointRCNN.geometry_group: FPS / ball queries / row lists / three-NN / the early SA levels,
        on a side stream) and, per MEMBER of the group (`pair` consecutive batches: 2 by default), four graphs: RPN stage (feature
        stream), proposal layer + RCNN geometry (tail stream), RCNN features (feature stream), final stage (behind them on the feature
        stream; FINAL_ON_FEATURE = 0: tail stream) -- pair x B scenes per launch of each;
      * depth / group + 1 group slots rotate: a slot is rewritten only after the RCNN stages of its previous batches (an event wait
        on the side stream, normally long past);
      * submit() hands back the detections of an EARLIER batch, in submit order, or None (a member's stages are launched when its
        last batch is submitted, its RCNN + final stages behind the next member's RPN stage: up to 2 pair - 1 submits late); flush()
        one batch per call until None.  They are views into the slot: valid for (slots - 1) * group - 2 pair further submits (8 by
        default) -- copy them out (on det["stream"]) before that, as eval_scenes and bench.py do right away.
    Every graph has a memory pool of its own (see _build); everything a later graph or the caller reads is kept referenced here.  Batches of another shape than the first one seen (the last, short batch of a split) run eagerly.
    Nondeterminism is that of the eager path: the worklists built with atomics (point groups, pooled tiles) come out in
    a different order every run, the results computed from them do not (tests/test_gpu_graphs.py: detections bit for bit)."""

    def __init__(self, model, cfg, device, depth=None):
        from . import GRAPH_REPLAY_SAFE
        if not GRAPH_REPLAY_SAFE:
            raise RuntimeError("GraphedRunner: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in place when the HIP runtime started "
                               "(see the package's __init__): replaying graphs is unsafe on this runtime; use make_runner()")
        self.model, self.cfg = model, cfg
        self.engine = FastPointRCNN(model, cfg)
        self.device = torch.device(device)
        self.group = max(1, int(os.environ.get("PRCNN_GEO_GROUP", "4")))
        self.depth = PipelinedRunner.default_depth() if depth is None else depth
        prio = 0           # (default priority: high-priority side streams starve the feature pass; the switch PRCNN_SIDE_PRIORITY is gone, round 6)
        self.tail, self.sides = _runner_streams(self.device, int(os.environ.get("PRCNN_SIDE_STREAMS", "2")), prio)
        have = _RUNNER_STREAMS[(str(self.device), prio)]
        # the feature-stream graphs are CAPTURED on a stream of their own (a capture cannot run on the default stream) and REPLAYED on
        # the caller's stream, as PipelinedRunner runs them: a fifth busy stream would share one of the four hardware queues with
        # another one and serialise behind it (measured: 4155 instead of 5470 scenes/s at K = 100).  The C library's scratch of the
        # capture stream is theirs alone -- every one of these graphs replays on the same stream, in order.
        if have.get("feat") is None:
            have["feat"] = torch.cuda.Stream(self.device)
        self.feat = have["feat"]
        # depth / group groups ahead + the one being consumed.  One more (the first version) costs 5.5 % at K = 96 -- 5100-5160 instead of
        # 5406 scenes/s, the eager runner's figure: a fifth of the slots' 23 GB more to walk through per rotation; one less stalls (2765)
        # never fewer than 2: the slot of the batch in flight must not be the one the next chain is written into
        want = int(os.environ.get("PRCNN_GRAPH_SLOTS", "0"))
        if want == 1 or want < 0:
            raise ValueError("PRCNN_GRAPH_SLOTS=%d: the graphed runner needs at least 2 group slots" % want)
        self.n_slots = want or max(2, -(-self.depth // self.group) + 1)
        # PAIR consecutive batches of a group share the launches of every stage behind the geometry (RPN stage, proposal layer + RoI
        # geometry, RCNN features, final stage): one MEMBER of a slot = `pair` batches = pair x B scenes per launch.  Fatter launches:
        # the same 160 / 800 scenes run 4 % / 3.5 % faster in steps of 16 than in steps of 8; the detections of a
        # batch come back up to 2 pair - 1 submits late instead of 1.
        self.pair = max(1, RCNN_PAIR)
        if self.group % self.pair:
            self.pair = 1
        self.shape = None
        self._assigned = []          # [(batch tensor, group slot, batch index within the slot)] chains launched, batch not yet submitted
        self._chains = self._assigned
        self._pending = None         # the member being filled: {"s": slot, "m": member, "halves": set of batch positions submitted so far}
        self._inflights = collections.deque()    # ("graph", slot, member, [valid batch positions]) | ("eager", det), oldest first
        self._out = collections.deque()          # detections finished and not handed back yet, in submit order
        self._next_slot = 0
        self.captures = 0

    # ---- capture -------------------------------------------------------------------------------------------------------------
    def _capture(self, stream, pool, fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            stream.synchronize()
            # thread_local: other threads of the process (pinned-memory loaders, writers) may call into HIP during a capture
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            try:
                out = fn()
            finally:
                g.capture_end()
        self.captures += 1
        return g, out

    @torch.no_grad()
    def _build(self, first):
        """slots, warm-up of every role stream (the C library's per-stream scratch must exist before a capture), the captures"""
        eng, cfg, G = self.engine, self.cfg, self.group
        B, N, _ = first.shape
        self.shape = tuple(first.shape)
        P = self.pair
        Bm, Gm = P * B, G // P                                   # scenes per member launch, members per slot
        eng.check_weights()
        torch.cuda.synchronize(self.device)
        self.xin = [torch.empty((G * B, N, 3), dtype=torch.float32, device=self.device) for _ in range(self.n_slots)]
        for x in self.xin:
            for k in range(G):
                x[k * B:(k + 1) * B].copy_(first)              # valid clouds everywhere: a partly filled group computes on them
        parts = lambda s: [self.xin[s][k * Bm:(k + 1) * Bm] for k in range(Gm)]

        def tail_stage(st):
            rois, roi_scores = eng.propose(st)
            return {"rois": rois, "roi_scores": roi_scores, "rg": eng.rcnn_geometry(st, rois)}

        def final_stage(tl, out):
            ret = {"rois": tl["rois"], "rcnn_cls": out["rcnn_cls"], "rcnn_reg": out["rcnn_reg"]}
            det = postprocess(cfg, ret, Bm, blob_scenes=B if P > 1 else None)
            det.update(ret)
            return det

        # warm-up, eagerly, once per stream that will capture (and to have real data behind every pointer while capturing)
        for side in self.sides:
            with torch.cuda.stream(side):
                geos = eng.geometry_group(parts(0))
            side.synchronize()
        with torch.cuda.stream(self.feat):
            st = eng.rpn_stage(parts(0)[0], geos[0])
        self.feat.synchronize()
        with torch.cuda.stream(self.tail):
            tl = tail_stage(st)
        self.tail.synchronize()
        with torch.cuda.stream(self.feat):
            out = eng.rcnn_features(tl["rg"])
        self.feat.synchronize()
        post_stream = self.feat if FINAL_ON_FEATURE else self.tail
        with torch.cuda.stream(post_stream):
            final_stage(tl, out)
        post_stream.synchronize()
        del geos, st, tl, out

        # ONE MEMORY POOL PER GRAPH.  Graphs that share a pool may only be replayed in the order of their capture with the outputs of
        # the later ones dead: the temporaries of an earlier capture are free memory when the later one allocates its OUTPUTS, so
        # replaying the earlier graph writes over them (seen: the geometry graph of slot 0 replayed while the RPN stages of slot 4
        # still needed slot 4's index tables -> memory fault).  The slots rotate, so no such order exists here.
        mem0 = torch.cuda.memory_reserved(self.device)
        pool = torch.cuda.graph_pool_handle
        self.slots = []
        for s in range(self.n_slots):
            side = self.sides[s % len(self.sides)]
            g_geo, geos = self._capture(side, pool(), lambda: eng.geometry_group(parts(s)))
            slot = {"side": side, "g_geo": g_geo, "geos": geos, "ev_geo": torch.cuda.Event(), "members": []}
            for k in range(Gm):
                xb = parts(s)[k]
                g_rpn, st = self._capture(self.feat, pool(), lambda: eng.rpn_stage(xb, geos[k]))
                g_tail, tl = self._capture(self.tail, pool(), lambda: tail_stage(st))
                g_rcnn, out = self._capture(self.feat, pool(), lambda: eng.rcnn_features(tl["rg"]))
                # (captured on the stream family it replays on: its kernels' library scratch is keyed by the capture stream, and graphs
                #  that share scratch must replay on one stream, in order)
                g_post, det = self._capture(post_stream, pool(), lambda: final_stage(tl, out))
                slot["members"].append({"g_rpn": g_rpn, "g_tail": g_tail, "g_rcnn": g_rcnn, "g_post": g_post,
                                        "st": st, "tl": tl, "out": out, "det": det,
                                        "ev_rpn": torch.cuda.Event(), "ev_prop": torch.cuda.Event(), "ev_rcnn": torch.cuda.Event(),
                                        "ready": torch.cuda.Event(), "used": False})
            self.slots.append(slot)
        self.graph_bytes = torch.cuda.memory_reserved(self.device) - mem0
        torch.cuda.synchronize(self.device)

    def _conforms(self, pts):
        return (pts is not None and pts.is_cuda and pts.dtype == torch.float32 and pts.dim() == 3 and pts.shape[-1] == 3 and
                (self.shape is None or tuple(pts.shape) == self.shape))

    # ---- replay --------------------------------------------------------------------------------------------------------------
    def _where(self, pts):
        for a in self._assigned:
            if a[0] is pts:
                return a
        return None

    def _target_slot_state(self):
        """the slot the next chain would be written into -> (slot index, holds a batch in flight or the member being filled?, holds
        assigned batches that were not submitted yet?)"""
        s = self._next_slot % self.n_slots
        busy = any(f[0] == "graph" and f[1] == s for f in self._inflights) or (self._pending is not None and self._pending["s"] == s)
        return s, busy, any(a[1] == s for a in self._assigned)

    def _launch_group(self, batch_list, main):
        s = self._next_slot % self.n_slots
        self._next_slot += 1
        slot = self.slots[s]
        side = slot["side"]
        B = self.shape[0]
        side.wait_stream(main)                              # the batches are ready on the caller's stream
        for m in slot["members"]:
            if m["used"]:
                side.wait_event(m["ev_rcnn"])               # the slot's previous batches have been read to the end
                m["used"] = False
        with torch.cuda.stream(side):
            for k, pts in enumerate(batch_list):
                self.xin[s][k * B:(k + 1) * B].copy_(pts, non_blocking=True)
            if _GRAPH_DEBUG & 1:
                torch.cuda.synchronize(self.device)
                print("[graph debug] geometry of slot %d: %d batches" % (s, len(batch_list)), flush=True)
            slot["g_geo"].replay()
            slot["ev_geo"].record(side)
            if _GRAPH_DEBUG & 2:
                torch.cuda.synchronize(self.device)
                print("[graph debug] geometry of slot %d done" % s, flush=True)
        for k, pts in enumerate(batch_list):
            self._assigned.append((pts, s, k))

    @torch.no_grad()
    def submit(self, cur, upcoming=None):
        """-> the detections of an EARLIER batch (in submit order), or None"""
        main = torch.cuda.current_stream(self.device)
        todo = [] if upcoming is None else (list(upcoming) if isinstance(upcoming, (list, tuple)) else [upcoming])
        todo = [p for p in todo if p is not None][:max(1, self.depth)]
        if self.shape is None and self._conforms(cur):
            self._build(cur)
        if not self._conforms(cur) or self.shape is None:
            return self._submit_eager(cur, main)
        todo = [p for p in todo if self._conforms(p)]
        a = self._where(cur)
        if a is None:                                       # cold start (or a caller that looks less far ahead)
            self.engine.check_weights()
            self._close_pending(main)                       # a member left half filled: its batches run now
            s_next, busy, holds_assigned = self._target_slot_state()
            if busy:                                        # few slots: a batch in flight lives where this chain goes -- its RCNN and
                self._finish_slot(s_next)                   # final stages are enqueued first (the chain waits for ev_rcnn)
            if holds_assigned:                              # batches announced earlier and never submitted: their chain is dropped
                self._assigned[:] = [x for x in self._assigned if x[1] != s_next]
            self._launch_group([cur] + [p for p in todo if self._where(p) is None][:self.group - 1], main)
            a = self._where(cur)
        self._assigned[:] = [x for x in self._assigned if x is not a]
        missing = [p for p in todo if self._where(p) is None]
        have = len(todo) - len(missing)
        # a look-ahead that got SHORTER than the caller's usual one: the run is ending and `missing` is all that is left -- its chain
        # starts now, not when the pipeline is about to run dry (the last, partly filled group of a run whose length is not a multiple
        # of the group used to be launched 1-2 steps before its first batch was due: a 3 ms chain, 2 ms of stall; K = 30: 5830 -> 6400)
        self._max_todo = max(getattr(self, "_max_todo", 0), len(todo))
        ending = len(todo) < self._max_todo
        if missing and (len(missing) >= self.group or have <= 1 or ending):
            s_next, busy, holds_assigned = self._target_slot_state()
            # a look-ahead chain is optional: it waits for a later submit while its slot still holds the current batch or batches
            # that were assigned and not submitted yet; the slot of a batch in flight is released by finishing that batch first
            pend_here = self._pending is not None and self._pending["s"] == s_next
            if not holds_assigned and s_next != a[1] and not pend_here:
                if busy:
                    self._finish_slot(s_next)
                self.engine.check_weights()
                self._launch_group(missing[:self.group], main)
        _, s, k = a
        m, h = k // self.pair, k % self.pair
        if self._pending is not None and (self._pending["s"], self._pending["m"]) != (s, m):
            self._close_pending(main)                       # the caller skipped the rest of that member
        if self._pending is None:
            self._pending = {"s": s, "m": m, "halves": set()}
        self._pending["halves"].add(h)
        if h == self.pair - 1:
            self._close_pending(main)
        return self._out.popleft() if self._out else None

    def _close_pending(self, main):
        """RPN stage and proposal stage of the member being filled (its batches not submitted hold an earlier pass's clouds: computed,
        never handed back), then the RCNN + final stages of the member in flight behind them"""
        p, self._pending = self._pending, None
        if p is None:
            return
        s, mi = p["s"], p["m"]
        slot = self.slots[s]
        m = slot["members"][mi]
        feat, tail = main, self.tail
        feat.wait_event(slot["ev_geo"])
        with torch.cuda.stream(feat):
            m["g_rpn"].replay()
            m["ev_rpn"].record(feat)
        if _GRAPH_DEBUG & 4:
            torch.cuda.synchronize(self.device)
            print("[graph debug] slot %d member %d rpn done" % (s, mi), flush=True)
        tail.wait_event(m["ev_rpn"])
        with torch.cuda.stream(tail):
            m["g_tail"].replay()
            m["ev_prop"].record(tail)
        if _GRAPH_DEBUG & 4:
            torch.cuda.synchronize(self.device)
            print("[graph debug] slot %d member %d tail done" % (s, mi), flush=True)
        if self._inflights:
            self._finish_inflight()
        m["used"] = True
        self._inflights.append(("graph", s, mi, sorted(p["halves"])))

    def _submit_eager(self, cur, main):
        self._close_pending(main)
        while self._inflights:                              # another shape: the pipeline drains first (results stay in order)
            self._finish_inflight()
        det = infer_batch(self.model, self.cfg, cur, engine=self.engine)
        ready = torch.cuda.Event()
        ready.record(main)
        det["ready"], det["stream"] = ready, main
        self._inflights.append(("eager", det))
        return self._out.popleft() if self._out else None

    def _finish_slot(self, s):
        """RCNN + final stages of every batch in flight up to the last one that lives in group slot `s` (in order)"""
        if self._pending is not None and self._pending["s"] == s:
            self._close_pending(torch.cuda.current_stream(self.device))
        while any(f[0] == "graph" and f[1] == s for f in self._inflights):
            self._finish_inflight()

    def _finish_inflight(self):
        """RCNN + final stages of the OLDEST member in flight; its batches' detections go onto self._out"""
        if not self._inflights:
            return
        f = self._inflights.popleft()
        if f[0] == "eager":
            self._out.append(f[1])
            return
        _, s, mi, halves = f
        m = self.slots[s]["members"][mi]
        feat, tail = torch.cuda.current_stream(self.device), self.tail
        feat.wait_event(m["ev_prop"])
        with torch.cuda.stream(feat):
            m["g_rcnn"].replay()
            m["ev_rcnn"].record(feat)
        if _GRAPH_DEBUG & 4:
            torch.cuda.synchronize(self.device)
            print("[graph debug] slot %d member %d rcnn done" % (s, mi), flush=True)
        post = feat if FINAL_ON_FEATURE else tail
        if not FINAL_ON_FEATURE:
            tail.wait_event(m["ev_rcnn"])
        with torch.cuda.stream(post):
            m["g_post"].replay()
            m["ready"].record(post)
        if _GRAPH_DEBUG & 4:
            torch.cuda.synchronize(self.device)
            print("[graph debug] slot %d member %d done" % (s, mi), flush=True)
        if self.pair == 1:
            det = dict(m["det"])
            det["ready"], det["stream"] = m["ready"], post
            self._out.append(det)
            return
        B = self.shape[0]
        for h in halves:                                    # one detections dict per batch: views of the member's tensors
            det = {}
            blobs = m["det"].get("blobs")
            for key, v in m["det"].items():
                if key in ("blob", "blobs"):
                    continue
                if isinstance(v, list):                     # per-batch views of the batch's own blob (boxes, scores, num)
                    det[key] = v[h]
                elif torch.is_tensor(v):
                    per = v.shape[0] // self.pair           # rows of this tensor per batch (B scenes, or B x rois)
                    det[key] = v[h * per:(h + 1) * per]
            # round 5: the final stage writes one blob per BATCH of the member (prcnn_rcnn_postprocess_blobs): one copy per batch
            det["blob"] = blobs[h] if blobs is not None else None
            det["ready"], det["stream"] = m["ready"], post
            self._out.append(det)

    @torch.no_grad()
    def flush(self):
        """Finish what is still in the pipeline and return the detections of the OLDEST batch not handed back yet; None when nothing is
        left (call until then: with pair = 2 up to three batches are outstanding).  Chains of batches never submitted are dropped."""
        self._assigned[:] = []
        if not self._out:
            self._close_pending(torch.cuda.current_stream(self.device))
            while self._inflights and not self._out:
                self._finish_inflight()
        return self._out.popleft() if self._out else None

    def drain(self):
        """flush() until the pipeline is empty: the detections of every batch not handed back yet, oldest first (probes and
        timing loops call this in front of a synchronize -- ONE flush() leaves up to two batches' RCNN + final stages unlaunched)."""
        out = []
        while True:
            det = self.flush()
            if det is None:
                return out
            out.append(det)


def _tensors(obj):
    if torch.is_tensor(obj):
        if obj.device.type != "meta":             # (a shape-only index tensor: dropin/pointnet2_cuda.py rcnn_roi_geometry_packs_wrapper)
            yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)
    elif hasattr(obj, "rowinfo"):                 # BallPack (distinct-row list of an index tensor)
        for v in obj.tensors():
            yield v


_RESULT_ROW = " -1 -1" + " %.4f" * 13 + "\n"


def kitti_result_text(calib, bbox3d, scores, img_shape, cls_name="Car"):
    """The result file of one scene as ONE string (tools/eval_rcnn.py:76-101 ``save_kitti_format``; pinned to the text the
    reference writes by tests/golden g11): per surviving box ``<class> -1 -1 alpha x1 y1 x2 y2 h w l x y z ry score``, %.4f.
    Whole-array form: the image boxes of all corners in one projection, clipped to the image; boxes that project wider or
    taller than 80 % of it are dropped; the observation angle alpha = ry + beta - sign(beta) * pi / 2 with beta = atan2(z, x) in the
    boxes' own precision; the 13 numeric columns of all rows go through a single format call."""
    n = int(bbox3d.shape[0])
    if n == 0:
        return ""
    bbox3d = np.asarray(bbox3d)
    img_boxes = calib.corners3d_to_img_boxes(kitti_utils.boxes3d_to_corners3d(bbox3d))[0]
    h, w = img_shape[0], img_shape[1]
    img_boxes = np.clip(img_boxes, 0, np.array([w - 1, h - 1, w - 1, h - 1]))
    ok = ((img_boxes[:, 2] - img_boxes[:, 0]) < w * 0.8) & ((img_boxes[:, 3] - img_boxes[:, 1]) < h * 0.8)
    beta = np.arctan2(bbox3d[:, 2], bbox3d[:, 0])
    alpha = -np.sign(beta) * np.pi / 2 + beta + bbox3d[:, 6]
    table = np.empty((n, 13), dtype=np.float64)
    table[:, 0] = alpha
    table[:, 1:5] = img_boxes
    table[:, 5:8] = bbox3d[:, 3:6]
    table[:, 8:11] = bbox3d[:, 0:3]
    table[:, 11] = bbox3d[:, 6]
    table[:, 12] = np.asarray(scores)
    table = table[ok]
    return ((cls_name + _RESULT_ROW) * len(table)) % tuple(table.reshape(-1).tolist())


def kitti_result_lines(calib, bbox3d, scores, img_shape, cls_name="Car"):
    """The same as a list of lines (for the in-memory AP evaluation)."""
    return kitti_result_text(calib, bbox3d, scores, img_shape, cls_name).split("\n")[:-1]


def save_kitti_format(sample_id, calib, bbox3d, kitti_output_dir, scores, img_shape, cls_name="Car"):
    """One result file per scene (empty when nothing survives); returns the number of lines."""
    text = kitti_result_text(calib, bbox3d, scores, img_shape, cls_name)
    with open(os.path.join(kitti_output_dir, "%06d.txt" % sample_id), "w") as f:
        f.write(text)
    return text.count("\n")


def detections_to_annos(table, counts, source, cls_name="Car"):
    """Gathered detection table [S, M, 9] (+ counts) -> (scene ids, KITTI annotation dicts), through the same
    %.4f text form the result files carry, so the in-memory AP equals the AP of the written files."""
    from . import kitti_eval
    ids, annos = [], []
    tb, ct = table.numpy(), counts.numpy()
    for s in np.argsort(tb[:, 0, 8], kind="stable"):
        sid, n = int(tb[s, 0, 8]), int(ct[s])
        calib, shape = source.calib_and_shape(sid)
        ids.append(sid)
        annos.append(kitti_eval.annos_from_lines(kitti_result_lines(calib, tb[s, :n, 0:7], tb[s, :n, 7], shape, cls_name)))
    return ids, annos


def evaluate_detections(table, counts, source, current_class=0, dataset="kitti", device_id=0):
    """Rank-0 tail of the sharded evaluation: AP of the gathered detections against the source's labels
    (tools/eval_rcnn.py:706-713 -> evaluate/evaluate.py).  Returns (result text, dict)."""
    from . import kitti_eval
    ids, dt_annos = detections_to_annos(table, counts, source)
    gt_annos = [kitti_eval.annos_from_lines(source.label_lines(i)) for i in ids]
    return kitti_eval.get_official_eval_result(gt_annos, dt_annos, current_class, dataset, device_id=device_id)


class RecallStats:
    """Recall of the RoIs and of the refined boxes against the ground truth (eval_rcnn.py:539-570, :669-679): per scene the
    3-D IoU matrix boxes x gt through the extension's BEV overlap kernel (iou3d_utils.boxes_iou3d_gpu -> K10), a gt box
    counts as recalled at threshold t when some box overlaps it with IoU > t.  Counters stay on the device until
    ``result()``; ALL M decoded boxes of a scene enter (before score threshold and NMS), as in the reference."""
    THRESH = (0.1, 0.3, 0.5, 0.7, 0.9)

    def __init__(self, device):
        self.device = torch.device(device)
        self.rcnn = torch.zeros(len(self.THRESH), dtype=torch.int64, device=self.device)
        self.roi = torch.zeros(len(self.THRESH), dtype=torch.int64, device=self.device)
        self.total_gt = 0
        self._th = torch.tensor(self.THRESH, dtype=torch.float32, device=self.device)

    @torch.no_grad()
    def update(self, pred_boxes3d, roi_boxes3d, gt_list):
        """pred_boxes3d / roi_boxes3d (B,M,7) device; gt_list: B arrays (n_k,7) [x,y,z,h,w,l,ry] (all-zero rows = padding)"""
        for k, gt in enumerate(gt_list):
            gt = np.asarray(gt, dtype=np.float32).reshape(-1, 7)
            n = gt.shape[0]
            while n > 0 and gt[n - 1].sum() == 0:           # trailing zero padding of the collated batch (:549-552)
                n -= 1
            if n == 0:
                continue
            g = torch.from_numpy(gt[:n]).to(self.device, non_blocking=True)
            for boxes, acc in ((pred_boxes3d[k], self.rcnn), (roi_boxes3d[k], self.roi)):
                iou = iou3d_utils.boxes_iou3d_gpu(boxes.contiguous(), g)
                best = iou.max(dim=0).values
                acc += (best.unsqueeze(0) > self._th.unsqueeze(1)).sum(dim=1)
            self.total_gt += n

    def result(self):
        rcnn, roi = self.rcnn.cpu().tolist(), self.roi.cpu().tolist()
        out = {"total_gt_bbox": self.total_gt}
        for i, t in enumerate(self.THRESH):
            out["rpn_recall(thresh=%.2f)" % t] = roi[i] / max(self.total_gt, 1.0)
            out["rcnn_recall(thresh=%.2f)" % t] = rcnn[i] / max(self.total_gt, 1.0)
            out["rpn_recalled(thresh=%.2f)" % t] = roi[i]
            out["rcnn_recalled(thresh=%.2f)" % t] = rcnn[i]
        return out


_NODE_AFFINITY = None          # the affinity this process STARTED with (captured once: eval_scenes may run several times per process)


def _node_affinity():
    global _NODE_AFFINITY
    if _NODE_AFFINITY is None:
        try:
            _NODE_AFFINITY = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            _NODE_AFFINITY = list(range(os.cpu_count() or 1))
    return _NODE_AFFINITY


def _cpu_topology(allowed):
    """-> [[(physical cpu, sibling, ...), ...] per NUMA node], restricted to ``allowed``, from sysfs; None when it cannot be read.
    (EPYC hosts number the first hardware threads 0 .. P-1 across the sockets and the SMT siblings P .. 2P-1: a CONTIGUOUS range of CPU
    ids is not a set of neighbouring cores -- on the 2 x 64-core MI355X boxes ids 64-127 are the OTHER socket, 128-191 the siblings of 0-63.)"""
    import glob

    def parse(text):
        out = []
        for part in text.strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                out += range(int(lo), int(hi or lo) + 1)
        return out
    try:
        allowed = set(allowed)
        nodes = []
        for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"), key=lambda q: int(q.split("node")[-1].split("/")[0])):
            with open(path) as f:
                cpus = [c for c in parse(f.read()) if c in allowed]
            groups, seen = [], set()
            for c in cpus:
                if c in seen:
                    continue
                with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                    sib = [x for x in parse(f.read()) if x in allowed]
                seen.update(sib)
                groups.append(tuple(sorted(sib)))
            if groups:
                nodes.append(groups)
        if {c for n in nodes for t in n for c in t} != allowed:     # sysfs does not describe the CPUs we were given: no topology
            return None
        return nodes or None
    except (OSError, ValueError):
        return None


def slice_topology(topo, world, k):
    """rank k of ``world`` local ranks on a host of ``topo`` (= _cpu_topology): the ranks are dealt to the NUMA nodes in order (GPUs
    0 .. W/2-1 hang off node 0, the rest off node 1 on the 2-socket MI355X hosts), a node's PHYSICAL cores are cut into equal runs, and
    a rank gets its run's first hardware threads followed by their SMT siblings -> list of CPU ids, physical cores first."""
    rpn = -(-world // len(topo))                                       # ranks per node
    node = min(k // rpn, len(topo) - 1)
    groups = topo[node]
    on_node = max(1, min(rpn, world - node * rpn))
    gper = max(1, len(groups) // on_node)
    g = groups[(k % rpn) * gper:(k % rpn) * gper + gper] or groups
    return [t[0] for t in g] + [c for t in g for c in t[1:]]


PIN_CORES = 32      # CPUs a rank's processes are confined to (profiles/r06_driver_sweep.md: the whole driver on 32 cores of the GPU's socket
#                     runs 5800-6200 scenes/s on the KITTI tree, on 64 cores 5000-5400, unconfined on 256 CPUs 4500-5600)


def host_budget(world=None, local_rank=None, cores=None):
    """The share of the host one rank may use when W ranks of a node each drive a GPU with loader and writer processes
    (at 16 loaders + 6 writers per rank, 8 ranks are 176 processes on 128-256 cores with no placement).
    -> dict: cores (the CPU ids of this rank: a contiguous slice of the node's CPUs -- on the MI355X hosts GPUs 0-3 hang off
    socket 0 and 4-7 off socket 1, and Linux numbers a socket's cores contiguously, so the slice stays on the GPU's NUMA node),
    loaders, writers.  PRCNN_LOADER_WORKERS / PRCNN_WRITER_PROCS override the counts, PRCNN_NO_AFFINITY=1 the pinning."""
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if world is None else int(world)
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    world = max(1, world)
    topo_ok = cores is None
    if cores is None:
        # always sliced from the affinity the process started with, never from a slice an earlier call pinned it to (ADVICE r3:
        # 128 cores became 16, then 2, then 1 over repeated eval_scenes calls); a rank its launcher already confined to a 1/W share
        # of the node (or less) keeps that share whole
        cores = _node_affinity()
        if world > 1 and len(cores) * world <= (os.cpu_count() or 0):
            world_slices = 1
        else:
            world_slices = world
    else:
        world_slices = world
    per = max(1, len(cores) // world_slices)
    k = local_rank % world_slices
    mine = cores[k * per:k * per + per] or cores
    pin = None
    if topo_ok:
        # the real machine: a rank's share = a run of PHYSICAL cores of one NUMA node (GPUs 0 .. W/2-1 hang off node 0, the rest off
        # node 1) plus their SMT siblings, physical cores first; the processes are pinned to the first PIN_CORES of it
        topo = _cpu_topology(cores)
        if topo:
            mine = slice_topology(topo, world_slices, k)
            pin = mine[:PIN_CORES]
    # one core for the thread that feeds the GPU, a quarter of the rest for the writers (text formatting), the rest for the loaders --
    # at most 6 + 2 (round 5, profiles/r05_driver_shares.md: the whole driver runs 5260-5400 scenes/s with 6 loaders + 2 writers,
    # 4550-5110 with 8 + 2, 3800-3900 with 8-10 + 3 and 3200-3600 with the 16 + 6 of rounds 2-4 on the same 256-core box: beyond what
    # the engine consumes, more producer processes only add wake-ups and pinned-memory traffic around the one thread that feeds the GPU)
    spare = max(1, len(mine) - 1)
    writers = max(1, min(2, spare // 4))
    loaders = max(1, min(6, spare - writers))
    if pin is None:                                     # no topology (or an explicit core list): a single rank takes the head of its cores
        pin = mine[:PIN_CORES] if world_slices == 1 else mine
    return {"cores": mine, "pin": pin,
            "loaders": int(os.environ.get("PRCNN_LOADER_WORKERS", loaders)),
            "writers": int(os.environ.get("PRCNN_WRITER_PROCS", writers)), "world": world, "local_rank": local_rank}


def pin_to_budget(budget):
    """Restrict this process (and the loader / writer processes it starts: affinity is inherited) to the rank's cores -- since round 6
    also a single rank, to budget["pin"]: PIN_CORES neighbouring cores next to its GPU.  eval_scenes restores the previous affinity
    when it returns.  PRCNN_NO_AFFINITY=1: leave the affinity alone."""
    if os.environ.get("PRCNN_NO_AFFINITY") == "1":
        return False
    try:
        os.sched_setaffinity(0, budget.get("pin") or budget["cores"])
        return True
    except (AttributeError, OSError):
        return False


def shard_scene_ids(num_scenes, rank, world):
    """Rank r evaluates scenes r, r+world, ... (independent units)."""
    return list(range(rank, num_scenes, world))


def pack_detections(scene_ids, det_batches, max_det):
    """Host-side table [S, max_det, 9] = 7 box + score + scene id, zero padded, plus counts [S].
    numpy on purpose: the per-batch slice assignments of the first version were torch CPU kernels, each of which may fan out
    over the host's OpenMP pool -- tens of milliseconds for 20 batches on a 128-core box, inside the bench's clock."""
    S = len(scene_ids)
    table = np.empty((S, max_det, 9), dtype=np.float32)
    det_batches = list(det_batches)
    if det_batches:
        table[:, :, 0:7] = np.concatenate([np.asarray(d[0]) for d in det_batches], 0)
        table[:, :, 7] = np.concatenate([np.asarray(d[1]) for d in det_batches], 0)
        counts = np.concatenate([np.asarray(d[2]) for d in det_batches], 0).astype(np.int32)
    else:
        counts = np.zeros((0,), dtype=np.int32)
    table[:, :, 8] = np.asarray(scene_ids, dtype=np.float32).reshape(-1, 1)
    return torch.from_numpy(table), torch.from_numpy(counts)


def all_gather_detections(table, counts, device, force=False):
    """The ONE collective of the job: all ranks exchange their padded detection tables
    (RCCL all_gather over xGMI when the backend is nccl; gloo in the CPU tests).  Tables are
    padded to the largest per-rank scene count so that all_gather_into_tensor applies; the padding rows
    (count -1) are stripped and the rows come back in scene-id order on every rank.
    ``force``: run the exchange on a world of ONE rank as well (the identity up to the id sort) instead of returning early -- how
    the RCCL leg (device-side padding, both all_gather_into_tensor calls on HIP tensors, strip, sort) is exercised on a 1-GPU box
    (tests/test_gpu_configs.py; bench.py --gpus 1 under torchrun sets it when PRCNN_FORCE_GATHER=1)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if force:
            raise RuntimeError("all_gather_detections(force=True): torch.distributed is not initialised")
        return table, counts
    if dist.get_world_size() == 1 and not force:
        return table, counts
    world = dist.get_world_size()
    n_local = torch.tensor([table.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    smax = int(max(int(s.item()) for s in sizes))
    pad_t = torch.zeros((smax,) + tuple(table.shape[1:]), dtype=table.dtype, device=device)
    pad_c = torch.full((smax,), -1, dtype=torch.int32, device=device)   # -1 marks padding rows
    pad_t[:table.shape[0]] = table.to(device)
    pad_c[:counts.shape[0]] = counts.to(device)
    out_t = torch.empty((world * smax,) + tuple(table.shape[1:]), dtype=table.dtype, device=device)
    out_c = torch.empty((world * smax,), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out_t, pad_t)
    dist.all_gather_into_tensor(out_c, pad_c)
    real = out_c >= 0
    out_t, out_c = out_t[real].cpu(), out_c[real].cpu()
    # rank-major as gathered (r, r + W, ... per rank) -> scene-id order, the order of a single-process run: what rank 0 writes and
    # scores does not depend on the world size (ids are exact in float32 up to 2^24 scenes)
    order = torch.argsort(out_t[:, 0, 8], stable=True) if out_t.shape[0] and out_t.shape[1] else torch.arange(out_t.shape[0])
    return out_t[order].contiguous(), out_c[order].contiguous()



LOADER_TARGET = 6500.0          # scenes/s the loader processes are sized for


def _calib_row(calib):
    """P2 | R0 | V2C of a calibration object as 33 floats (what a loader process sends back instead of the object)"""
    row = np.zeros(33, dtype=np.float32)
    row[0:12] = np.asarray(calib.P2, np.float32).reshape(-1)
    r0, v2c = getattr(calib, "R0", None), getattr(calib, "V2C", None)
    row[12:21] = (np.eye(3, dtype=np.float32) if r0 is None else np.asarray(r0, np.float32)).reshape(-1)
    row[21:33] = (np.eye(3, 4, dtype=np.float32) if v2c is None else np.asarray(v2c, np.float32)).reshape(-1)
    return row


class _RowCalib:
    """the parent's side of _calib_row: projection for the result writer + the arrays DeviceInputStage.pack_calib reads"""

    def __init__(self, row):
        self.P2, self.R0, self.V2C = row[0:12].reshape(3, 4), row[12:21].reshape(3, 3), row[21:33].reshape(3, 4)

    def corners3d_to_img_boxes(self, corners3d):
        n = corners3d.shape[0]
        hom = np.concatenate((corners3d, np.ones((n, 8, 1))), axis=2)
        img = np.matmul(hom, self.P2.T)
        x, y = img[:, :, 0] / img[:, :, 2], img[:, :, 1] / img[:, :, 2]
        boxes = np.stack((np.min(x, axis=1), np.min(y, axis=1), np.max(x, axis=1), np.max(y, axis=1)), axis=1)
        return boxes, np.stack((x, y), axis=2)


def _shm_worker(source, scene_ids, batch_size, raw, buf, tasks, results, pin=None):
    """A loader process of _ShmFeed: takes (batch index, slot), produces the batch's scenes STRAIGHT INTO the slot of the shared,
    pinned buffer -- sampled clouds (B, npoints, c) or, for the device input stage, raw clouds packed (B, n_max, stride) -- and sends
    back only the small things: per-scene counts, calibration rows, image shapes."""
    _limit_worker_threads()
    if pin:                     # (a fork server that was started before the parent confined itself hands out its own, wider affinity)
        try:
            os.sched_setaffinity(0, pin)
        except (AttributeError, OSError):
            pass
    flat = buf.numpy()
    try:
        while True:
            t = tasks.get()
            if t is None:
                return
            b, slot = t
            ids = scene_ids[b * batch_size:(b + 1) * batch_size]
            loaded = [source.load_raw(i) if raw else source.load(i) for i in ids]
            arrs = [np.ascontiguousarray(l[0], dtype=np.float32) for l in loaded]
            n_max, c = max(1, max(a.shape[0] for a in arrs)), arrs[0].shape[1]
            if len(ids) * n_max * c > flat.shape[1]:
                raise RuntimeError("a batch of %d clouds x %d points x %d floats does not fit a loader slot of %d floats "
                                   "(PRCNN_RAW_SLOT_POINTS)" % (len(ids), n_max, c, flat.shape[1]))
            dst = flat[slot, :len(ids) * n_max * c].reshape(len(ids), n_max, c)
            for k, a in enumerate(arrs):
                dst[k, :a.shape[0]] = a                           # (rows beyond a raw cloud's length are never read: counts)
            results.put((b, slot, n_max, c, [a.shape[0] for a in arrs], np.stack([_calib_row(l[1]) for l in loaded], 0),
                         [tuple(l[2]) for l in loaded]))
    except BaseException as e:                                      # noqa: BLE001 -- the parent re-raises it
        import traceback
        results.put((-1, -1, 0, 0, "loader process failed: %r\n%s" % (e, traceback.format_exc()), None, None))


class _ShmFeed:
    """Loader processes that write into ONE shared, page-locked buffer (round 6).  torch's DataLoader hands every
    batch (device input stage: every raw cloud) over as a tensor in a shared-memory file of its own -- a file descriptor through a unix
    socket per tensor, a copy into pinned memory by a thread of the parent, then the upload: 2.4 ms of the feeding thread per batch of raw
    clouds (profiles/r06_driver.md), the whole driver at 2400 scenes/s beside an engine that does 6500.  Here the buffer is allocated once
    (torch shared memory), registered with the HIP runtime (hipHostRegister: uploads from it are asynchronous DMA), cut into slots, and a
    loader fills the slot its task names; the parent receives a dozen integers and 33 floats per scene, views the slot and enqueues the
    upload.  A slot returns to the free list when its upload has completed (an event).  Batches come back in order."""

    def __init__(self, source, scene_ids, batch_size, raw, workers, ctx, pin, slot_floats, cpus=None):
        import torch.multiprocessing as tmp
        self.batch_size, self.raw = batch_size, raw
        self.n_batches = -(-len(scene_ids) // batch_size)
        n_slots = min(self.n_batches, 2 * workers + 4)
        self.buf = torch.empty((n_slots, slot_floats), dtype=torch.float32).share_memory_()
        self.registered = False
        if pin:
            rc = torch.cuda.cudart().cudaHostRegister(self.buf.data_ptr(), self.buf.numel() * 4, 0)
            self.registered = int(rc) == 0
        mp = tmp.get_context(ctx)
        self.tasks, self.results = mp.Queue(), mp.Queue()
        self.procs = [mp.Process(target=_shm_worker, args=(source, list(scene_ids), batch_size, raw, self.buf, self.tasks, self.results, cpus), daemon=True)
                      for _ in range(workers)]
        for p in self.procs:
            p.start()
        self.free, self.busy, self.ready = collections.deque(range(n_slots)), collections.deque(), {}
        self.next_task = self.next_out = 0
        self._feed()

    def _feed(self):
        while self.busy and self.busy[0][1].query():
            self.free.append(self.busy.popleft()[0])
        while self.free and self.next_task < self.n_batches:
            self.tasks.put((self.next_task, self.free.popleft()))
            self.next_task += 1

    def next(self):
        """-> (host tensor (B, n_max, c): a view of the slot, slot, counts, calibs, shapes) of the next batch, or None at the end"""
        if self.next_out >= self.n_batches:
            return None
        while self.next_out not in self.ready:
            if not self.free and self.next_task < self.n_batches and self.busy and len(self.ready) == 0 and self.results.empty():
                self.busy[0][1].synchronize()                     # every slot is waiting for its upload: wait for the oldest one
                self._feed()
            try:
                r = self.results.get(timeout=1.0)
            except Exception:                                      # noqa: BLE001 (queue.Empty): keep the workers fed, notice dead ones
                self._feed()
                dead = [p.exitcode for p in self.procs if p.exitcode not in (None, 0)]
                if dead or not any(p.is_alive() for p in self.procs):
                    raise RuntimeError("eval_scenes: a loader process has exited (exit codes %s): its batch will never arrive" % (dead or "0"))
                continue
            if r[0] < 0:
                raise RuntimeError(r[4])
            self.ready[r[0]] = r[1:]
        slot, n_max, c, counts, rows, shapes = self.ready.pop(self.next_out)
        self.next_out += 1
        host = self.buf[slot, :len(counts) * n_max * c].view(len(counts), n_max, c)
        return host, slot, counts, [_RowCalib(rows[k]) for k in range(len(counts))], shapes

    def release(self, slot, event):
        """the slot's upload has been enqueued behind `event` (None: the data has been consumed already)"""
        if event is None:
            self.free.append(slot)
        else:
            self.busy.append((slot, event))
        self._feed()

    def close(self):
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if self.registered:
            torch.cuda.synchronize()
            torch.cuda.cudart().cudaHostUnregister(self.buf.data_ptr())
            self.registered = False


def eval_scenes(*args, **kwargs):
    """eval_scenes_pinned with the process's CPU affinity put back afterwards (the driver confines itself and its loader / writer
    processes to the rank's cores: host_budget / pin_to_budget; a caller that goes on to other work -- bench.py's CPU baseline --
    gets the whole host back)."""
    try:
        before = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        before = None
    try:
        return eval_scenes_pinned(*args, **kwargs)
    finally:
        if before is not None:
            # EVERY thread of the process: sched_setaffinity(0) moves the calling thread only, and pool threads (OpenMP, BLAS) that
            # were created while the driver was confined inherited its 32 cores -- bench.py's CPU baseline, which runs afterwards
            # on 128 threads, took 138 s instead of 27 behind a driver leg that had restored the main thread alone
            try:
                tids = [int(t) for t in os.listdir("/proc/self/task")]
            except OSError:
                tids = [0]
            for tid in tids:
                try:
                    os.sched_setaffinity(tid, before)
                except OSError:
                    pass


@torch.no_grad()
def eval_scenes_pinned(model, cfg, device, source, scene_ids, batch_size=8, output_dir=None, workers=None, device_input=False,
                       recall=None, stats=None):
    """Evaluate ``scene_ids`` of a scene source (kitti_io.KittiSource / SyntheticSource) on this rank:
    the counterpart of the batch loop of eval_one_epoch_joint (eval_rcnn.py:493-649) incl. the KITTI
    result files.  Returns (table, counts) as pack_detections.

    ``workers`` loader processes read / filter / subsample scenes ahead of the device (the reference's
    DataLoader workers, eval_rcnn.py:868-871): a scene costs milliseconds of numpy on the host, the device
    needs ~1.3 ms per scene, so a single-threaded loader would be the bottleneck.  Batches arrive in order.
    ``device_input``: the loaders only READ the raw clouds (``source.load_raw``); rectification, validity filter
    and the near/far sampler run on the device (kitti_io.DeviceInputStage, csrc/input_stage.hip).
    ``recall``: a RecallStats that receives every batch's RoIs / refined boxes and the source's ground-truth boxes
    (the reference's recall statistics, skipped with --test).  ``stats``: a dict that receives the completion time of
    every batch (``batch_done``), for steady-state throughput measurements."""
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
    M = cfg.TEST.RPN_POST_NMS_TOP_N
    on_gpu = torch.device(device).type == "cuda"
    runner = make_runner(model, cfg, device) if on_gpu else None
    budget = host_budget()
    if stats is not None:
        stats["host_budget"] = {"loaders": budget["loaders"], "writers": budget["writers"], "cores": len(budget["cores"]), "pin": len(budget["pin"]),
                                "pinned": pin_to_budget(budget)}
    else:
        pin_to_budget(budget)
    starts = list(range(0, len(scene_ids), batch_size))
    if workers is None:
        workers = budget["loaders"]
        if "PRCNN_LOADER_WORKERS" not in os.environ and len(starts) > 2:
            # as many loaders as THIS source needs to feed the engine: a scene is timed here, in the parent (the uniform generator costs
            # ~0.8 ms, the KITTI reader + sampler of kitti_io.KittiSource ~2 ms, a raw .bin read 0.3 ms), and the count covers
            # LOADER_TARGET scenes/s with a third to spare -- at least the budget's 6, at most 12, inside the rank's share of the host.  More is NOT better
            # (profiles/r06_driver_sweep.md, KITTI tree + host sampler: 6 loaders 4060, 12: 5200-5650, 18: 4900, 24: 4700 scenes/s, and
            # 4 writers instead of 2 cost another 15 %; round 5 found the same for the uniform source): every runnable process
            # beyond what the engine consumes takes clock and memory bandwidth from the one thread that feeds the GPU
            load_one = source.load_raw if device_input else source.load
            load_one(scene_ids[0])                                  # (first touch: imports, page cache)
            t0 = time.perf_counter()
            load_one(scene_ids[min(1, len(scene_ids) - 1)])
            t_load = time.perf_counter() - t0
            spare = max(1, len(budget.get("pin") or budget["cores"]) - 1 - budget["writers"])
            workers = int(max(min(workers, spare), min(np.ceil(t_load * LOADER_TARGET * 1.33), 12, spare)))
            if stats is not None:
                stats["loader_calibration"] = {"ms_per_scene": round(t_load * 1e3, 2), "loaders": workers}
    stage = None
    if device_input:
        if not on_gpu:
            raise RuntimeError("eval_scenes: device_input needs a GPU")
        from . import kitti_io
        stage = kitti_io.DeviceInputStage(cfg, device, npoints_faraway=getattr(source, "npoints_faraway", 4000),
                                          seed=getattr(source, "seed", 1024))
    feed = None
    if workers > 0 and len(starts) > 2:
        # loader PROCESSES (the scene generator / KITTI reader is Python + numpy: threads would serialise on the GIL) that write into
        # one shared page-locked buffer (_ShmFeed); the parent views a slot and enqueues its upload.
        # Forking a process whose HIP runtime is already initialised is unsupported (sporadic hangs at worker start or
        # exit): once the GPU has been touched the workers come from a fork SERVER (clean interpreters, picklable dataset).
        ctx = "forkserver" if (on_gpu and torch.cuda.is_initialized()) else "fork"
        ctx = os.environ.get("PRCNN_LOADER_CONTEXT", ctx)
        if stage is not None:
            slot_floats = batch_size * int(os.environ.get("PRCNN_RAW_SLOT_POINTS", "200000")) * 4
        else:
            slot_floats = batch_size * cfg.RPN.NUM_POINTS * (4 if cfg.RPN.USE_INTENSITY else 3)
        feed = _ShmFeed(source, scene_ids, batch_size, stage is not None, workers, ctx, on_gpu, slot_floats,
                        cpus=None if os.environ.get("PRCNN_NO_AFFINITY") == "1" else list(budget.get("pin") or budget["cores"]))
        if stats is not None:
            stats["loader_buffer"] = {"slots": int(feed.buf.shape[0]), "MB": round(feed.buf.numel() * 4 / 1e6, 1), "page_locked": feed.registered}

    writers = None
    try:                                   # (the loader processes, their page-locked buffer and the writer pool are released on every way out)
        feed_wait = [0.0]                      # seconds the feeding thread spent inside feed.next() (waiting for a loader + unpacking its message)

        def load(s):
            ids = scene_ids[s:s + batch_size]
            if not ids:
                return None, ids, None
            if feed is not None:
                tw = time.perf_counter()
                host, slot, counts, calibs, shapes = feed.next()
                feed_wait[0] += time.perf_counter() - tw
                meta = list(zip(calibs, shapes))
                if stage is not None:
                    pts, _ = stage.from_packed(host, counts, calibs, shapes, ids, lidar_frame=source.raw_in_lidar_frame,
                                               image_filter=source.raw_needs_image_filter)
                    feed.release(slot, stage.last_done)
                    return pts, ids, meta
                if not on_gpu:
                    pts = host.clone()
                    feed.release(slot, None)
                    return pts, ids, meta
                if feed.registered:
                    pts = host.to(device, non_blocking=True)
                else:                                                   # registration refused: through a pinned staging copy
                    pts = host.pin_memory().to(device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
                feed.release(slot, ev)
                return pts, ids, meta
            if stage is not None:
                raws = [source.load_raw(i)[0] for i in ids]
                meta = [source.calib_and_shape(i) for i in ids]
                pts, _ = stage(raws, [m[0] for m in meta], [m[1] for m in meta], ids,
                               lidar_frame=source.raw_in_lidar_frame, image_filter=source.raw_needs_image_filter)
                return pts, ids, meta
            loaded = [source.load(i) for i in ids]
            host = torch.from_numpy(np.stack([l[0] for l in loaded], 0))
            host = host.pin_memory() if on_gpu else host
            meta = [(l[1], l[2]) for l in loaded]
            return host.to(device, non_blocking=True), ids, meta

        # Results are consumed LATE: the D2H copy of a batch is queued (on the stream that produced it) the moment the batch
        # is submitted, but the host only waits for it `lag` batches later, when it has long completed -- the host thread
        # never stalls on the GPU inside the loop and keeps enqueueing ahead of it.  The KITTI text files are written by a
        # small thread pool (numpy projection + formatting + file I/O per scene; the reference does this inline, :635).
        import collections
        from concurrent.futures import ProcessPoolExecutor
        import multiprocessing
        lag = int(os.environ.get("PRCNN_RESULT_LAG", "3")) if runner is not None else 0
        inflight = collections.deque()
        # writer PROCESSES: the formatting of ~40 text lines per scene is pure Python and would hold the GIL of the thread
        # that feeds the GPU; one job per batch
        writers = None
        if output_dir:
            wctx = "forkserver" if (on_gpu and torch.cuda.is_initialized()) else "fork"
            writers = ProcessPoolExecutor(max_workers=budget["writers"], initializer=_limit_worker_threads,
                                          mp_context=multiprocessing.get_context(os.environ.get("PRCNN_LOADER_CONTEXT", wctx)))
        jobs = []
        results = {}

        def start_copy(det, ids, meta, order):
            with torch.cuda.stream(det["stream"]) if "stream" in det else contextlib.nullcontext():
                if on_gpu and det.get("blob") is not None:
                    hb = torch.empty(det["blob"].shape, dtype=torch.float32, pin_memory=True)
                    hb.copy_(det["blob"], non_blocking=True)            # boxes | scores | num in one transfer
                    host = list(split_detections(hb, det["boxes"].shape[0], det["boxes"].shape[1]))
                    done = torch.cuda.Event()
                    done.record()
                elif on_gpu:
                    host = [torch.empty(det[k].shape, dtype=det[k].dtype, pin_memory=True) for k in ("boxes", "scores", "num")]
                    for h, k in zip(host, ("boxes", "scores", "num")):
                        h.copy_(det[k], non_blocking=True)
                    done = torch.cuda.Event()
                    done.record()
                else:
                    host, done = [det[k] for k in ("boxes", "scores", "num")], None
                if recall is not None:
                    recall.update(det["pred_boxes3d"], det["rois"], [source.gt_boxes3d(i) for i in ids])
            inflight.append((host, done, ids, meta, order))

        def consume():
            (boxes, scores, num), done, ids, meta, order = inflight.popleft()
            if done is not None:
                done.synchronize()
            results[order] = (boxes, scores, num)
            if stats is not None:
                stats.setdefault("batch_done", []).append(time.perf_counter())
            if output_dir:
                nn = num.tolist()
                jobs.append(writers.submit(_write_batch, ids, [m[0] for m in meta], [m[1] for m in meta],
                                           [boxes[k, :nn[k]].numpy() for k in range(len(ids))],
                                           [scores[k, :nn[k]].numpy() for k in range(len(ids))], output_dir, cfg.CLASSES))

        # software pipeline: while batch i is on the device, batch i+1 is loaded and (three-stream runner) the RCNN +
        # final stage of batch i-1 complete; results are consumed `lag` batches late
        submitted = collections.deque()        # (ids, meta, order) of the batches whose detections have not come back yet, oldest first
        order = 0
        depth = runner.depth if runner is not None else 1
        ahead = [load(k * batch_size) for k in range(depth)]   # `depth` batches ahead: the runner starts their geometry chains early
        phase = {"load": 0.0, "submit": 0.0, "copy": 0.0, "consume": 0.0} if stats is not None else None     # host seconds of the feeding thread by phase
        clock = time.perf_counter
        for s in range(0, len(scene_ids), batch_size):
            pts, ids, meta = ahead.pop(0)
            t0, w0 = clock(), feed_wait[0]
            ahead.append(load(s + depth * batch_size))
            t1 = clock()
            if runner is not None:
                det = runner.submit(pts, [a[0] for a in ahead])        # an earlier batch's detections (in submit order), or None
                submitted.append((ids, meta, order))
                t2 = clock()
                if det is not None:
                    start_copy(det, *submitted.popleft())
            else:
                t2 = clock()
                start_copy(infer_batch(model, cfg, pts), ids, meta, order)
            t3 = clock()
            order += 1
            while len(inflight) > lag:
                consume()
            if phase is not None and order > depth:                    # (steady state: behind the first look-ahead's worth of batches)
                phase["feed_next"] = phase.get("feed_next", 0.0) + feed_wait[0] - w0
                phase["load"] += t1 - t0 - (feed_wait[0] - w0); phase["submit"] += t2 - t1; phase["copy"] += t3 - t2; phase["consume"] += clock() - t3
                phase["batches"] = phase.get("batches", 0) + 1
        if phase is not None:
            stats["host_phases_ms_per_batch"] = {k: round(v / max(1, phase.get("batches", 1)) * 1e3, 3) for k, v in phase.items() if k != "batches"}
        while runner is not None and submitted:
            det = runner.flush()                                        # one batch per call, oldest first
            if det is None:
                raise RuntimeError("eval_scenes: the runner returned no detections for %d submitted batches" % len(submitted))
            start_copy(det, *submitted.popleft())
        while inflight:
            consume()
        for j in jobs:
            j.result()
    finally:
        if writers is not None:
            writers.shutdown(wait=True, cancel_futures=True)
        if feed is not None:
            feed.close()
    return pack_detections(scene_ids, [results[k] for k in sorted(results)], M)


def steady_state_rate(stats, batch_size):
    """scenes/s between the completion of the first and of the last batch (loader / writer process start-up excluded)"""
    t = stats.get("batch_done", [])
    if len(t) < 3:
        return float("nan")
    return (len(t) - 1) * batch_size / max(t[-1] - t[0], 1e-9)


def _write_batch(ids, calibs, shapes, boxes, scores, output_dir, cls_name):
    """one writer job: the KITTI result files of one batch (runs in a writer process)"""
    return sum(save_kitti_format(sid, c, b, output_dir, s, sh, cls_name) for sid, c, sh, b, s in zip(ids, calibs, shapes, boxes, scores))


class _SceneDataset(torch.utils.data.Dataset):
    """What a loader process produces for scene k: the sampled cloud (host input stage) or the raw cloud (device stage)."""

    def __init__(self, source, scene_ids, raw):
        self.source, self.scene_ids, self.raw = source, list(scene_ids), raw

    def __len__(self):
        return len(self.scene_ids)

    def __getitem__(self, k):
        if self.raw:
            return torch.from_numpy(np.ascontiguousarray(self.source.load_raw(self.scene_ids[k])[0], dtype=np.float32))
        return torch.from_numpy(self.source.load(self.scene_ids[k])[0])


def _identity(items):
    return items


def _limit_worker_threads(_worker_id=None):
    """Start-up hook of every loader / writer process: ONE thread for the numeric libraries.  A loader's numpy work is a few small
    products per scene (Calibration.lidar_to_rect: (n, 4) x (4, 3)); left alone, each of them fans out over the BLAS / OpenMP pool of
    the whole host (256 threads on the MI355X boxes) -- processes x cores runnable threads around the one thread that feeds the GPU.
    PRCNN_LOADER_THREADS (default 1; 0: leave the libraries alone)."""
    n = int(os.environ.get("PRCNN_LOADER_THREADS", "1"))
    if n <= 0:
        return
    try:
        torch.set_num_threads(n)
    except RuntimeError:
        pass
    try:
        import threadpoolctl
        global _THREAD_LIMIT
        _THREAD_LIMIT = threadpoolctl.threadpool_limits(limits=n)      # kept alive: the limit lasts as long as the object
    except Exception:                                                   # noqa: BLE001 -- no threadpoolctl: the environment variables below
        pass


_THREAD_LIMIT = None


def eval_synthetic(model, cfg, device, scene_ids, batch_size=8, npoints=16384, output_dir=None, raw_points=None):
    """eval_scenes over the synthetic generator (seed = scene id); ``raw_points`` generates denser raw
    clouds that are reduced with the reference's near/far sampler (cross-domain config)."""
    from . import kitti_io
    assert npoints == cfg.RPN.NUM_POINTS, "npoints must equal cfg.RPN.NUM_POINTS"
    src = kitti_io.SyntheticSource(cfg, 0, raw_points=raw_points)
    return eval_scenes(model, cfg, device, src, scene_ids, batch_size, output_dir)


def main(argv=None):
    ap = argparse.ArgumentParser(description="PointRCNN joint evaluation on synthetic KITTI-shaped scenes (MI355X)")
    ap.add_argument("--cfg_file", type=str, default=None, help="reference-style yaml (tools/cfgs/*.yaml)")
    ap.add_argument("--eval_mode", type=str, default="rcnn")
    ap.add_argument("--ckpt", type=str, default=None, help="reference .pth checkpoint (random init if omitted)")
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--scenes", type=int, default=16, help="number of synthetic scenes (ignored with --data_root)")
    ap.add_argument("--raw_points", type=int, default=None,
                    help="synthetic scenes: generate this many raw points per scene (e.g. 180000, the cross-domain "
                         "dense-cloud case) and reduce them with the near/far sampler")
    ap.add_argument("--workers", type=int, default=None, help="host loader processes (default 8)")
    ap.add_argument("--device_input", action="store_true",
                    help="loaders only read raw clouds; rectification, validity filter and the near/far sampler run on the GPU")
    ap.add_argument("--data_root", type=str, default=None, help="directory holding KITTI/object/... and KITTI/ImageSets")
    ap.add_argument("--split", type=str, default=None, help="ImageSets split (default cfg.TEST.SPLIT)")
    ap.add_argument("--output_dir", type=str, default=None)
    ap.add_argument("--eval_ap", action="store_true", help="rank 0: KITTI AP of the gathered detections vs the labels")
    ap.add_argument("--recall", action="store_true", help="RoI / refined-box recall vs the ground truth (the reference's statistics without --test)")
    ap.add_argument("--set", dest="set_cfgs", default=None, nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)

    cfg = config_mod.make_cfg()
    config_mod.apply_eval_defaults(cfg, args.eval_mode)
    if args.cfg_file:
        config_mod.cfg_from_file(cfg, args.cfg_file)
    if args.set_cfgs:
        config_mod.cfg_from_list(cfg, args.set_cfgs)
    if not torch.cuda.is_available():
        raise RuntimeError("eval_rcnn: no GPU visible; this build has no CPU path")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
    model = build_model(cfg, device)
    if args.ckpt:
        load_checkpoint(model, args.ckpt)
    out = os.path.join(args.output_dir, "final_result", "data") if args.output_dir else None
    from . import kitti_io
    if args.data_root:
        source = kitti_io.KittiSource(args.data_root, cfg, args.split or cfg.TEST.SPLIT)
    else:
        source = kitti_io.SyntheticSource(cfg, args.scenes, raw_points=args.raw_points)
    my_ids = [source.ids[i] for i in shard_scene_ids(len(source.ids), rank, world)]
    t0 = time.perf_counter()
    recall = RecallStats(device) if args.recall else None
    stats = {}
    table, counts = eval_scenes(model, cfg, device, source, my_ids, args.batch_size, out, workers=args.workers,
                                device_input=args.device_input, recall=recall, stats=stats)
    elapsed = time.perf_counter() - t0
    if recall is not None:
        for k, v in recall.result().items():
            print("rank %d  %s: %s" % (rank, k, v))
    table, counts = all_gather_detections(table, counts, device)
    if rank == 0:
        print("scenes=%d detections=%d  (%.1f scenes/s on this rank incl. the host input stage%s; steady state %.1f scenes/s, "
              "loader start-up excluded)" %
              (table.shape[0], int(counts.sum()), len(my_ids) / max(elapsed, 1e-9),
               " and the result writer" if out else "", steady_state_rate(stats, args.batch_size)))
        if args.eval_ap:
            text, _ = evaluate_detections(table, counts, source, device_id=device.index or 0)
            print(text)
            if args.output_dir:
                with open(os.path.join(args.output_dir, "final_result", "ap.txt"), "w") as f:
                    f.write(text)


if __name__ == "__main__":
    main()
