"""Offline KITTI AP evaluator around the rotated-IoU HIP kernel (counterpart of evaluate/eval2.py:8-690,
evaluate/kitti_common.py:307-360 and the ``evaluate()`` driver evaluate/evaluate.py:88-135 in its
plain form: no re-scaling / grounding analysis switches).

Same protocol as the reference fork: 41 recall sample points, 11-point mAP (every 4th), six DISTANCE
based "difficulties" instead of KITTI's easy/moderate/hard (eval2.py:48-52):

    level        0        1        2        3        4        5
    depth (m)  (0,30)   (0,70)   (0,70)   (0,30)  (30,50)  (50,70)
    occlusion   <=0      <=1      <=2      <=2      <=2      <=2
    truncation <=.15     <=.3     <=.5     <=.5     <=.5     <=.5

What is different in shape, not in result: the reference cuts the split into ~50 parts, computes a dense
rotated-IoU matrix per part on the GPU (cross-image pairs are discarded) and runs the greedy matching as
numba-jitted Python.  Here every image is one segment of a single block-diagonal launch
(``prcnn_rotate_iou_eval_segmented``), and the matching / PR accumulation are host functions of the same
library (csrc/kitti_stats.hip) fed with the whole split at once.
"""
import ctypes
import io
import os
import pathlib
import re

import numpy as np

from . import _lib

CLASS_NAMES = ["car", "pedestrian", "cyclist"]
CLASS_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting"}
DIST_BOUNDARY = np.array([[0, 0, 0, 0, 30, 50], [30, 70, 70, 30, 50, 70]], dtype=np.float64)
MAX_OCCLUSION = [0, 1, 2, 2, 2, 2]
MAX_TRUNCATION = [0.15, 0.3, 0.5, 0.5, 0.5, 0.5]
N_SAMPLE_PTS = 41


# ---------------------------------------------------------------------------------------------------
# label files (kitti_common.py:307-360)
# ---------------------------------------------------------------------------------------------------
def _anno_from_rows(rows):
    """rows: list of token lists (15 or 16 tokens).  dimensions are stored l, h, w (file order h, w, l)."""
    n = len(rows)
    anno = {
        "name": np.array([r[0] for r in rows]),
        "truncated": np.array([float(r[1]) for r in rows]),
        "occluded": np.array([int(r[2]) for r in rows]),
        "alpha": np.array([float(r[3]) for r in rows]),
        "bbox": np.array([[float(v) for v in r[4:8]] for r in rows], dtype=np.float64).reshape(-1, 4),
        "dimensions": np.array([[float(v) for v in r[8:11]] for r in rows], dtype=np.float64).reshape(-1, 3)[:, [2, 0, 1]],
        "location": np.array([[float(v) for v in r[11:14]] for r in rows], dtype=np.float64).reshape(-1, 3),
        "rotation_y": np.array([float(r[14]) for r in rows]).reshape(-1),
    }
    if n != 0 and len(rows[0]) == 16:
        anno["score"] = np.array([float(r[15]) for r in rows])
    else:
        anno["score"] = np.zeros([n])
    return anno


def get_label_anno(label_path):
    with open(label_path, "r") as f:
        rows = [line.strip().split(" ") for line in f.readlines()]
    return _anno_from_rows(rows)


def get_label_annos(label_folder, image_ids=None):
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        pat = re.compile(r"^\d{6}.txt$")
        image_ids = sorted(int(p.stem) for p in folder.glob("*.txt") if pat.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    return [get_label_anno(folder / ("%06d.txt" % i)) for i in image_ids]


def filter_annos_low_score(annos, thresh):
    out = []
    for a in annos:
        keep = [i for i, s in enumerate(a["score"]) if s >= thresh]
        out.append({k: v[keep] for k, v in a.items()})
    return out


def annos_from_lines(lines):
    """KITTI label lines (strings, as eval_rcnn.save_kitti_format writes them) -> annotation dict."""
    return _anno_from_rows([l.strip().split(" ") for l in lines if l.strip()])


# ---------------------------------------------------------------------------------------------------
# overlaps
# ---------------------------------------------------------------------------------------------------
def image_box_overlap(boxes, query_boxes, criterion=-1):
    """(N,4) x (K,4) [x1,y1,x2,y2] -> (N,K) in boxes.dtype (eval2.py:102-128): no +1 on the extents."""
    boxes = np.asarray(boxes)
    query_boxes = np.asarray(query_boxes)
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = np.zeros((n, k), dtype=boxes.dtype)
    if n == 0 or k == 0:
        return out
    iw = np.minimum(boxes[:, None, 2], query_boxes[None, :, 2]) - np.maximum(boxes[:, None, 0], query_boxes[None, :, 0])
    ih = np.minimum(boxes[:, None, 3], query_boxes[None, :, 3]) - np.maximum(boxes[:, None, 1], query_boxes[None, :, 1])
    barea = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))[:, None]
    qarea = ((query_boxes[:, 2] - query_boxes[:, 0]) * (query_boxes[:, 3] - query_boxes[:, 1]))[None, :]
    inter = iw * ih
    if criterion == -1:
        ua = barea + qarea - inter
    elif criterion == 0:
        ua = np.broadcast_to(barea, inter.shape)
    elif criterion == 1:
        ua = np.broadcast_to(qarea, inter.shape)
    else:
        ua = np.ones_like(inter)
    hit = (iw > 0) & (ih > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        out[hit] = (inter / ua)[hit]
    return out


def rotate_iou_segmented(boxes_list, query_list, criterion=-1, device_id=0):
    """Per-image rotated IoU in one launch.  boxes_list[i] (n_i,5), query_list[i] (k_i,5)
    [cx, cy, w, h, angle] -> list of (n_i,k_i) f32 arrays and the flat concatenation."""
    import torch
    nseg = len(boxes_list)
    n = np.array([len(b) for b in boxes_list], dtype=np.int64)
    k = np.array([len(q) for q in query_list], dtype=np.int64)
    box_off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
    q_off = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
    out_off = np.concatenate([[0], np.cumsum(n * k)]).astype(np.int64)
    total = int(out_off[-1])
    flat = np.zeros((total,), dtype=np.float32)
    if total > 0:
        dev = torch.device("cuda", device_id)
        cat = lambda xs: np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float32).reshape(-1, 5) for x in xs], 0))
        b = torch.from_numpy(cat(boxes_list)).to(dev)
        q = torch.from_numpy(cat(query_list)).to(dev)
        oo, bo, qo = (torch.from_numpy(a).to(dev) for a in (out_off, box_off, q_off))
        out = torch.empty((total,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("prcnn_rotate_iou_eval_segmented", nseg, total, oo.data_ptr(), bo.data_ptr(), qo.data_ptr(),
                      b.data_ptr(), q.data_ptr(), out.data_ptr(), int(criterion), _lib.current_stream(out))
        flat = out.cpu().numpy()
    blocks = [flat[out_off[i]:out_off[i + 1]].reshape(int(n[i]), int(k[i])) for i in range(nseg)]
    return blocks, flat


def _bev_boxes(anno):
    return np.concatenate([anno["location"][:, [0, 2]], anno["dimensions"][:, [0, 2]], anno["rotation_y"][..., np.newaxis]], axis=1)


def _d3_boxes(anno):
    return np.concatenate([anno["location"], anno["dimensions"], anno["rotation_y"][..., np.newaxis]], axis=1)


def calculate_iou(dt_annos, gt_annos, metric, device_id=0):
    """Per-image overlap blocks (n_dt_i, n_gt_i) f64 for metric 0 image box / 1 BEV / 2 3D
    (eval2.py:352-427 called with (dt, gt) as at :492)."""
    assert len(gt_annos) == len(dt_annos)
    if metric == 0:
        return [image_box_overlap(d["bbox"], g["bbox"]) for d, g in zip(dt_annos, gt_annos)]
    if metric == 1:
        blocks, _ = rotate_iou_segmented([_bev_boxes(d) for d in dt_annos], [_bev_boxes(g) for g in gt_annos], -1, device_id)
        return [b.astype(np.float64) for b in blocks]
    if metric == 2:
        db, gb = [_d3_boxes(d) for d in dt_annos], [_d3_boxes(g) for g in gt_annos]
        blocks, _ = rotate_iou_segmented([b[:, [0, 2, 3, 5, 6]] for b in db], [b[:, [0, 2, 3, 5, 6]] for b in gb], 2, device_id)
        out = []
        for rinc, boxes, qboxes in zip(blocks, db, gb):
            # height overlap x BEV intersection / union volume, in f64 (d3_box_overlap_kernel, eval2.py:136-161);
            # y is the box bottom in the camera frame, the box extends to y - h
            rinc = rinc.astype(np.float64)
            iw = (np.minimum(boxes[:, None, 1], qboxes[None, :, 1]) -
                  np.maximum(boxes[:, None, 1] - boxes[:, None, 4], qboxes[None, :, 1] - qboxes[None, :, 4]))
            area1 = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
            area2 = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
            inc = iw * rinc
            with np.errstate(divide="ignore", invalid="ignore"):
                val = inc / (area1 + area2 - inc)
            out.append(np.where(rinc > 0, np.where(iw > 0, val, 0.0), rinc))
        return out
    raise ValueError("unknown metric")


# ---------------------------------------------------------------------------------------------------
# per-image bookkeeping
# ---------------------------------------------------------------------------------------------------
def clean_data(gt_anno, dt_anno, current_class, dataset, difficulty):
    """-> num_valid_gt, ignored_gt (n_gt) i64: 0 care / 1 ignore / -1 other class, ignored_dt (n_dt) i64, dc_bboxes (k, 4)
    (the per-image bookkeeping of evaluate/eval2.py:28-98, whole-array): a ground-truth box is CARED FOR when it carries the evaluated
    class name and passes the difficulty level's caps -- occlusion, truncation, and the distance band the reference uses instead of
    the official image-height rule; it is IGNORED (matches cost nothing) when it fails a cap or carries the neighbouring class
    (Van for Car, Person_sitting for Pedestrian); everything else is another class.  A detection outside the distance band is
    ignored, one of another class does not take part.  DontCare boxes are handed back for the false-positive exemption."""
    cls = CLASS_NAMES[current_class]
    near, far = DIST_BOUNDARY[0, difficulty], DIST_BOUNDARY[1, difficulty]
    raw = np.asarray(gt_anno["name"], dtype=str).reshape(-1)
    names = np.char.lower(raw) if raw.size else raw
    depth = np.asarray(gt_anno["location"], dtype=np.float64).reshape(-1, 3)[:, 2]
    capped = ((np.asarray(gt_anno["occluded"]).reshape(-1) > MAX_OCCLUSION[difficulty]) |
              (np.asarray(gt_anno["truncated"]).reshape(-1) > MAX_TRUNCATION[difficulty]) | ~((near < depth) & (depth < far)))
    own = names == cls
    sibling = names == {"pedestrian": "person_sitting", "car": "van"}.get(cls, "\0")
    ignored_gt = np.full(names.shape, -1, dtype=np.int64)
    ignored_gt[sibling | (own & capped)] = 1
    ignored_gt[own & ~capped] = 0
    dc_bboxes = np.asarray(gt_anno["bbox"], dtype=np.float64).reshape(-1, 4)[raw == "DontCare"]
    det = np.asarray(dt_anno["name"], dtype=str).reshape(-1)
    det_depth = np.asarray(dt_anno["location"], dtype=np.float64).reshape(-1, 3)[:, 2]
    in_band = (near < det_depth) & (det_depth < far)
    det_own = (np.char.lower(det) if det.size else det) == cls
    ignored_dt = np.where(in_band, np.where(det_own, 0, -1), 1).astype(np.int64)
    return int(np.count_nonzero(ignored_gt == 0)), ignored_gt, ignored_dt, dc_bboxes


def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """Score thresholds at (about) equally spaced recall positions -- the selection rule of evaluate/eval2.py:8-25 in closed form.
    With the true-positive scores in descending order, score i spans the recall interval [(i + 1) / num_gt, (i + 2) / num_gt] and
    the targets are c_t = t / (num_sample_pts - 1), accumulated by repeated addition as the reference does.  Target t may take score
    i unless the interval's far end is closer to the target than its near end (then a later score serves it better); the last score
    serves any target.  So a_t = the first score target t may take, and since every score is offered to one target only, the score
    taken for target t is i_t = max(a_t, i_(t-1) + 1), i.e. t + running max of (a_s - s): one comparison matrix, no loop."""
    s = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    n = len(s)
    if n == 0:
        return []
    step = 1 / (num_sample_pts - 1.0)
    n_targets = min(n, int(np.ceil((n + 2) / (num_gt * step))) + 2)                 # targets beyond recall n / num_gt all take the last score
    targets = np.concatenate([[0.0], np.cumsum(np.full(max(n_targets - 1, 0), step))])     # 0, step, step + step, ...: the reference's sums
    pos = np.arange(n)
    near_end, far_end = (pos + 1) / num_gt, np.where(pos < n - 1, (pos + 2) / num_gt, (pos + 1) / num_gt)
    later_is_better = (far_end[None, :] - targets[:, None]) < (targets[:, None] - near_end[None, :])
    later_is_better[:, n - 1] = False
    first_ok = np.argmin(later_is_better, axis=1)                                   # a_t (the last column is always admissible)
    t = np.arange(len(targets))
    taken = t + np.maximum.accumulate(first_ok - t)
    return list(s[taken[taken < n]])


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _Split:
    """Concatenated per-image arrays of one (class, difficulty) in the layout csrc/kitti_stats.hip reads."""

    def __init__(self, gt_annos, dt_annos, current_class, dataset, difficulty):
        ig, idt, dcs, gts, dts = [], [], [], [], []
        self.num_valid_gt = 0
        for g, d in zip(gt_annos, dt_annos):
            nv, ignored_gt, ignored_dt, dc = clean_data(g, d, current_class, dataset, difficulty)
            self.num_valid_gt += nv
            ig.append(ignored_gt)
            idt.append(ignored_dt)
            dcs.append(dc)
            gts.append(np.concatenate([g["bbox"], g["alpha"][..., np.newaxis]], 1).astype(np.float64).reshape(-1, 5))
            dts.append(np.concatenate([d["bbox"], d["alpha"][..., np.newaxis], d["score"][..., np.newaxis]], 1)
                       .astype(np.float64).reshape(-1, 6))
        cat = lambda xs, w, t: np.ascontiguousarray(np.concatenate(xs, 0) if xs else np.zeros((0, w) if w else (0,), t))
        self.gt_nums = np.array([len(x) for x in ig], dtype=np.int64)
        self.dt_nums = np.array([len(x) for x in idt], dtype=np.int64)
        self.dc_nums = np.array([len(x) for x in dcs], dtype=np.int64)
        self.ignored_gts, self.ignored_dets = cat(ig, 0, np.int64), cat(idt, 0, np.int64)
        self.dontcares, self.gt_datas, self.dt_datas = cat(dcs, 4, np.float64), cat(gts, 5, np.float64), cat(dts, 6, np.float64)


def eval_class(gt_annos, dt_annos, current_classes, dataset, difficultys, metric, min_overlaps, compute_aos=False,
               device_id=0, overlaps=None):
    """-> dict(recall, precision, orientation), each [num_class, num_difficulty, num_minoverlap, 41]
    (eval2.py:460-569).  min_overlaps: [num_minoverlap, metric, num_class]."""
    assert len(gt_annos) == len(dt_annos)
    n_img = len(gt_annos)
    if overlaps is None:
        overlaps = calculate_iou(dt_annos, gt_annos, metric, device_id)
    flat = np.ascontiguousarray(np.concatenate([o.reshape(-1) for o in overlaps]) if n_img else np.zeros((0,)), dtype=np.float64)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            sp = _Split(gt_annos, dt_annos, current_class, dataset, difficulty)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                scores = np.zeros((max(1, int(sp.gt_nums.sum())),), dtype=np.float64)
                n_scores = ctypes.c_longlong(0)
                _lib.call("prcnn_kitti_collect_scores", n_img, _ptr(sp.gt_nums), _ptr(sp.dt_nums), _ptr(flat),
                          _ptr(sp.gt_datas), _ptr(sp.dt_datas), _ptr(sp.ignored_gts), _ptr(sp.ignored_dets), int(metric),
                          float(min_overlap), _ptr(scores), ctypes.cast(ctypes.pointer(n_scores), ctypes.c_void_p))
                thresholds = np.array(get_thresholds(scores[:n_scores.value], sp.num_valid_gt), dtype=np.float64)
                pr = np.zeros([len(thresholds), 4])
                if len(thresholds):
                    _lib.call("prcnn_kitti_accumulate_pr", n_img, _ptr(sp.gt_nums), _ptr(sp.dt_nums), _ptr(sp.dc_nums),
                              _ptr(flat), _ptr(sp.gt_datas), _ptr(sp.dt_datas), _ptr(sp.dontcares), _ptr(sp.ignored_gts),
                              _ptr(sp.ignored_dets), int(metric), float(min_overlap), _ptr(thresholds), len(thresholds),
                              int(bool(compute_aos)), _ptr(pr))
                with np.errstate(divide="ignore", invalid="ignore"):
                    for i in range(len(thresholds)):
                        recall[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 2])
                        precision[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 1])
                        if compute_aos:
                            aos[m, l, k, i] = pr[i, 3] / (pr[i, 0] + pr[i, 1])
                for i in range(len(thresholds)):      # monotone envelope, right to left
                    precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                    recall[m, l, k, i] = np.max(recall[m, l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP(prec):
    """11-point interpolated AP in percent from the 41-sample precision curve: the samples at recall 0, 0.1, ..., 1 (every 4th one),
    added left to right (a cumulative sum: the reference's order of additions, evaluate/eval2.py:572-576), over 11."""
    return np.cumsum(prec[..., ::4], axis=-1)[..., -1] / 11 * 100


def do_eval(gt_annos, dt_annos, current_classes, dataset, min_overlaps, compute_aos=False, device_id=0):
    difficultys = [0, 1, 2, 3, 4, 5]
    ret = eval_class(gt_annos, dt_annos, current_classes, dataset, difficultys, 0, min_overlaps, compute_aos, device_id)
    mAP_bbox = get_mAP(ret["precision"])
    mAP_aos = get_mAP(ret["orientation"]) if compute_aos else None
    mAP_bev = get_mAP(eval_class(gt_annos, dt_annos, current_classes, dataset, difficultys, 1, min_overlaps,
                                 device_id=device_id)["precision"])
    mAP_3d = get_mAP(eval_class(gt_annos, dt_annos, current_classes, dataset, difficultys, 2, min_overlaps,
                                device_id=device_id)["precision"])
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


def _line(value):
    s = io.StringIO()
    print(value, file=s)
    return s.getvalue()


def get_official_eval_result(gt_annos, dt_annos, current_classes, dataset="kitti", dense_sample=False, device_id=0):
    """-> (result text, dict) in the reference's format (eval2.py:608-716)."""
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.7, 0.5, 0.5, 0.7, 0.5], [0.7, 0.5, 0.5, 0.7, 0.5]])
    overlap_0_5 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])
    extra = []
    if dense_sample:
        for i in range(101):
            tmp = np.zeros((3, 5))
            tmp[:, 0] = i / 100.0
            extra.append(tmp)
    min_overlaps = np.stack([overlap_0_7, overlap_0_5] + extra, axis=0)
    name_to_class = {v: n for n, v in CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = min_overlaps[:, :, current_classes]
    compute_aos = False
    for anno in dt_annos:
        if anno["alpha"].shape[0] != 0:
            if anno["alpha"][0] != -10:
                compute_aos = True
            break
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval(gt_annos, dt_annos, current_classes, dataset, min_overlaps, compute_aos, device_id)
    result = ""
    res = {}
    for j, curcls in enumerate(current_classes):
        res[curcls] = {}
        for i in range(min_overlaps.shape[0]):
            head = "%s AP@%.2f, %.2f, %.2f" % ((CLASS_TO_NAME[curcls],) + tuple(min_overlaps[i, :, j]))
            res[curcls][head] = {"mAPbbox": mAPbbox[j, :, i], "mAPbev": mAPbev[j, :, i], "mAP3d": mAP3d[j, :, i]}
            result += _line(head + ":")
            for tag, arr in (("bbox AP:", mAPbbox), ("bev  AP:", mAPbev), ("3d   AP:", mAP3d)):
                result += _line(tag + "".join("%.4f, " % arr[j, d, i] for d in range(6)))
            if compute_aos:
                result += _line("aos  AP:" + ", ".join("%.2f" % mAPaos[j, d, i] for d in range(6)))
    ret = {"result": res}
    for tag, arr in (("3d", mAP3d), ("bev", mAPbev), ("image", mAPbbox)):
        for d, name in enumerate(("easy", "moderate", "hard")):
            ret["Car_%s_%s" % (tag, name)] = arr[0, d, 0]
    return result, ret


def evaluate(result_path, label_path, image_ids, current_class=0, dataset="kitti", score_thresh=-1, device_id=0):
    """Result folder + label folder + id list -> (result text, dict) (evaluate.py:88-135, plain branch)."""
    dt_annos = get_label_annos(result_path, list(image_ids))
    if score_thresh > 0:
        dt_annos = filter_annos_low_score(dt_annos, score_thresh)
    gt_annos = get_label_annos(label_path, list(image_ids))
    return get_official_eval_result(gt_annos, dt_annos, current_class, dataset, device_id=device_id)
