// kitti_stats.hip -- host side of the KITTI AP evaluator: the per-image greedy matching and the
// precision/recall accumulation (evaluate/eval2.py:170-349; numba CPU jit in the reference), plus the
// segmented launch of the rotated-IoU kernel that replaces the reference's 50 "parts".
//
// Host code only (no device kernels here): the matching is a short sequential scan per image with
// data-dependent control flow -- it stays on the CPU in the reference too -- but it is called
// images x difficulties x overlap levels x 42 times per metric, so it lives in the library instead of
// a Python loop.  All arithmetic is f64, as numba types it from the f64 label arrays.
#include "common.hpp"
#include <math.h>
#include <vector>

namespace prcnn {

// image_box_overlap(boxes, query, criterion = 0) of one pair (eval2.py:102-128): intersection / area(box)
static double overlap_over_box_area(const double *box, const double *q)
{
    const double iw = fmin(box[2], q[2]) - fmax(box[0], q[0]);
    if (!(iw > 0)) return 0.0;
    const double ih = fmin(box[3], q[3]) - fmax(box[1], q[1]);
    if (!(ih > 0)) return 0.0;
    const double ua = (box[2] - box[0]) * (box[3] - box[1]);
    return iw * ih / ua;
}

struct ImageStats {
    long long tp, fp, fn;
    double similarity;
    int n_thresholds;
};

// compute_statistics_jit (eval2.py:170-289).  overlaps is (n_dt, n_gt) row-major; gt_datas (n_gt,5) =
// [bbox x4, alpha]; dt_datas (n_dt,6) = [bbox x4, alpha, score]; thresholds_out has room for n_gt values.
static ImageStats image_stats(int n_gt, int n_dt, int n_dc, const double *overlaps, const double *gt_datas,
                              const double *dt_datas, const long long *ignored_gt, const long long *ignored_det,
                              const double *dc_bboxes, int metric, double min_overlap, double thresh, bool compute_fp,
                              bool compute_aos, double *thresholds_out, std::vector<char> &assigned,
                              std::vector<char> &below, std::vector<double> &delta)
{
    assigned.assign((size_t)n_dt, 0);
    below.assign((size_t)n_dt, 0);
    delta.clear();
    if (compute_fp)
        for (int j = 0; j < n_dt; ++j) below[j] = dt_datas[6 * j + 5] < thresh;
    const double NO_DETECTION = -10000000.0;
    ImageStats st = {0, 0, 0, 0.0, 0};
    for (int i = 0; i < n_gt; ++i) {
        if (ignored_gt[i] == -1) continue;
        int det_idx = -1;
        double valid_detection = NO_DETECTION, max_overlap = 0.0;
        bool assigned_ignored_det = false;
        for (int j = 0; j < n_dt; ++j) {
            if (ignored_det[j] == -1 || assigned[j] || below[j]) continue;
            const double overlap = overlaps[(size_t)j * n_gt + i];
            const double score = dt_datas[6 * j + 5];
            if (!compute_fp && overlap > min_overlap && score > valid_detection) {
                det_idx = j;
                valid_detection = score;
            } else if (compute_fp && overlap > min_overlap && (overlap > max_overlap || assigned_ignored_det) &&
                       ignored_det[j] == 0) {
                max_overlap = overlap;
                det_idx = j;
                valid_detection = 1;
                assigned_ignored_det = false;
            } else if (compute_fp && overlap > min_overlap && valid_detection == NO_DETECTION && ignored_det[j] == 1) {
                det_idx = j;
                valid_detection = 1;
                assigned_ignored_det = true;
            }
        }
        if (valid_detection == NO_DETECTION && ignored_gt[i] == 0) {
            st.fn += 1;
        } else if (valid_detection != NO_DETECTION && (ignored_gt[i] == 1 || ignored_det[det_idx] == 1)) {
            assigned[det_idx] = 1;
        } else if (valid_detection != NO_DETECTION) {
            st.tp += 1;
            if (thresholds_out) thresholds_out[st.n_thresholds] = dt_datas[6 * det_idx + 5];
            st.n_thresholds += 1;
            if (compute_aos) delta.push_back(gt_datas[5 * i + 4] - dt_datas[6 * det_idx + 4]);
            assigned[det_idx] = 1;
        }
    }
    if (compute_fp) {
        for (int j = 0; j < n_dt; ++j)
            if (!(assigned[j] || ignored_det[j] == -1 || ignored_det[j] == 1 || below[j])) st.fp += 1;
        long long nstuff = 0;
        if (metric == 0) {
            for (int i = 0; i < n_dc; ++i)
                for (int j = 0; j < n_dt; ++j) {
                    if (assigned[j] || ignored_det[j] == -1 || ignored_det[j] == 1 || below[j]) continue;
                    if (overlap_over_box_area(dt_datas + 6 * j, dc_bboxes + 4 * i) > min_overlap) {
                        assigned[j] = 1;
                        nstuff += 1;
                    }
                }
        }
        st.fp -= nstuff;
        if (compute_aos) {
            if (st.tp > 0 || st.fp > 0) {
                double s = 0.0;   // fp zeros first, then (1 + cos(delta)) / 2 per true positive, summed in that order
                for (double d : delta) s += (1.0 + cos(d)) / 2.0;
                st.similarity = s;
            } else {
                st.similarity = -1;
            }
        }
    }
    return st;
}

}  // namespace prcnn

using namespace prcnn;

extern "C" int prcnn_kitti_image_stats(int n_gt, int n_dt, int n_dc, const double *overlaps, const double *gt_datas,
                                       const double *dt_datas, const long long *ignored_gt, const long long *ignored_det,
                                       const double *dc_bboxes, int metric, double min_overlap, double thresh,
                                       int compute_fp, int compute_aos, long long *tp_fp_fn, double *similarity,
                                       double *thresholds, int *n_thresholds)
{
    PRCNN_REQUIRE(n_gt >= 0 && n_dt >= 0 && n_dc >= 0, "kitti_image_stats: bad sizes");
    PRCNN_REQUIRE(tp_fp_fn && similarity && n_thresholds, "kitti_image_stats: null output");
    PRCNN_REQUIRE((n_gt == 0 || (gt_datas && ignored_gt && thresholds)) && (n_dt == 0 || (dt_datas && ignored_det)) &&
                      (n_gt == 0 || n_dt == 0 || overlaps) && (n_dc == 0 || dc_bboxes),
                  "kitti_image_stats: null input");
    std::vector<char> assigned, below;
    std::vector<double> delta;
    const ImageStats st = image_stats(n_gt, n_dt, n_dc, overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes,
                                      metric, min_overlap, thresh, compute_fp != 0, compute_aos != 0, thresholds, assigned,
                                      below, delta);
    tp_fp_fn[0] = st.tp; tp_fp_fn[1] = st.fp; tp_fp_fn[2] = st.fn;
    *similarity = st.similarity;
    *n_thresholds = st.n_thresholds;
    return PRCNN_OK;
}

// First pass of eval_class (eval2.py:512-527): scores of the matched detections of every image
// (compute_fp = False, thresh = 0), concatenated into scores_out (room for sum(gt_nums)); *n_scores = count.
extern "C" int prcnn_kitti_collect_scores(int n_img, const long long *gt_nums, const long long *dt_nums,
                                          const double *overlaps_flat, const double *gt_datas, const double *dt_datas,
                                          const long long *ignored_gts, const long long *ignored_dets, int metric,
                                          double min_overlap, double *scores_out, long long *n_scores)
{
    PRCNN_REQUIRE(n_img >= 0 && n_scores, "kitti_collect_scores: bad arguments");
    std::vector<char> assigned, below;
    std::vector<double> delta;
    size_t gt0 = 0, dt0 = 0, ov0 = 0;
    long long total = 0;
    for (int i = 0; i < n_img; ++i) {
        const int ng = (int)gt_nums[i], nd = (int)dt_nums[i];
        const ImageStats st = image_stats(ng, nd, 0, overlaps_flat + ov0, gt_datas + 5 * gt0, dt_datas + 6 * dt0,
                                          ignored_gts + gt0, ignored_dets + dt0, nullptr, metric, min_overlap, 0.0, false,
                                          false, scores_out + total, assigned, below, delta);
        total += st.n_thresholds;
        gt0 += ng; dt0 += nd; ov0 += (size_t)ng * nd;
    }
    *n_scores = total;
    return PRCNN_OK;
}

// fused_compute_statistics (eval2.py:300-349) over all images: pr (n_thresh, 4) += [tp, fp, fn, similarity].
extern "C" int prcnn_kitti_accumulate_pr(int n_img, const long long *gt_nums, const long long *dt_nums,
                                         const long long *dc_nums, const double *overlaps_flat, const double *gt_datas,
                                         const double *dt_datas, const double *dontcares, const long long *ignored_gts,
                                         const long long *ignored_dets, int metric, double min_overlap,
                                         const double *thresholds, int n_thresh, int compute_aos, double *pr)
{
    PRCNN_REQUIRE(n_img >= 0 && n_thresh >= 0 && (n_thresh == 0 || (thresholds && pr)), "kitti_accumulate_pr: bad arguments");
    std::vector<char> assigned, below;
    std::vector<double> delta, scratch;
    size_t gt0 = 0, dt0 = 0, dc0 = 0, ov0 = 0;
    for (int i = 0; i < n_img; ++i) {
        const int ng = (int)gt_nums[i], nd = (int)dt_nums[i], nc = (int)dc_nums[i];
        scratch.resize((size_t)ng + 1);
        for (int t = 0; t < n_thresh; ++t) {
            const ImageStats st = image_stats(ng, nd, nc, overlaps_flat + ov0, gt_datas + 5 * gt0, dt_datas + 6 * dt0,
                                              ignored_gts + gt0, ignored_dets + dt0, dontcares + 4 * dc0, metric, min_overlap,
                                              thresholds[t], true, compute_aos != 0, scratch.data(), assigned, below, delta);
            pr[4 * t + 0] += (double)st.tp;
            pr[4 * t + 1] += (double)st.fp;
            pr[4 * t + 2] += (double)st.fn;
            if (st.similarity != -1) pr[4 * t + 3] += st.similarity;
        }
        gt0 += ng; dt0 += nd; dc0 += nc; ov0 += (size_t)ng * nd;
    }
    return PRCNN_OK;
}
