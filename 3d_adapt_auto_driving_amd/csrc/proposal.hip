// proposal.hip -- the RPN proposal layer as five launches (counterpart of
// pointrcnn/lib/rpn/proposal_layer.py:15-119 with decode_bbox_target of lib/utils/bbox_transform.py:24-121).
//
// The reference does this per scene in Python: ~25 elementwise torch kernels for the decode, a sort,
// boolean-mask selections with two device->host syncs, and two NMS calls with host round trips; the
// batched torch formulation in net/proposal_layer.py still needs ~90 short launches.  Here:
//   1. rpn_decode_kernel     one lane per point: arg-max of the x / z / heading bins, residuals, box
//                            (same f32 operation order as the torch expression, no fma)
//   2. score_sort_kernel     one workgroup per scene: bitonic sort of (score desc, index asc) keys in LDS
//   3. band_select_kernel    one workgroup per scene: ordered compaction (wave ballot + LDS prefix) of the
//                            sorted proposals into the (0,40] m and (40,80] m tables, top 70 % / 30 % of
//                            RPN_PRE_NMS_TOP_N, with the reference's "no far points" fallback
//   4. nms_lazy_kernel       (iou3d.hip) all 2*B problems in one launch, stops at the post-NMS quota
//   5. assemble_rois_kernel  near keeps, then far keeps, zero padded -> rois (B, M, 7), scores (B, M)
// No host synchronisation anywhere.
#include "common.hpp"
#include "rbox_iou.hpp"
#include <stdlib.h>
#include <math.h>

namespace prcnn {

int nms_device(int nprob, int n_max, const int *counts, const float *boxes, float thresh, int rotated,
               int max_keep, int *keep, int *num_keep, hipStream_t st);   // iou3d.hip

struct DecodeCfg {
    float loc_scope, loc_bin_size;
    int nbin;          // per_loc_bin_num = 2 * int(loc_scope / loc_bin_size)
    int num_head_bin;
    int xz_fine;
    float anchor[3];
    int channels;
};

__device__ __forceinline__ int argmax_row(const float *p, int n)
{
    int best = 0;
    float bv = p[0];
    for (int i = 1; i < n; ++i)                    // first maximum; a NaN counts as the largest value (torch.argmax)
        if (p[i] > bv || (p[i] != p[i] && bv == bv)) { bv = p[i]; best = i; }
    return best;
}

// torch.remainder for f32: fmod, then shifted into the sign of the divisor
__device__ __forceinline__ float torch_remainder(float a, float b)
{
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}

// decode_bbox_target(xyz (N,3), reg, get_xz_fine, get_y_by_bin=False, get_ry_fine=False), then
// proposals[:, 1] += proposals[:, 3] / 2 (proposal_layer.py:31)
__global__ __launch_bounds__(256) void rpn_decode_kernel(long total, DecodeCfg c, const float *__restrict__ xyz,
                                                         const float *__restrict__ reg, float *__restrict__ boxes)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float *r = reg + i * c.channels;
    const float *p = xyz + i * 3;
    const int nb = c.nbin;
    const int xb = argmax_row(r, nb), zb = argmax_row(r + nb, nb);
    const float half_bin = c.loc_bin_size / 2;
    float px = __fsub_rn(__fadd_rn(__fmul_rn((float)xb, c.loc_bin_size), half_bin), c.loc_scope);
    float pz = __fsub_rn(__fadd_rn(__fmul_rn((float)zb, c.loc_bin_size), half_bin), c.loc_scope);
    int cur = 2 * nb;
    if (c.xz_fine) {
        px = __fadd_rn(px, __fmul_rn(r[2 * nb + xb], c.loc_bin_size));
        pz = __fadd_rn(pz, __fmul_rn(r[3 * nb + zb], c.loc_bin_size));
        cur = 4 * nb;
    }
    float py = __fadd_rn(p[1], r[cur]);
    cur += 1;
    const int rb = argmax_row(r + cur, c.num_head_bin);
    const float res_norm = r[cur + c.num_head_bin + rb];
    const float apc = (float)((2.0 * M_PI) / c.num_head_bin);            // python double -> f32 scalar
    const float apc_half = (float)(((2.0 * M_PI) / c.num_head_bin) / 2.0);
    const float two_pi = (float)(2.0 * M_PI), pi = (float)M_PI;
    float ry = torch_remainder(__fadd_rn(__fmul_rn((float)rb, apc), __fmul_rn(res_norm, apc_half)), two_pi);
    if (ry > pi) ry = __fsub_rn(ry, two_pi);
    cur += 2 * c.num_head_bin;
    const float h = __fadd_rn(__fmul_rn(r[cur], c.anchor[0]), c.anchor[0]);
    const float w = __fadd_rn(__fmul_rn(r[cur + 1], c.anchor[1]), c.anchor[1]);
    const float l = __fadd_rn(__fmul_rn(r[cur + 2], c.anchor[2]), c.anchor[2]);
    float *o = boxes + i * 7;
    o[0] = __fadd_rn(px, p[0]);
    o[1] = __fadd_rn(py, h / 2);
    o[2] = __fadd_rn(pz, p[2]);
    o[3] = h; o[4] = w; o[5] = l; o[6] = ry;
}

// key: descending score, ties by ascending index
__device__ __forceinline__ unsigned long long sort_key(float score, unsigned idx)
{
    unsigned u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-order image of the float
    if (score != score) u = 0xffffffffu;              // NaN of either sign sorts FIRST, as in torch.sort(descending=True)
    return ((unsigned long long)(~u) << 32) | idx;     // inverted: larger score sorts first
}

// one workgroup per scene; n padded to npad = 2^k <= 16384 with +inf keys; writes order (b, n) i32
__global__ __launch_bounds__(1024) void score_sort_kernel(int n, int npad, const float *__restrict__ scores,
                                                          int *__restrict__ order)
{
    extern __shared__ unsigned long long keys[];
    const int b = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < npad; i += 1024)
        keys[i] = i < n ? sort_key(scores[(long)b * n + i], (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j >= 1; j >>= 1) {
            for (int i = t; i < npad / 2; i += 1024) {
                const int lo = ((i / j) * 2 * j) + (i % j), hi = lo + j;   // pair (lo, lo + j)
                const bool up = ((lo & k) == 0);
                const unsigned long long a = keys[lo], c = keys[hi];
                if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = t; i < n; i += 1024) order[(long)b * n + i] = (int)(keys[i] & 0xffffffffu);
}


// ---- the same order from several workgroups per scene (round 3) ------------------------------------------------------------
// score_sort_kernel is one workgroup per scene: 105 bitonic passes of 8 rounds each over 16384 keys, 183 us on 8 CUs while the
// proposal stream waits.  Here every 4096-key chunk of a scene is sorted by its own workgroup (78 passes of 2
// rounds, 32 KB of LDS), and the sorted runs are merged pairwise by merge-path kernels (a thread finds its 8 outputs' split point
// by binary search and merges them): 4 x 4096 -> 2 x 8192 -> 16384.  Keys are distinct (the index is part of the key), so the
// result is THE sorted order: identical to the one-workgroup kernel's (tests/test_gpu_e2e.py compares the two).
constexpr int SS_CHUNK = 4096;
constexpr int SS_MERGE_V = 8;

__global__ __launch_bounds__(1024) void score_sort_chunk_kernel(int n, int npad, const float *__restrict__ scores,
                                                                unsigned long long *__restrict__ keys_out)
{
    __shared__ unsigned long long keys[SS_CHUNK];
    const int b = blockIdx.y, c0 = blockIdx.x * SS_CHUNK, t = threadIdx.x;
    for (int i = t; i < SS_CHUNK; i += 1024) {
        const int g = c0 + i;
        keys[i] = g < n ? sort_key(scores[(long)b * n + g], (unsigned)g) : ~0ull;
    }
    __syncthreads();
    for (int k = 2; k <= SS_CHUNK; k <<= 1)
        for (int j = k >> 1; j >= 1; j >>= 1) {
            for (int i = t; i < SS_CHUNK / 2; i += 1024) {
                const int lo = ((i / j) * 2 * j) + (i % j), hi = lo + j;
                const bool up = ((lo & k) == 0);
                const unsigned long long a = keys[lo], c = keys[hi];
                if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = t; i < SS_CHUNK; i += 1024) keys_out[(long)b * npad + c0 + i] = keys[i];
}

// src: per scene npad keys = sorted runs of `run` keys; dst <- runs of 2 * run (or, last round, order (b, n) i32 <- the indices)
__global__ __launch_bounds__(256) void score_merge_kernel(int n, int npad, int run, const unsigned long long *__restrict__ src,
                                                          unsigned long long *__restrict__ dst, int *__restrict__ order)
{
    const int b = blockIdx.y;
    const int g0 = (blockIdx.x * 256 + threadIdx.x) * SS_MERGE_V;
    if (g0 >= npad) return;
    const int pair = g0 / (2 * run), d = g0 - pair * 2 * run;
    const unsigned long long *__restrict__ A = src + (long)b * npad + (long)pair * 2 * run;
    const unsigned long long *__restrict__ B = A + run;
    int lo = max(0, d - run), hi = min(d, run);            // how many of the first d outputs come from A
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A[mid] < B[d - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    int ia = lo, ib = d - lo;
    unsigned long long va = ia < run ? A[ia] : ~0ull, vb = ib < run ? B[ib] : ~0ull;
#pragma unroll
    for (int q = 0; q < SS_MERGE_V; ++q) {
        // exhausted runs read as the all-ones pad key, which also ends every run's tail: take A on equality (only pads are equal)
        const bool from_a = ib >= run || (ia < run && va <= vb);
        const unsigned long long v = from_a ? va : vb;
        if (from_a) { ++ia; va = ia < run ? A[ia] : ~0ull; } else { ++ib; vb = ib < run ? B[ib] : ~0ull; }
        const int g = g0 + q;
        if (order) { if (g < n) order[(long)b * n + g] = (int)(v & 0xffffffffu); }
        else dst[(long)b * npad + g] = v;
    }
}

// tables: payload (b, 2, rows, 8) = box7 + score, bev (b, 2, rows, 5), counts (b, 2)
__global__ __launch_bounds__(1024) void band_select_kernel(
    int n, int rows, int pre_near, int pre_far, const float *__restrict__ boxes, const float *__restrict__ scores,
    const int *__restrict__ order, float *__restrict__ payload, float *__restrict__ bev, int *__restrict__ counts)
{
    __shared__ int wcnt[2][16];
    __shared__ int tot[2];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *__restrict__ bx = boxes + (long)b * n * 7;
    const float *__restrict__ sc = scores + (long)b * n;
    const int *__restrict__ ord = order + (long)b * n;
    float *__restrict__ pay = payload + (long)b * 2 * rows * 8;
    float *__restrict__ bv = bev + (long)b * 2 * rows * 5;
    if (t < 2) tot[t] = 0;
    __syncthreads();

    auto emit = [&](int band, int slot, int k) {
        float *o = pay + ((long)band * rows + slot) * 8;
        const float *q = bx + (long)k * 7;
#pragma unroll
        for (int e = 0; e < 7; ++e) o[e] = q[e];
        o[7] = sc[k];
        float *v = bv + ((long)band * rows + slot) * 5;
        const float hl = q[5] / 2, hw = q[4] / 2;               // boxes3d_to_bev_torch (kitti_utils.py:134-147)
        v[0] = q[0] - hl; v[1] = q[2] - hw; v[2] = q[0] + hl; v[3] = q[2] + hw; v[4] = q[6];
    };

    // pass 0: near band gets near ranks [0, pre_near), far band far ranks [0, pre_far)
    // pass 1 (only if the scene has no far point): far band gets near ranks [pre_near, pre_near + pre_far)
    for (int pass = 0; pass < 2; ++pass) {
        int near_seen = 0, far_seen = 0;
        for (int base = 0; base < n; base += 1024) {
            if (pass == 0 && near_seen >= pre_near && far_seen >= pre_far) break;
            if (pass == 1 && near_seen >= pre_near + pre_far) break;
            const int s = base + t;
            int k = -1;
            bool is_near = false, is_far = false;
            if (s < n) {
                k = ord[s];
                const float dist = bx[(long)k * 7 + 2];
                is_near = dist > 0.f && dist <= 40.0f;
                is_far = dist > 40.0f && dist <= 80.0f;
            }
            const unsigned long long mn = __ballot(is_near), mf = __ballot(is_far);
            if (lane == 0) { wcnt[0][w] = __popcll(mn); wcnt[1][w] = __popcll(mf); }
            __syncthreads();
            int nb = near_seen, fb = far_seen, nt = 0, ft = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q < w) { nb += wcnt[0][q]; fb += wcnt[1][q]; }
                nt += wcnt[0][q]; ft += wcnt[1][q];
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (is_near) {
                const int rank = nb + __popcll(mn & below);
                if (pass == 0 && rank < pre_near) emit(0, rank, k);
                if (pass == 1 && rank >= pre_near && rank < pre_near + pre_far) emit(1, rank - pre_near, k);
            }
            if (is_far && pass == 0) {
                const int rank = fb + __popcll(mf & below);
                if (rank < pre_far) emit(1, rank, k);
            }
            near_seen += nt; far_seen += ft;
            __syncthreads();
        }
        if (pass == 0) {
            if (t == 0) { tot[0] = near_seen; tot[1] = far_seen; }
            __syncthreads();
            // the scan stops early once both quotas are full, so "no far point" is only known if it ran
            // to the end; a scene with an empty far band never fills the far quota, hence it did
            if (tot[1] != 0) {
                if (t == 0) { counts[b * 2] = min(tot[0], pre_near); counts[b * 2 + 1] = min(tot[1], pre_far); }
                return;
            }
        } else if (t == 0) {
            counts[b * 2] = min(tot[0], pre_near);
            counts[b * 2 + 1] = max(0, min(near_seen, pre_near + pre_far) - pre_near);
        }
    }
}

__global__ __launch_bounds__(128) void assemble_rois_kernel(
    int rows, int keep_stride, int post_near, int post_far, const float *__restrict__ payload,
    const int *__restrict__ keep, const int *__restrict__ num, float *__restrict__ rois, float *__restrict__ scores)
{
    const int b = blockIdx.x, j = threadIdx.x;
    const int post = post_near + post_far;
    if (j >= post) return;
    const int kn = min(num[b * 2], post_near), kf = min(num[b * 2 + 1], post_far);
    float *o = rois + ((long)b * post + j) * 7;
    int band = -1, slot = 0;
    if (j < kn) { band = 0; slot = keep[(b * 2) * keep_stride + j]; }
    else if (j < kn + kf) { band = 1; slot = keep[(b * 2 + 1) * keep_stride + (j - kn)]; }
    if (band < 0) {
#pragma unroll
        for (int e = 0; e < 7; ++e) o[e] = 0.f;
        scores[(long)b * post + j] = 0.f;
        return;
    }
    const float *q = payload + (((long)b * 2 + band) * rows + slot) * 8;
#pragma unroll
    for (int e = 0; e < 7; ++e) o[e] = q[e];
    scores[(long)b * post + j] = q[7];
}

// ---- final stage (eval_rcnn.py:506-530, 611-629): decode the RCNN regression against its RoI, score
// threshold, order by raw score, BEV boxes for the rotated NMS.  One workgroup (128 lanes) per scene, one
// lane per RoI; the <= 128 scores are sorted with a bitonic network in LDS.
struct RcnnCfg {
    float loc_scope, loc_bin_size, loc_y_scope, loc_y_bin_size, score_thresh;
    int nbin, nbin_y, num_head_bin, y_by_bin, channels;
    float anchor[3];
};

// decode_bbox_target of one RoI (bbox_transform.py:30-127 with get_xz_fine = get_ry_fine = True): r = its regression row, roi = its box
__device__ __forceinline__ void rcnn_decode_box(const RcnnCfg &c, const float *__restrict__ r, const float *__restrict__ roi, float (&box)[7])
{
    const int nb = c.nbin;
    const int xb = argmax_row(r, nb), zb = argmax_row(r + nb, nb);
    const float half_bin = c.loc_bin_size / 2;
    float px = __fsub_rn(__fadd_rn(__fmul_rn((float)xb, c.loc_bin_size), half_bin), c.loc_scope);
    float pz = __fsub_rn(__fadd_rn(__fmul_rn((float)zb, c.loc_bin_size), half_bin), c.loc_scope);
    px = __fadd_rn(px, __fmul_rn(r[2 * nb + xb], c.loc_bin_size));      // get_xz_fine = True
    pz = __fadd_rn(pz, __fmul_rn(r[3 * nb + zb], c.loc_bin_size));
    int cur = 4 * nb;
    float py;
    if (c.y_by_bin) {
        const int yb = argmax_row(r + cur, c.nbin_y);
        const float y_res = __fmul_rn(r[cur + c.nbin_y + yb], c.loc_y_bin_size);
        py = __fadd_rn(__fsub_rn(__fadd_rn(__fmul_rn((float)yb, c.loc_y_bin_size), c.loc_y_bin_size / 2), c.loc_y_scope), y_res);
        py = __fadd_rn(py, roi[1]);
        cur += 2 * c.nbin_y;
    } else {
        py = __fadd_rn(roi[1], r[cur]);
        cur += 1;
    }
    const int rb = argmax_row(r + cur, c.num_head_bin);
    const float res_norm = r[cur + c.num_head_bin + rb];
    // get_ry_fine = True: bins over +-pi/4 around the RoI heading
    const float apc = (float)((M_PI / 2) / c.num_head_bin);
    const float apc_half = (float)(((M_PI / 2) / c.num_head_bin) / 2.0);
    float ry = __fsub_rn(__fadd_rn(__fadd_rn(__fmul_rn((float)rb, apc), apc_half), __fmul_rn(res_norm, apc_half)), (float)(M_PI / 4));
    cur += 2 * c.num_head_bin;
    const float h = __fadd_rn(__fmul_rn(r[cur], c.anchor[0]), c.anchor[0]);
    const float w = __fadd_rn(__fmul_rn(r[cur + 1], c.anchor[1]), c.anchor[1]);
    const float l = __fadd_rn(__fmul_rn(r[cur + 2], c.anchor[2]), c.anchor[2]);
    // rotate (px, pz) by -roi_ry back to the scene frame (rotate_pc_along_y_torch with -ry), add the RoI centre
    const float roi_ry = roi[6];
    const float cosa = cosf(-roi_ry), sina = sinf(-roi_ry);
    const float nx = __fadd_rn(__fmul_rn(px, cosa), __fmul_rn(pz, -sina));
    const float nz = __fadd_rn(__fmul_rn(px, sina), __fmul_rn(pz, cosa));
    box[0] = __fadd_rn(nx, roi[0]); box[1] = py; box[2] = __fadd_rn(nz, roi[2]);
    box[3] = h; box[4] = w; box[5] = l; box[6] = __fadd_rn(ry, roi_ry);
}

__global__ __launch_bounds__(128) void rcnn_decode_select_kernel(
    int m, RcnnCfg c, const float *__restrict__ rois, const float *__restrict__ reg, const float *__restrict__ cls,
    float *__restrict__ pred /* (b,m,7) decoded, RoI order */, float *__restrict__ sorted /* (b,m,8) box7+raw, score order */,
    float *__restrict__ bev /* (b,m,5) */, int *__restrict__ counts)
{
    __shared__ unsigned long long keys[128];
    __shared__ int nsel;
    const int b = blockIdx.x, t = threadIdx.x;
    float box[7] = {0, 0, 0, 0, 0, 0, 0};
    float raw = 0.f;
    bool selected = false;
    if (t == 0) nsel = 0;
    __syncthreads();
    if (t < m) {
        rcnn_decode_box(c, reg + ((long)b * m + t) * c.channels, rois + ((long)b * m + t) * 7, box);
        raw = cls[(long)b * m + t];
        const float norm = 1.0f / (1.0f + expf(-raw));
        selected = norm > c.score_thresh;
        float *o = pred + ((long)b * m + t) * 7;
#pragma unroll
        for (int e = 0; e < 7; ++e) o[e] = box[e];
    }
    // selected RoIs first, by descending raw score; unselected after them
    keys[t] = (t < m && selected) ? sort_key(raw, (unsigned)t) : (~0ull - 127 + t);
    if (t < m && selected) atomicAdd(&nsel, 1);
    __syncthreads();
    for (int k = 2; k <= 128; k <<= 1)
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const int partner = t ^ j;
            if (partner > t) {
                const bool up = ((t & k) == 0);
                const unsigned long long a = keys[t], d = keys[partner];
                if ((a > d) == up) { keys[t] = d; keys[partner] = a; }
            }
            __syncthreads();
        }
    // slot t of the sorted table takes the box lane `src` decoded (its global write is ordered by the barriers above)
    if (t < m) {
        const unsigned long long kk = keys[t];
        const bool sel_slot = t < nsel;
        const int src = sel_slot ? (int)(kk & 0xffffffffu) : -1;
        float *o = sorted + ((long)b * m + t) * 8;
        float *v = bev + ((long)b * m + t) * 5;
        if (src >= 0) {
            const float *q = pred + ((long)b * m + src) * 7;
            o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3]; o[4] = q[4]; o[5] = q[5]; o[6] = q[6];
            o[7] = cls[(long)b * m + src];
            const float hl = q[5] / 2, hw = q[4] / 2;
            v[0] = q[0] - hl; v[1] = q[2] - hw; v[2] = q[0] + hl; v[3] = q[2] + hw; v[4] = q[6];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
            for (int e = 0; e < 5; ++e) v[e] = 0.f;
        }
    }
    if (t == 0) counts[b] = nsel;
}

__global__ __launch_bounds__(128) void rcnn_final_gather_kernel(int m, const float *__restrict__ sorted,
                                                               const int *__restrict__ keep, const int *__restrict__ num,
                                                               float *__restrict__ boxes, float *__restrict__ scores)
{
    const int b = blockIdx.x, j = threadIdx.x;
    if (j >= m) return;
    float *o = boxes + ((long)b * m + j) * 7;
    if (j < num[b]) {
        const float *q = sorted + ((long)b * m + keep[(long)b * m + j]) * 8;
#pragma unroll
        for (int e = 0; e < 7; ++e) o[e] = q[e];
        scores[(long)b * m + j] = q[7];
    } else {
#pragma unroll
        for (int e = 0; e < 7; ++e) o[e] = 0.f;
        scores[(long)b * m + j] = 0.f;
    }
}

// ---- the whole final stage of a scene in ONE workgroup (round 4).  The four launches above and in iou3d.hip (decode + sort, all-pairs
// suppression mask, resolve, gather: 88 us of the feature stream per step, most of it launch gaps and half-empty waves) become one:
//   decode and score-sort as rcnn_decode_select_kernel (same code), the sorted BEV rows stay in LDS;
//   CANDIDATE pairs: every (row r, column c > r) whose boxes can touch at all (rbox_far_apart: bounding circles) goes onto a list in
//     LDS -- of the 4950 pairs of 100 boxes a few hundred; the rotated IoU (~600 VALU instructions, divergent) then runs over the
//     compacted list only, full waves;
//   the greedy resolve over the row masks (iou3d.cpp:100-119) by one thread, then the gather of the kept boxes.
// Same expressions in the same (row, column) order as nms_dense_mask_kernel: the same keep list (a pair the filter drops has overlap
// exactly 0 in the reference's arithmetic as well -- no vertex of the clipped polygon exists).
constexpr int RF_THREADS = 256, RF_MAX = 128;

__global__ __launch_bounds__(RF_THREADS) void rcnn_final_kernel(
    int m, RcnnCfg c, float nms_thresh, const float *__restrict__ rois, const float *__restrict__ reg, const float *__restrict__ cls,
    float *__restrict__ pred /* (b,m,7) decoded, RoI order */, float *__restrict__ boxes /* (b,m,7) */, float *__restrict__ scores /* (b,m) */,
    int *__restrict__ num /* (b) */, int spb /* > 0: boxes = a sequence of BLOBS of spb scenes each, [spb m 7 boxes | spb m scores | spb num] */)
{
    if (spb > 0) {
        // (round 5) one blob per batch of spb scenes inside a launch over several batches: each batch's detections leave the device
        // with ONE copy (a pair of batches per launch used to cost three copies per batch)
        const int bl = blockIdx.x / spb, si = blockIdx.x - bl * spb;
        float *base = boxes + (long)bl * spb * (m * 8 + 1);
        // re-based so that the scene indexing below (b = blockIdx.x) lands on this scene's slices
        boxes = base + ((long)si - blockIdx.x) * m * 7;
        scores = base + (long)spb * m * 7 + ((long)si - blockIdx.x) * m;
        num = reinterpret_cast<int *>(base + (long)spb * m * 8) + (si - (int)blockIdx.x);
    }
    __shared__ unsigned long long keys[RF_MAX];
    __shared__ unsigned long long s_mask[RF_MAX][2];
    __shared__ float s_box[RF_MAX * 8];                  // decoded box + raw score, RoI order
    __shared__ float s_row[RF_MAX * 7];                  // BEV box + cos, sin of the heading, score order
    __shared__ int s_src[RF_MAX], s_keep[RF_MAX];
    __shared__ unsigned int s_pair[RF_MAX * (RF_MAX - 1) / 2];      // (r << 16) | c
    __shared__ int nsel, s_npair, s_nkeep;
    __shared__ float s_poly[POLY_LDS_FLOATS * RF_THREADS];          // the clipped polygons (72 KB): see PolyLds
    const int b = blockIdx.x, t = threadIdx.x;
    float raw = 0.f;
    bool selected = false;
    if (t == 0) { nsel = 0; s_npair = 0; }
    __syncthreads();
    if (t < m) {
        float box[7];
        rcnn_decode_box(c, reg + ((long)b * m + t) * c.channels, rois + ((long)b * m + t) * 7, box);
        raw = cls[(long)b * m + t];
        const float norm = 1.0f / (1.0f + expf(-raw));
        selected = norm > c.score_thresh;
        float *o = pred + ((long)b * m + t) * 7;
#pragma unroll
        for (int e = 0; e < 7; ++e) { o[e] = box[e]; s_box[t * 8 + e] = box[e]; }
        s_box[t * 8 + 7] = raw;
    }
    if (t < RF_MAX) {
        keys[t] = (t < m && selected) ? sort_key(raw, (unsigned)t) : (~0ull - 127 + t);
        if (t < m && selected) atomicAdd(&nsel, 1);
        s_mask[t][0] = 0ull; s_mask[t][1] = 0ull;
    }
    __syncthreads();
    for (int k = 2; k <= RF_MAX; k <<= 1)
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const int partner = t ^ j;
            if (t < RF_MAX && partner > t) {
                const bool up = ((t & k) == 0);
                const unsigned long long a = keys[t], d = keys[partner];
                if ((a > d) == up) { keys[t] = d; keys[partner] = a; }
            }
            __syncthreads();
        }
    const int n = nsel;
    if (t < n) {                                         // slot t of the score order holds RoI `src`
        const int src = (int)(keys[t] & 0xffffffffu);
        s_src[t] = src;
        const float *q = s_box + src * 8;
        const float hl = q[5] / 2, hw = q[4] / 2;
        float *v = s_row + t * 7;
        v[0] = q[0] - hl; v[1] = q[2] - hw; v[2] = q[0] + hl; v[3] = q[2] + hw; v[4] = q[6];
        v[5] = cos_f32(q[6]); v[6] = sin_f32(q[6]);
    }
    __syncthreads();
    // candidate pairs: e = r (2 n - r - 1) / 2 + (c - r - 1) enumerates (r, c > r) row by row
    const int P = n * (n - 1) / 2;
    for (int e = t; e < P; e += RF_THREADS) {
        const float tn = (float)(2 * n - 1);
        int r = (int)((tn - sqrtf(tn * tn - 8.0f * (float)e)) * 0.5f);
        r = min(max(r, 0), n - 2);
        while (r > 0 && r * (2 * n - r - 1) / 2 > e) --r;
        while ((r + 1) * (2 * n - r - 2) / 2 <= e) ++r;
        const int cc = r + 1 + (e - r * (2 * n - r - 1) / 2);
        if (!rbox_far_apart(s_row + r * 7, s_row + cc * 7)) s_pair[atomicAdd(&s_npair, 1)] = ((unsigned)r << 16) | (unsigned)cc;
    }
    __syncthreads();
    const int np = s_npair;
    for (int i = t; i < np; i += RF_THREADS) {
        const unsigned int rc = s_pair[i];
        const int r = (int)(rc >> 16), cc = (int)(rc & 0xffffu);
        RBox C;
#pragma unroll
        for (int q = 0; q < 5; ++q) C.v[q] = s_row[cc * 7 + q];
        C.cosv = s_row[cc * 7 + 5]; C.sinv = s_row[cc * 7 + 6];
        if (suppresses<true, true>(s_row, r, C, nms_thresh, s_poly + t, RF_THREADS))
            atomicOr(&s_mask[r][cc >> 6], 1ull << (cc & 63));
    }
    __syncthreads();
    if (t == 0) {                                        // the host loop of iou3d.cpp:100-119 over the row masks
        unsigned long long gone0 = 0ull, gone1 = 0ull;
        int nk = 0;
        for (int cc = 0; cc < n; ++cc) {
            const bool gone = cc < 64 ? (gone0 >> cc) & 1ull : (gone1 >> (cc - 64)) & 1ull;
            if (gone) continue;
            s_keep[nk++] = cc;
            gone0 |= s_mask[cc][0];
            gone1 |= s_mask[cc][1];
        }
        s_nkeep = nk;
        num[b] = nk;
    }
    __syncthreads();
    if (t < m) {
        float *o = boxes + ((long)b * m + t) * 7;
        if (t < s_nkeep) {
            const float *q = s_box + s_src[s_keep[t]] * 8;
#pragma unroll
            for (int e = 0; e < 7; ++e) o[e] = q[e];
            scores[(long)b * m + t] = q[7];
        } else {
#pragma unroll
            for (int e = 0; e < 7; ++e) o[e] = 0.f;
            scores[(long)b * m + t] = 0.f;
        }
    }
}

static size_t aligned(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace prcnn

using namespace prcnn;

// The per-point RCNN inputs of point_rcnn.py:44-52 / rcnn_net.py:131-137 in ONE launch (round 4; torch ran them as six: sigmoid, compare,
// cast, norm, divide, subtract): seg = sigmoid(score) > thresh as 0 / 1 floats, depth = |xyz|, depth_norm = depth / 70 - 0.5.
namespace prcnn {
__global__ __launch_bounds__(256) void point_aux_kernel(long rows, float thresh, const float *__restrict__ scores,
                                                        const float *__restrict__ xyz, float *__restrict__ seg,
                                                        float *__restrict__ depth, float *__restrict__ depth_norm)
{
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float sg = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-scores[r])));
    seg[r] = sg > thresh ? 1.0f : 0.0f;
    const float x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
    const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    depth[r] = d;
    depth_norm[r] = __fsub_rn(__fdiv_rn(d, 70.0f), 0.5f);
}
}  // namespace prcnn

extern "C" int prcnn_point_aux(long rows, float thresh, const float *scores, const float *xyz, float *seg, float *depth,
                               float *depth_norm, void *stream)
{
    PRCNN_REQUIRE(rows >= 0, "point_aux: bad sizes");
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(scores && xyz && seg && depth && depth_norm, "point_aux: null pointer");
    hipLaunchKernelGGL(point_aux_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, thresh, scores, xyz,
                       seg, depth, depth_norm);
    return check_launch("point_aux");
}

// the proposal layer behind the decode: boxes_in != NULL = every point's decoded box (b,n,7) as prcnn_rpn_tail_lin_boxes leaves it
// (round 5: the decode rides in the RPN tail kernel), else reg + xyz are decoded here first
static int rpn_proposals_any(int b, int n, const DecodeCfg *dc, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int rotated_nms,
                             const float *xyz, const float *scores, const float *reg, const float *boxes_in, float *rois,
                             float *roi_scores, void *stream)
{
    int npad = 1;
    while (npad < n) npad <<= 1;
    PRCNN_REQUIRE(npad <= 65536, "rpn_proposals: n=%d > 65536 points per scene unsupported by the fused path", n);   // (round 5: 32768 = tools/cfgs/double.yaml; the chunked sort and the band scan take any n)
    const int pre_near = (int)(pre_nms_top_n * 0.7), pre_far = pre_nms_top_n - pre_near;
    const int post_near = (int)(post_nms_top_n * 0.7), post_far = post_nms_top_n - post_near;
    PRCNN_REQUIRE(post_nms_top_n <= 128 && post_near >= post_far && pre_near >= pre_far, "rpn_proposals: unsupported quotas");
    if (b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(scores && rois && roi_scores && (boxes_in || (xyz && reg)), "rpn_proposals: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int rows = pre_near;
    const size_t o_boxes = 0;
    const size_t o_order = o_boxes + (boxes_in ? 0 : aligned((size_t)b * n * 7 * 4));
    const size_t o_pay = o_order + aligned((size_t)b * n * 4);
    const size_t o_bev = o_pay + aligned((size_t)b * 2 * rows * 8 * 4);
    const size_t o_cnt = o_bev + aligned((size_t)b * 2 * rows * 5 * 4);
    const size_t o_keep = o_cnt + aligned((size_t)b * 2 * 4);
    const size_t o_num = o_keep + aligned((size_t)b * 2 * post_near * 4);
    const size_t need = o_num + aligned((size_t)b * 2 * 4);
    char *base = scratch_for(st, need, 2);
    if (!base) { set_error("rpn_proposals: cannot allocate %zu bytes of scratch", need); return PRCNN_ELAUNCH; }
    const float *boxes = boxes_in ? boxes_in : (const float *)(base + o_boxes);
    int *order = (int *)(base + o_order);
    float *payload = (float *)(base + o_pay), *bev = (float *)(base + o_bev);
    int *counts = (int *)(base + o_cnt), *keep = (int *)(base + o_keep), *num = (int *)(base + o_num);

    const long total = (long)b * n;
    if (!boxes_in)
        hipLaunchKernelGGL(rpn_decode_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, total, *dc, xyz, reg, (float *)(base + o_boxes));
    if (npad > SS_CHUNK) {        // (small problems: one workgroup sorts a scene with all keys in LDS; round 6: switch PRCNN_SORT_SPLIT removed)
        // chunks of 4096 keys sorted by a workgroup each, then pairwise merge-path rounds (ping-pong between two key buffers)
        unsigned long long *kbuf = (unsigned long long *)scratch_for(st, (size_t)2 * b * npad * sizeof(unsigned long long), 11);
        if (!kbuf) { set_error("rpn_proposals: cannot allocate the sort scratch"); return PRCNN_ELAUNCH; }
        unsigned long long *ka = kbuf, *kb = kbuf + (size_t)b * npad;
        hipLaunchKernelGGL(score_sort_chunk_kernel, dim3(npad / SS_CHUNK, b), dim3(1024), 0, st, n, npad, scores, ka);
        for (int run = SS_CHUNK; run < npad; run <<= 1) {
            const bool final_round = 2 * run >= npad;
            hipLaunchKernelGGL(score_merge_kernel, dim3(ceil_div(npad, 256 * SS_MERGE_V), b), dim3(256), 0, st, n, npad, run, ka,
                               kb, final_round ? order : nullptr);
            unsigned long long *tmp = ka; ka = kb; kb = tmp;
        }
    } else {
        const size_t lds = (size_t)npad * sizeof(unsigned long long);
        if (lds > 64 * 1024) {
            const int rc = ensure_dynamic_lds((const void *)score_sort_kernel, lds, "rpn_proposals");
            if (rc != PRCNN_OK) return rc;
        }
        hipLaunchKernelGGL(score_sort_kernel, dim3(b), dim3(1024), lds, st, n, npad, scores, order);
    }
    // (round 4 tried the selection in two steps -- one scan over 16 consecutive positions per thread + table emission from LDS: 12 us instead of
    //  61 alone, 6460 / 6484 scenes/s against 6518 / 6522 at K = 96: the proposal stream's latency kernels do not bound a step.  Not kept.)
    hipLaunchKernelGGL(band_select_kernel, dim3(b), dim3(1024), 0, st, n, rows, pre_near, pre_far, boxes, scores, order,
                       payload, bev, counts);
    int rc = check_launch("rpn_proposals");
    if (rc != PRCNN_OK) return rc;
    rc = nms_device(2 * b, rows, counts, bev, nms_thresh, rotated_nms, post_near, keep, num, st);
    if (rc != PRCNN_OK) return rc;
    hipLaunchKernelGGL(assemble_rois_kernel, dim3(b), dim3(128), 0, st, rows, post_near, post_near, post_far, payload, keep,
                       num, rois, roi_scores);
    return check_launch("rpn_proposals");
}

// xyz (b,n,3), scores (b,n) raw RPN scores, reg (b,n,channels) -> rois (b, post_top_n, 7), roi_scores (b, post_top_n)
// distance-based proposal (RPN_DISTANCE_BASED_PROPOSE), get_y_by_bin = False, get_ry_fine = False.
extern "C" int prcnn_rpn_proposals(int b, int n, int channels, float loc_scope, float loc_bin_size,
                                   int num_head_bin, int xz_fine, const float *anchor_size_host,
                                   int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int rotated_nms,
                                   const float *xyz, const float *scores, const float *reg, float *rois,
                                   float *roi_scores, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n > 0 && channels > 0 && num_head_bin > 0 && loc_bin_size > 0, "rpn_proposals: bad sizes");
    PRCNN_REQUIRE(anchor_size_host, "rpn_proposals: anchor size missing");
    DecodeCfg c;
    c.loc_scope = loc_scope; c.loc_bin_size = loc_bin_size;
    c.nbin = (int)(loc_scope / loc_bin_size) * 2;
    c.num_head_bin = num_head_bin; c.xz_fine = xz_fine ? 1 : 0; c.channels = channels;
    for (int i = 0; i < 3; ++i) c.anchor[i] = anchor_size_host[i];
    const int expect = c.nbin * (c.xz_fine ? 4 : 2) + 1 + 2 * num_head_bin + 3;
    PRCNN_REQUIRE(channels == expect, "rpn_proposals: %d regression channels, layout needs %d", channels, expect);
    PRCNN_REQUIRE(b == 0 || (xyz && reg), "rpn_proposals: null pointer");
    return rpn_proposals_any(b, n, &c, pre_nms_top_n, post_nms_top_n, nms_thresh, rotated_nms, xyz, scores, reg, nullptr, rois, roi_scores,
                             stream);
}

// the same layer over boxes decoded already (prcnn_rpn_tail_lin_boxes): boxes (b,n,7), scores (b,n)
extern "C" int prcnn_rpn_proposals_boxes(int b, int n, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int rotated_nms,
                                         const float *scores, const float *boxes, float *rois, float *roi_scores, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n > 0, "rpn_proposals_boxes: bad sizes");
    PRCNN_REQUIRE(b == 0 || boxes, "rpn_proposals_boxes: null pointer");
    return rpn_proposals_any(b, n, nullptr, pre_nms_top_n, post_nms_top_n, nms_thresh, rotated_nms, nullptr, scores, nullptr, boxes, rois,
                             roi_scores, stream);
}


// rois (b,m,7), rcnn_reg (b,m,channels), rcnn_cls (b,m) raw -> pred_boxes3d (b,m,7) decoded in RoI order,
// boxes (b,m,7) / scores (b,m) = survivors of score threshold + rotated NMS in descending score order, zero
// padded, num (b) i32.  get_xz_fine = get_ry_fine = True (eval_rcnn.py:516-523).  m <= 128.
static int rcnn_postprocess_any(int b, int m, int channels, float loc_scope, float loc_bin_size,
                                int num_head_bin, int y_by_bin, float loc_y_scope, float loc_y_bin_size,
                                const float *anchor_size_host, float score_thresh, float nms_thresh,
                                const float *rois, const float *rcnn_reg, const float *rcnn_cls,
                                float *pred_boxes3d, float *boxes, float *scores, int *num, int scenes_per_blob, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && m > 0 && m <= 128 && channels > 0 && num_head_bin > 0, "rcnn_postprocess: bad sizes (m <= 128)");
    PRCNN_REQUIRE(anchor_size_host, "rcnn_postprocess: anchor size missing");
    RcnnCfg c;
    c.loc_scope = loc_scope; c.loc_bin_size = loc_bin_size; c.loc_y_scope = loc_y_scope; c.loc_y_bin_size = loc_y_bin_size;
    c.score_thresh = score_thresh;
    c.nbin = (int)(loc_scope / loc_bin_size) * 2;
    c.nbin_y = (int)(loc_y_scope / loc_y_bin_size) * 2;
    c.num_head_bin = num_head_bin; c.y_by_bin = y_by_bin ? 1 : 0; c.channels = channels;
    for (int i = 0; i < 3; ++i) c.anchor[i] = anchor_size_host[i];
    const int expect = c.nbin * 4 + (c.y_by_bin ? 2 * c.nbin_y : 1) + 2 * num_head_bin + 3;
    PRCNN_REQUIRE(channels == expect, "rcnn_postprocess: %d regression channels, layout needs %d", channels, expect);
    if (b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rois && rcnn_reg && rcnn_cls && pred_boxes3d && boxes && scores && num, "rcnn_postprocess: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool fused = true;                               // (round 6: switch PRCNN_FINAL_FUSED removed; the four-launch form serves nms_thresh < 0)
    PRCNN_REQUIRE(scenes_per_blob == 0 || nms_thresh >= 0.f, "rcnn_postprocess_blobs: needs the one-workgroup final stage (nms_thresh >= 0)");
    if (fused && nms_thresh >= 0.f) {
        hipLaunchKernelGGL(rcnn_final_kernel, dim3(b), dim3(RF_THREADS), 0, st, m, c, nms_thresh, rois, rcnn_reg, rcnn_cls, pred_boxes3d,
                           boxes, scores, num, scenes_per_blob);
        return check_launch("rcnn_postprocess");
    }
    const size_t o_sorted = 0;
    const size_t o_bev = o_sorted + aligned((size_t)b * m * 8 * 4);
    const size_t o_cnt = o_bev + aligned((size_t)b * m * 5 * 4);
    const size_t o_keep = o_cnt + aligned((size_t)b * 4);
    const size_t need = o_keep + aligned((size_t)b * m * 4);
    char *base = scratch_for(st, need, 3);
    if (!base) { set_error("rcnn_postprocess: cannot allocate %zu bytes of scratch", need); return PRCNN_ELAUNCH; }
    float *sorted = (float *)(base + o_sorted), *bev = (float *)(base + o_bev);
    int *counts = (int *)(base + o_cnt), *keep = (int *)(base + o_keep);
    hipLaunchKernelGGL(rcnn_decode_select_kernel, dim3(b), dim3(128), 0, st, m, c, rois, rcnn_reg, rcnn_cls, pred_boxes3d,
                       sorted, bev, counts);
    int rc = check_launch("rcnn_postprocess");
    if (rc != PRCNN_OK) return rc;
    rc = nms_device(b, m, counts, bev, nms_thresh, 1, m, keep, num, st);
    if (rc != PRCNN_OK) return rc;
    hipLaunchKernelGGL(rcnn_final_gather_kernel, dim3(b), dim3(128), 0, st, m, sorted, keep, num, boxes, scores);
    return check_launch("rcnn_postprocess");
}

extern "C" int prcnn_rcnn_postprocess(int b, int m, int channels, float loc_scope, float loc_bin_size,
                                      int num_head_bin, int y_by_bin, float loc_y_scope, float loc_y_bin_size,
                                      const float *anchor_size_host, float score_thresh, float nms_thresh,
                                      const float *rois, const float *rcnn_reg, const float *rcnn_cls,
                                      float *pred_boxes3d, float *boxes, float *scores, int *num, void *stream)
{
    return rcnn_postprocess_any(b, m, channels, loc_scope, loc_bin_size, num_head_bin, y_by_bin, loc_y_scope, loc_y_bin_size,
                                anchor_size_host, score_thresh, nms_thresh, rois, rcnn_reg, rcnn_cls, pred_boxes3d, boxes, scores, num, 0,
                                stream);
}

// The same with the results as one BLOB per batch of scenes_per_blob scenes (b % scenes_per_blob == 0): blobs (b / spb, spb (8 m + 1))
// f32, each [spb m 7 boxes | spb m scores | spb num (i32 bits)] -- eval_rcnn.split_detections' layout per batch, so that a launch over a
// PAIR of batches still hands every batch's detections to the host with one copy.  Needs the one-workgroup final kernel (nms_thresh >= 0).
extern "C" int prcnn_rcnn_postprocess_blobs(int b, int m, int channels, float loc_scope, float loc_bin_size,
                                            int num_head_bin, int y_by_bin, float loc_y_scope, float loc_y_bin_size,
                                            const float *anchor_size_host, float score_thresh, float nms_thresh,
                                            const float *rois, const float *rcnn_reg, const float *rcnn_cls,
                                            float *pred_boxes3d, float *blobs, int scenes_per_blob, void *stream)
{
    PRCNN_REQUIRE(scenes_per_blob > 0 && b % scenes_per_blob == 0 && nms_thresh >= 0.f, "rcnn_postprocess_blobs: %d scenes in blobs of %d",
                  b, scenes_per_blob);
    PRCNN_REQUIRE(blobs || b == 0, "rcnn_postprocess_blobs: null pointer");
    return rcnn_postprocess_any(b, m, channels, loc_scope, loc_bin_size, num_head_bin, y_by_bin, loc_y_scope, loc_y_bin_size,
                                anchor_size_host, score_thresh, nms_thresh, rois, rcnn_reg, rcnn_cls, pred_boxes3d, blobs, blobs,
                                reinterpret_cast<int *>(blobs), scenes_per_blob, stream);
}
