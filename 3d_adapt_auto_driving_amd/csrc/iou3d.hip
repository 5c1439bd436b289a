// iou3d.hip -- BEV rotated overlap / IoU (K10, K11), rotated and axis-aligned greedy NMS
// (K12, K13) for gfx950.
//
// Reference behaviour restated: lib/utils/iou3d/src/iou3d_kernel.cu:14-348 and the host greedy
// reduce of iou3d.cpp:73-170.
//
// NMS design.  The reference builds the full n x n/64 suppression mask on the device, copies it
// to the host and reduces it serially there (3 round trips per scene in eval_rcnn).  Here one
// workgroup per problem walks the score-sorted boxes in blocks of 64 rows and evaluates IoUs
// LAZILY: for row block r it computes, for every still-alive later column, the 64-bit word
// "which rows of block r suppress me" (rows staged in LDS, one column per lane), resolves the
// 64 rows of the block serially against the already-known words (same order as the host loop),
// then kills the columns suppressed by the rows that were kept.  It stops as soon as
// `max_keep` boxes are kept -- the proposal layer only ever uses the first 70/30 -- and the
// kept list is, by construction, the prefix of the reference's list.  No host round trip, no
// mask in HBM.
#include "common.hpp"
#include "rbox_iou.hpp"
#include <math.h>
#include <stdlib.h>

namespace prcnn {

template <bool IOU>
__global__ __launch_bounds__(256) void pair_kernel(int na, const float *__restrict__ a, int nb,
                                                   const float *__restrict__ b, float *__restrict__ out)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)na * nb) return;
    const int i = (int)(e / nb), j = (int)(e - (long)i * nb);
    const RBox A = make_rbox(a + 5 * i), B = make_rbox(b + 5 * j);
    out[e] = IOU ? rbox_iou(A, B) : rbox_overlap(A, B);
}

// ---- lazy greedy NMS, one workgroup per problem ----------------------------------------
constexpr int NMS_THREADS = 512;
constexpr int NMS_MAX_N = 65536;   // removed-bitmask lives in LDS (8 KiB)
constexpr int NMS_RCH = 8;         // rows per work item

// ---- quota form: max_keep <= NMS_QUOTA_MAX (the proposal layer keeps 70 + 30 of up to 6300 + 2700 boxes per scene) ------------
// The general kernel below lets the kept rows of a block knock out EVERY later column before it moves on: 6236 columns x 64
// rows of IoUs after the first block of a 6300-box problem whose quota is reached inside the second block (0.22 ms, the
// longest kernel of the proposal stream).  Here the kept boxes (at most max_keep of them) stay in LDS and a block's 64 columns
// are tested against them when the block is staged: 64 x kept IoUs per block, nothing for columns that are never reached.
// Same decisions: a column is dropped iff an earlier KEPT row overlaps it (row, column argument order as nms_kernel :285).
constexpr int NMS_QUOTA_MAX = 256;

__global__ __launch_bounds__(NMS_THREADS) void nms_quota_kernel(
    int n_max, const int *__restrict__ counts, const float *__restrict__ boxes_all, float thresh,
    int max_keep, int *__restrict__ keep_all, int *__restrict__ num_keep_all)
{
    __shared__ float s_row[64 * 7];             // the 64 row boxes of the block
    __shared__ float s_kbox[NMS_QUOTA_MAX * 7]; // the kept boxes so far (same 7-float slots as s_row)
    __shared__ unsigned long long s_diag[64];   // word(c): rows of the block with IoU > thresh, r < c
    __shared__ unsigned long long s_gone;       // columns of the block dropped by earlier kept rows
    __shared__ int s_nkeep;

    const int prob = blockIdx.x;
    int n = counts ? counts[prob] : n_max;
    n = min(max(n, 0), n_max);
    const float *__restrict__ boxes = boxes_all + (long)prob * n_max * 5;
    int *__restrict__ keep = keep_all + (long)prob * max_keep;
    const int t = threadIdx.x;
    for (int i = t; i < max_keep; i += NMS_THREADS) keep[i] = -1;
    if (t == 0) s_nkeep = 0;
    __syncthreads();

    const int nblocks = (n + 63) / 64;
    for (int rb = 0; rb < nblocks; ++rb) {
        const int r0 = rb * 64;
        const int rows = min(64, n - r0);
        const int nk0 = s_nkeep;                 // kept before this block
        if (t < rows) {
            const float *p = boxes + (long)(r0 + t) * 5;
#pragma unroll
            for (int q = 0; q < 5; ++q) s_row[t * 7 + q] = p[q];
        }
        if (t < 64) s_diag[t] = 0ull;
        if (t == 0) s_gone = 0ull;
        __syncthreads();
        // A: the block's columns against the kept boxes of earlier blocks; B: against the earlier rows of the block itself.
        //    item = (column, chunk): 512 threads = 64 columns x 8 chunks
        {
            const int cl = t & 63, ch = t >> 6;
            if (cl < rows) {
                RBox C;
#pragma unroll
                for (int q = 0; q < 5; ++q) C.v[q] = s_row[cl * 7 + q];
                C.cosv = 1.f; C.sinv = 0.f;
                bool gone = false;
                for (int k = ch; k < nk0 && !gone; k += 8) gone = aabox_iou(&s_kbox[k * 7], C.v) > thresh;
                if (gone) atomicOr(&s_gone, 1ull << cl);
                const int rlo = ch * NMS_RCH, rhi = min(rlo + NMS_RCH, cl);
                unsigned long long w = 0;
                for (int r = rlo; r < rhi; ++r)
                    if (aabox_iou(&s_row[r * 7], C.v) > thresh) w |= 1ull << r;
                if (w) atomicOr(&s_diag[cl], w);
            }
        }
        __syncthreads();
        // C: serial resolve of the 64 rows, exactly the host loop of iou3d.cpp:100-119
        if (t == 0) {
            unsigned long long kept = 0;
            const unsigned long long gone = s_gone;
            int nk = nk0;
            for (int cl = 0; cl < rows && nk < max_keep; ++cl) {
                if ((gone >> cl) & 1ull) continue;
                if (s_diag[cl] & kept) continue;
                kept |= 1ull << cl;
#pragma unroll
                for (int q = 0; q < 5; ++q) s_kbox[nk * 7 + q] = s_row[cl * 7 + q];
                keep[nk++] = r0 + cl;
            }
            s_nkeep = nk;
        }
        __syncthreads();
        if (s_nkeep >= max_keep) break;
    }
    if (t == 0) num_keep_all[prob] = s_nkeep;
}

template <bool ROTATED>
__global__ __launch_bounds__(NMS_THREADS) void nms_lazy_kernel(
    int n_max, const int *__restrict__ counts, const float *__restrict__ boxes_all, float thresh,
    int max_keep, int *__restrict__ keep_all, int *__restrict__ num_keep_all)
{
    __shared__ float s_row[64 * 7];             // the 64 row boxes of the block (+cos,sin)
    __shared__ unsigned long long s_diag[64];   // word(c): rows of the block with IoU > thresh, r < c
    __shared__ unsigned long long s_kept;       // rows of the block that survive
    __shared__ unsigned int s_removed[NMS_MAX_N / 32];
    __shared__ int s_nkeep;

    const int prob = blockIdx.x;
    int n = counts ? counts[prob] : n_max;
    n = min(max(n, 0), n_max);
    const float *__restrict__ boxes = boxes_all + (long)prob * n_max * 5;
    int *__restrict__ keep = keep_all + (long)prob * max_keep;
    const int t = threadIdx.x;

    for (int i = t; i < (n + 31) / 32; i += NMS_THREADS) s_removed[i] = 0u;
    for (int i = t; i < max_keep; i += NMS_THREADS) keep[i] = -1;
    if (t == 0) s_nkeep = 0;
    __syncthreads();

    const int nblocks = (n + 63) / 64;
    for (int rb = 0; rb < nblocks; ++rb) {
        const int r0 = rb * 64;
        const int rows = min(64, n - r0);
        // A: stage the row boxes
        if (t < rows) {
            const float *p = boxes + (long)(r0 + t) * 5;
#pragma unroll
            for (int q = 0; q < 5; ++q) s_row[t * 7 + q] = p[q];
            if (ROTATED) {
                s_row[t * 7 + 5] = cos_f32(p[4]);
                s_row[t * 7 + 6] = sin_f32(p[4]);
            }
        }
        if (t < 64) s_diag[t] = 0ull;
        __syncthreads();

        // B: the block's own 64 columns; item = (column, chunk of NMS_RCH rows)
        {
            const int cl = t & 63, ch = t >> 6;  // 512 threads = 64 columns x 8 chunks
            if (cl < rows && !((s_removed[(r0 + cl) >> 5] >> ((r0 + cl) & 31)) & 1u)) {
                const int rlo = ch * NMS_RCH, rhi = min(rlo + NMS_RCH, cl);  // rows before the column
                if (rlo < rhi) {
                    const RBox C = load_col<ROTATED>(boxes + (long)(r0 + cl) * 5);
                    unsigned long long w = 0;
                    for (int r = rlo; r < rhi; ++r)
                        if (suppresses<ROTATED>(s_row, r, C, thresh)) w |= 1ull << r;
                    if (w) atomicOr(&s_diag[cl], w);
                }
            }
        }
        __syncthreads();

        // C: serial resolve of the 64 rows, exactly the host loop of iou3d.cpp:100-119
        if (t == 0) {
            unsigned long long kept = 0;
            int nk = s_nkeep;
            for (int cl = 0; cl < rows && nk < max_keep; ++cl) {
                const int c = r0 + cl;
                if ((s_removed[c >> 5] >> (c & 31)) & 1u) continue;
                if (s_diag[cl] & kept) continue;
                kept |= 1ull << cl;
                keep[nk++] = c;
            }
            s_kept = kept;
            s_nkeep = nk;
        }
        __syncthreads();
        if (s_nkeep >= max_keep) break;

        // D: kept rows knock out later columns
        const unsigned long long kept = s_kept;
        const int c_first = r0 + 64;
        const long items = (long)max(0, n - c_first) * (64 / NMS_RCH);
        for (long e = t; e < items; e += NMS_THREADS) {
            const int c = c_first + (int)(e / (64 / NMS_RCH));
            const int ch = (int)(e % (64 / NMS_RCH));
            const unsigned int chunk_bits = (unsigned int)((kept >> (ch * NMS_RCH)) & ((1u << NMS_RCH) - 1u));
            if (!chunk_bits) continue;
            if ((s_removed[c >> 5] >> (c & 31)) & 1u) continue;  // monotone flag: a stale 0 only costs work
            const RBox C = load_col<ROTATED>(boxes + (long)c * 5);
            for (int q = 0; q < NMS_RCH; ++q) {
                if (!((chunk_bits >> q) & 1u)) continue;
                if (suppresses<ROTATED>(s_row, ch * NMS_RCH + q, C, thresh)) {
                    atomicOr(&s_removed[c >> 5], 1u << (c & 31));
                    break;
                }
            }
        }
        __syncthreads();
    }
    if (t == 0) num_keep_all[prob] = s_nkeep;
}


// ---- dense form for small problems (n <= 128: the FINAL rotated NMS of a scene runs over <= 100 boxes, eval_rcnn.py:626) --------
// The lazy kernel above is one workgroup per problem walking 64-row blocks: for 100 boxes its critical path is ~24 rotated IoUs
// evaluated one after the other by the same thread (175 us, the longest kernel of the final stage).  Here the reference's own
// split is used (iou3d_kernel.cu:250-292 + iou3d.cpp:100-119): every pair (row r, column c > r) is one IoU evaluated by its own
// thread -- 8 rows x 32 column lanes per workgroup, ~1.5 IoUs per thread -- into a 128-bit suppression mask per row, and one wave
// per problem then runs the host loop over the masks.  Same argument order (row, column), same expression: same keep list.
constexpr int ND_ROWS = 8, ND_LANES = 32, ND_MAX = 128;

template <bool ROTATED>
__global__ __launch_bounds__(ND_ROWS * ND_LANES) void nms_dense_mask_kernel(
    int n_max, const int *__restrict__ counts, const float *__restrict__ boxes_all, float thresh,
    unsigned long long *__restrict__ mask_all)
{
    __shared__ float s_row[ND_ROWS * 7];
    __shared__ unsigned long long s_mask[ND_ROWS][2];
    __shared__ float s_poly[ROTATED ? POLY_LDS_FLOATS * ND_ROWS * ND_LANES : 1];      // the clipped polygons (72 KB): see PolyLds
    const int prob = blockIdx.y, r0 = blockIdx.x * ND_ROWS;
    int n = counts ? counts[prob] : n_max;
    n = min(max(n, 0), n_max);
    if (r0 >= n) return;
    const float *__restrict__ boxes = boxes_all + (long)prob * n_max * 5;
    const int t = threadIdx.x, rl = t / ND_LANES, cl = t % ND_LANES;
    const int rows = min(ND_ROWS, n - r0);
    if (t < rows) {
        const float *p = boxes + (long)(r0 + t) * 5;
#pragma unroll
        for (int q = 0; q < 5; ++q) s_row[t * 7 + q] = p[q];
        if (ROTATED) {
            s_row[t * 7 + 5] = cos_f32(p[4]);
            s_row[t * 7 + 6] = sin_f32(p[4]);
        }
    }
    if (t < ND_ROWS * 2) s_mask[t >> 1][t & 1] = 0ull;
    __syncthreads();
    if (rl < rows) {
        unsigned long long w0 = 0ull, w1 = 0ull;
        for (int c = r0 + rl + 1 + cl; c < n; c += ND_LANES) {
            const RBox C = load_col<ROTATED>(boxes + (long)c * 5);
            if (suppresses<ROTATED, ROTATED>(s_row, rl, C, thresh, s_poly + t, ND_ROWS * ND_LANES)) {
                if (c < 64) w0 |= 1ull << c; else w1 |= 1ull << (c - 64);
            }
        }
        if (w0) atomicOr(&s_mask[rl][0], w0);
        if (w1) atomicOr(&s_mask[rl][1], w1);
    }
    __syncthreads();
    if (t < rows * 2) mask_all[((long)prob * ND_MAX + r0 + (t >> 1)) * 2 + (t & 1)] = s_mask[t >> 1][t & 1];
}

// one wave per problem: the host loop of iou3d.cpp:100-119 over the row masks
__global__ __launch_bounds__(64) void nms_dense_resolve_kernel(
    int n_max, const int *__restrict__ counts, const unsigned long long *__restrict__ mask_all, int max_keep,
    int *__restrict__ keep_all, int *__restrict__ num_keep_all)
{
    __shared__ unsigned long long s_m[ND_MAX][2];
    const int prob = blockIdx.x, t = threadIdx.x;
    int n = counts ? counts[prob] : n_max;
    n = min(max(n, 0), n_max);
    int *__restrict__ keep = keep_all + (long)prob * max_keep;
    for (int i = t; i < 2 * n; i += 64) s_m[i >> 1][i & 1] = mask_all[((long)prob * ND_MAX) * 2 + i];
    for (int i = t; i < max_keep; i += 64) keep[i] = -1;
    __syncthreads();
    if (t == 0) {
        unsigned long long gone0 = 0ull, gone1 = 0ull;
        int nk = 0;
        for (int c = 0; c < n && nk < max_keep; ++c) {
            const bool gone = c < 64 ? (gone0 >> c) & 1ull : (gone1 >> (c - 64)) & 1ull;
            if (gone) continue;
            keep[nk++] = c;
            gone0 |= s_m[c][0];
            gone1 |= s_m[c][1];
        }
        num_keep_all[prob] = nk;
    }
}

// device scratch of the blocking API: slot 7 of the per-(device, stream) cache

int nms_device(int nprob, int n_max, const int *counts, const float *boxes, float thresh,
               int rotated, int max_keep, int *keep, int *num_keep, hipStream_t st)
{
    PRCNN_REQUIRE(nprob >= 0 && n_max >= 0 && max_keep >= 0, "nms: bad sizes");
    PRCNN_REQUIRE(n_max <= NMS_MAX_N, "nms: %d boxes > %d unsupported", n_max, NMS_MAX_N);
    if (nprob == 0) return PRCNN_OK;
    PRCNN_REQUIRE(num_keep && (keep || max_keep == 0) && (boxes || n_max == 0), "nms: null pointer");
    const bool quota_form = true, dense_form = true;      // (round 6: the switches PRCNN_NMS_QUOTA / _DENSE / _FULL are gone; the forms are chosen by shape)
    if (dense_form && n_max >= 1 && n_max <= ND_MAX) {
        // small problems (the final stage: <= 100 boxes per scene): all pairs at once + a one-wave resolve
        unsigned long long *mask = (unsigned long long *)scratch_for(st, (size_t)nprob * ND_MAX * 2 * sizeof(unsigned long long), 10);
        if (!mask) { set_error("nms: cannot allocate the mask scratch"); return PRCNN_ELAUNCH; }
        const dim3 grid(ceil_div(n_max, ND_ROWS), nprob);
        if (rotated)
            hipLaunchKernelGGL(nms_dense_mask_kernel<true>, grid, dim3(ND_ROWS * ND_LANES), 0, st, n_max, counts, boxes, thresh, mask);
        else
            hipLaunchKernelGGL(nms_dense_mask_kernel<false>, grid, dim3(ND_ROWS * ND_LANES), 0, st, n_max, counts, boxes, thresh, mask);
        hipLaunchKernelGGL(nms_dense_resolve_kernel, dim3(nprob), dim3(64), 0, st, n_max, counts, mask, max_keep, keep, num_keep);
        return check_launch("nms(dense)");
    }
    if (rotated)
        hipLaunchKernelGGL(nms_lazy_kernel<true>, dim3(nprob), dim3(NMS_THREADS), 0, st, n_max, counts, boxes, thresh, max_keep, keep, num_keep);
    else if (quota_form && max_keep >= 1 && max_keep <= NMS_QUOTA_MAX)
        hipLaunchKernelGGL(nms_quota_kernel, dim3(nprob), dim3(NMS_THREADS), 0, st, n_max, counts, boxes, thresh, max_keep, keep, num_keep);
    else
        hipLaunchKernelGGL(nms_lazy_kernel<false>, dim3(nprob), dim3(NMS_THREADS), 0, st, n_max, counts, boxes, thresh, max_keep, keep, num_keep);
    return check_launch("nms");
}

// ---- full-mask form for the blocking API (one problem, every kept box wanted: iou3d_cuda.nms_gpu / nms_normal_gpu) --------------
// The lazy kernels above are one workgroup per problem and stop at max_keep: right for 8 scenes x 100 proposals, wrong for the
// reference's own call shape (one scene, up to 9000 boxes, the whole keep list: 8-12 ms).  Here every (row block, column block
// >= row block) pair is its own workgroup -- lane = column, and one __ballot per row IS the 64-bit mask word of that row for the
// column block (wave64: the word the reference builds bit by bit, iou3d_kernel.cu:277-290) -- and one workgroup then walks the
// blocks in order: the 64 rows of a block are resolved on a wave from the diagonal words (iou3d.cpp:100-119's loop), the rows
// it kept OR their words into the removed bitmap, all threads in parallel.  Same (row, column) argument order: same keep list.
constexpr int NF_WAVES = 4;

template <bool ROTATED>
__global__ __launch_bounds__(64 * NF_WAVES) void nms_full_mask_kernel(
    int n, int W, const float *__restrict__ boxes, float thresh, unsigned long long *__restrict__ mask)
{
    __shared__ float s_row[64 * 7];
    // blockIdx.x enumerates the pairs (rb <= cb) of the upper triangle, row block major
    int rb = 0, rest = blockIdx.x;
    while (rest >= W - rb) { rest -= W - rb; ++rb; }
    const int cb = rb + rest;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int r0 = rb * 64, rows = min(64, n - r0);
    if (t < rows) {
        const float *p = boxes + (long)(r0 + t) * 5;
#pragma unroll
        for (int q = 0; q < 5; ++q) s_row[t * 7 + q] = p[q];
        if (ROTATED) {
            s_row[t * 7 + 5] = cos_f32(p[4]);
            s_row[t * 7 + 6] = sin_f32(p[4]);
        }
    }
    __syncthreads();
    const int c = cb * 64 + lane;
    const bool live = c < n;
    const RBox C = load_col<ROTATED>(boxes + (long)(live ? c : 0) * 5);
    for (int r = wv; r < rows; r += NF_WAVES) {
        const bool hit = live && c > r0 + r && suppresses<ROTATED>(s_row, r, C, thresh);
        const unsigned long long w = __ballot(hit);
        if (lane == 0) mask[(long)(r0 + r) * W + cb] = w;
    }
}

constexpr int NF_RES_THREADS = 1024;
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)     // l wave-uniform
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
// OR over the 64 lanes of a wave (wave-uniform result): four DPP steps inside the rows of 16 lanes, then the four rows.  (s_nop 1: a DPP
// operand written by the previous VALU instruction needs two wait states; the hazard recogniser does not look into inline assembly.)
#define NF_DPP_OR(CTRL_TEXT) asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 " CTRL_TEXT " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v))
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int v)
{
    unsigned int r;
    NF_DPP_OR("quad_perm:[1,0,3,2]"); v = r;
    NF_DPP_OR("quad_perm:[2,3,0,1]"); v = r;
    NF_DPP_OR("row_half_mirror"); v = r;
    NF_DPP_OR("row_mirror"); v = r;
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 0) | (unsigned int)__builtin_amdgcn_readlane((int)v, 16) |
           (unsigned int)__builtin_amdgcn_readlane((int)v, 32) | (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v)
{
    return ((unsigned long long)wave_or_u32((unsigned int)(v >> 32)) << 32) | wave_or_u32((unsigned int)v);
}
// The host loop of iou3d.cpp:100-119 over the mask words, one workgroup.  Block b = rows 64 b .. 64 b + 63:
//   * wave 0 resolves the block's diagonal word (row by row, jumping from kept row to kept row: a removed row costs nothing) and, from the
//     words it fetched for ALL 64 rows one block ahead, ORs the kept rows' words b + 1 .. b + 3 into three carries -- the removed bits of
//     the next three blocks are complete without a global round trip on the critical path;
//   * waves 1-15 OR the kept rows' words b + 4 .. W - 1 into the bitmap in LDS TWO blocks behind wave 0: the loads of block b go out
//     behind barrier b and are consumed behind barrier b + 2 (the barrier orders LDS only; the raw words wait in registers), so
//     a block costs wave 0's walk, not a global round trip.
//   * the words BEHIND that window (w >= b + NF_NEAR_END: there are none up to 8448 boxes) are not needed before block b + NF_NEAR_END; every
//     64 blocks the whole workgroup ORs them in for the kept rows of the 64 blocks just resolved (a wave per block, lanes over the words:
//     coalesced), from the kept words wave 0 left in LDS.  (Round 5: until then they were dropped -- keep lists too long beyond 8448 boxes.)
// (Round 4, second session.  Before: per block a diagonal load, a 64-step walk, a barrier, every thread ORing its words, a barrier -- two
// global round trips per block on the critical path: 394 us for 6300 proposals, 6.3 of the 63 ms the reference-order graph takes per batch.)
constexpr int NF_FAST = 3;           // words of a block's rows that wave 0 handles itself (b + 1 .. b + NF_FAST)
constexpr int NF_PASSES = 2;         // passes of 64 words that waves 1-15 keep in flight per block
constexpr int NF_NEAR_END = 1 + NF_FAST + 64 * NF_PASSES;   // block b's words b + 1 .. b + NF_NEAR_END - 1 are handled inside the walk
constexpr int NF_FAR_PERIOD = 64;    // the far words are ORed in every NF_FAR_PERIOD blocks (must be < NF_NEAR_END)
static_assert(NF_FAR_PERIOD < NF_NEAR_END, "a far pass must run before its first word is read");
__global__ __launch_bounds__(NF_RES_THREADS) void nms_full_resolve_kernel(
    int n, int W, const unsigned long long *__restrict__ mask, int *__restrict__ keep, int *__restrict__ num_keep)
{
    __shared__ unsigned long long s_removed[NMS_MAX_N / 64];
    __shared__ unsigned long long s_kept[2];
    __shared__ unsigned long long s_hist[NF_FAR_PERIOD];         // kept words of the last NF_FAR_PERIOD blocks (far pass)
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);       // wave-uniform: the walk below runs on the scalar unit
    for (int i = t; i < W; i += NF_RES_THREADS) s_removed[i] = 0ull;
    // wave 0: this block's diagonal word and the next NF_FAST words of its rows; the carries of the next NF_FAST blocks
    unsigned long long diag = 0ull, nextw[NF_FAST] = {0ull, 0ull, 0ull}, carry[NF_FAST] = {0ull, 0ull, 0ull};
    if (wv == 0 && lane < n) {
        diag = mask[(long)lane * W];
#pragma unroll
        for (int f = 0; f < NF_FAST; ++f)
            if (1 + f < W) nextw[f] = mask[(long)lane * W + 1 + f];
    }
    // waves 1-15: the raw words of the block before (newer) and of the block before that (older): 5 rows x 2 passes of 64 words
    unsigned long long rawN[5][NF_PASSES], rawO[5][NF_PASSES];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int k = 0; k < NF_PASSES; ++k) rawN[q][k] = rawO[q][k] = 0ull;
    __syncthreads();
    const bool far_words = W > NF_NEAR_END;            // n > 8448
    int nk = 0;                                        // tracked by every thread (uniform)
    for (int b = 0; b < W; ++b) {
        const int r0 = b * 64, rows = min(64, n - r0);
        if (far_words && b > 0 && b % NF_FAR_PERIOD == 0) {
            // blocks b - 64 .. b - 1 are resolved: their kept rows' words w >= block + NF_NEAR_END go into the bitmap now (the earliest of
            // them is read at block b - 64 + NF_NEAR_END > b)
            for (int q = wv; q < NF_FAR_PERIOD; q += NF_RES_THREADS / 64) {
                const int bb = b - NF_FAR_PERIOD + q;
                unsigned long long kk = s_hist[q];
                while (kk) {
                    const int cl = __builtin_ctzll(kk);
                    kk &= kk - 1ull;
                    const unsigned long long *row = mask + (long)(bb * 64 + cl) * W;
                    for (int w = bb + NF_NEAR_END + lane; w < W; w += 64) {
                        const unsigned long long v = row[w];
                        if (v) atomicOr(&s_removed[w], v);
                    }
                }
            }
            __syncthreads();
        }
        if (wv == 0) {
            const unsigned long long valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
            // (the walk runs on the scalar unit: its state is made wave-uniform in SGPRs here)
            const unsigned long long rem_lds = s_removed[b];
            unsigned long long rem = (((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(rem_lds >> 32)) << 32) |
                                      (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)rem_lds)) | carry[0];
            unsigned long long kept = 0ull;                                       // words of blocks <= b - 4 | of blocks b - 3 .. b - 1
#pragma unroll
            for (int f = 0; f + 1 < NF_FAST; ++f) carry[f] = carry[f + 1];
            carry[NF_FAST - 1] = 0ull;
            unsigned long long cand = ~rem & valid;
            while (cand) {                                                          // wave-uniform
                const int cl = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(cand));
                kept |= 1ull << cl;
                rem |= readlane_u64(diag, cl);
                cand = ~rem & valid & ~((2ull << cl) - 1ull);                       // rows behind cl that are still standing
            }
            // the kept rows' next words: one OR over the wave per word (inside the walk they were six more v_readlane per kept row)
            const bool mine = (kept >> lane) & 1ull;
#pragma unroll
            for (int f = 0; f < NF_FAST; ++f) carry[f] |= wave_or_u64(mine ? nextw[f] : 0ull);
            if ((kept >> lane) & 1ull) keep[nk + __popcll(kept & ((1ull << lane) - 1ull))] = r0 + lane;
            if (lane == 0) { s_kept[b & 1] = kept; s_hist[b % NF_FAR_PERIOD] = kept; }
            // the next block's words, for all of its rows: independent of every decision, fetched BEHIND this block's walk (in front of it the walk's first use of `diag` made the compiler wait for them as well)
            unsigned long long ndiag = 0ull, nnext[NF_FAST] = {0ull, 0ull, 0ull};
            const long row1 = (long)r0 + 64 + lane;
            if (b + 1 < W && row1 < n) {
                ndiag = mask[row1 * W + b + 1];
#pragma unroll
                for (int f = 0; f < NF_FAST; ++f)
                    if (b + 2 + f < W) nnext[f] = mask[row1 * W + b + 2 + f];
            }
            diag = ndiag;
#pragma unroll
            for (int f = 0; f < NF_FAST; ++f) nextw[f] = nnext[f];
        }
        lds_barrier();                                 // block b's kept rows are known (LDS only: the words in flight stay in flight)
        const unsigned long long kept = s_kept[b & 1];
        nk += __popcll(kept);
        if (wv != 0) {
            // consume what was fetched two blocks ago (block b - 2: words b + 2 .. ), then fetch this block's words b + 1 + NF_FAST ..
            if (b >= 2) {
#pragma unroll
                for (int k = 0; k < NF_PASSES; ++k) {
                    const int w = (b - 2) + 1 + NF_FAST + lane + 64 * k;
                    unsigned long long acc = 0ull;
#pragma unroll
                    for (int q = 0; q < 5; ++q) acc |= rawO[q][k];
                    if (w < W && acc) atomicOr(&s_removed[w], acc);
                }
            }
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int k = 0; k < NF_PASSES; ++k) rawO[q][k] = rawN[q][k];
#pragma unroll
            for (int k = 0; k < NF_PASSES; ++k) {
                const int w = b + 1 + NF_FAST + lane + 64 * k;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int cl = (wv - 1) + 15 * q;         // 15 waves: wave v takes rows v - 1, v + 14, ... of the block
                    rawN[q][k] = (w < W && cl < 64 && ((kept >> cl) & 1ull)) ? mask[(long)(r0 + cl) * W + w] : 0ull;
                }
            }
        }
    }
    if (t == 0) *num_keep = nk;
}

static int nms_full(int n, const float *boxes, float thresh, int rotated, int *keep, int *num_keep, hipStream_t st)
{
    const int W = ceil_div(n, 64);
    unsigned long long *mask = (unsigned long long *)scratch_for(st, (size_t)n * W * sizeof(unsigned long long), 10);
    if (!mask) { set_error("nms: cannot allocate %zu bytes of mask scratch", (size_t)n * W * 8); return PRCNN_ELAUNCH; }
    const long pairs = (long)W * (W + 1) / 2;
    if (rotated)
        hipLaunchKernelGGL(nms_full_mask_kernel<true>, dim3((unsigned)pairs), dim3(64 * NF_WAVES), 0, st, n, W, boxes, thresh, mask);
    else
        hipLaunchKernelGGL(nms_full_mask_kernel<false>, dim3((unsigned)pairs), dim3(64 * NF_WAVES), 0, st, n, W, boxes, thresh, mask);
    hipLaunchKernelGGL(nms_full_resolve_kernel, dim3(1), dim3(NF_RES_THREADS), 0, st, n, W, mask, keep, num_keep);
    return check_launch("nms(full mask)");
}

static int nms_blocking(int n, const float *boxes, long long *keep_host, float thresh, int rotated, hipStream_t st)
{
    PRCNN_REQUIRE(n >= 0, "nms: negative box count");
    if (n == 0) return 0;
    PRCNN_REQUIRE(boxes && keep_host, "nms: null pointer");
    int *g_scratch = (int *)scratch_for(st, ((size_t)n + 1) * sizeof(int), 7);
    if (!g_scratch) { set_error("nms: cannot allocate %zu bytes of scratch", ((size_t)n + 1) * sizeof(int)); return PRCNN_ELAUNCH; }
    const bool full_form = true;
    PRCNN_REQUIRE(n <= NMS_MAX_N, "nms: %d boxes > %d unsupported", n, NMS_MAX_N);
    int rc = full_form && n > ND_MAX ? nms_full(n, boxes, thresh, rotated, g_scratch + 1, g_scratch, st)
                                     : nms_device(1, n, nullptr, boxes, thresh, rotated, n, g_scratch + 1, g_scratch, st);
    if (rc != PRCNN_OK) return rc;
    int *host = (int *)malloc(sizeof(int) * ((size_t)n + 1));
    if (!host) { set_error("nms: host allocation failed"); return PRCNN_ELAUNCH; }
    hipError_t e = hipMemcpyAsync(host, g_scratch, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        free(host);
        set_error("nms: copy back failed: %s", hipGetErrorString(e));
        return PRCNN_ELAUNCH;
    }
    const int k = host[0];
    for (int i = 0; i < k; ++i) keep_host[i] = host[1 + i];
    free(host);
    return k;
}

}  // namespace prcnn

using namespace prcnn;

extern "C" int prcnn_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                       float *ans_overlap, void *stream)
{
    PRCNN_REQUIRE(num_a >= 0 && num_b >= 0, "boxes_overlap_bev: bad sizes");
    if (num_a == 0 || num_b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(boxes_a && boxes_b && ans_overlap, "boxes_overlap_bev: null pointer");
    hipLaunchKernelGGL(pair_kernel<false>, dim3(ceil_div((long)num_a * num_b, 256)), dim3(256), 0,
                       (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans_overlap);
    return check_launch("boxes_overlap_bev");
}

extern "C" int prcnn_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                   float *ans_iou, void *stream)
{
    PRCNN_REQUIRE(num_a >= 0 && num_b >= 0, "boxes_iou_bev: bad sizes");
    if (num_a == 0 || num_b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(boxes_a && boxes_b && ans_iou, "boxes_iou_bev: null pointer");
    hipLaunchKernelGGL(pair_kernel<true>, dim3(ceil_div((long)num_a * num_b, 256)), dim3(256), 0,
                       (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans_iou);
    return check_launch("boxes_iou_bev");
}

extern "C" int prcnn_nms(int boxes_num, const float *boxes, long long *keep_host, float thresh, void *stream)
{
    return nms_blocking(boxes_num, boxes, keep_host, thresh, 1, (hipStream_t)stream);
}

extern "C" int prcnn_nms_normal(int boxes_num, const float *boxes, long long *keep_host, float thresh, void *stream)
{
    return nms_blocking(boxes_num, boxes, keep_host, thresh, 0, (hipStream_t)stream);
}

extern "C" int prcnn_nms_device(int nprob, int n_max, const int *counts, const float *boxes, float thresh,
                                int rotated, int max_keep, int *keep, int *num_keep, void *stream)
{
    return nms_device(nprob, n_max, counts, boxes, thresh, rotated, max_keep, keep, num_keep, (hipStream_t)stream);
}
