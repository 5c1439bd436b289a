// roipool.hip -- RoI point pooling (K14-K16) for gfx950.
//
// Reference behaviour restated: lib/utils/roipool3d/src/roipool3d_kernel.cu:14-28 (point in box),
// :97-120 (dense B*N*M assignment matrix in a cudaMalloc'ed temp), :123-160 (one thread per box
// scanning its column of that matrix for the first S hits, wrap-around fill), :163-194 (gather),
// launcher :209-237.
//
// Design: one workgroup per (scene, box).  The 4 waves sweep the cloud 256 points at a time with
// coalesced loads; each wave ballots its in-box lanes (already in index order), the per-wave
// popcounts are prefix-summed through LDS, and the selected indices are appended to an LDS list.
// The sweep stops as soon as S points are found.  The same workgroup then copies the S rows
// (3 xyz + C feature floats, contiguous in the point-major feature tensor) with coalesced
// loads/stores, applying the wrap-around duplication on the fly.  No assignment matrix, no
// temporary allocation, nothing written for empty boxes except their flag.
#include "common.hpp"
#include <math.h>
#include <stdint.h>

namespace prcnn {

constexpr int RP_THREADS = 256;
constexpr int RP_RANK_MAX = 512;    // hits of a box placed by counting (above: bitonic sort)
constexpr int RP_MAX_S = 2048;

// Sweep of one (scene, box): first `sampled` in-box point indices in index order -> s_sel; returns min(#hits, sampled).
// bx = the (already enlarged) box [x, y_bottom, z, h, w, l, ry].
// Each of the 4 waves sweeps its own contiguous QUARTER of the cloud, 64 points per round, and appends its hits (already
// in index order: ballot + lane prefix) to a private LDS list -- no barrier inside the sweep (round 1 swept 256 points
// per round with two workgroup barriers each: 64 rounds x 2 barriers per box made the kernel latency-bound at 0.25 of
// the HBM roofline).  The four lists are concatenated in wave order afterwards = index order.  A wave stops early once
// it alone holds `sampled` hits.  LDS (dynamic): s_sel[sampled] | s_part[4][sampled] | s_cnt[4].
__device__ __forceinline__ int select_points(int pts_num, int sampled, const float *__restrict__ pts, float bx0, float bx1,
                                             float bx2, float h, float w, float l, float ry, int *s_sel, int *s_part, int *s_cnt)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // pt_in_box3d :14-28; the trig pair and cy depend on the box only
    const float cx = bx0, cz = bx2;
    const float cy = (float)((double)bx1 - (double)h / 2.0);
    const float cosa = cos_f32(ry), sina = sin_f32(ry);
    const float hh = h * 0.5f, hl = l * 0.5f, hw = w * 0.5f;  // exact halves
    const int quarter = ((pts_num + 4 * 64 - 1) / (4 * 64)) * 64;          // multiple of 64
    const int lo = wave * quarter, hi = min(pts_num, lo + quarter);
    int *mine = s_part + wave * sampled;
    int cnt = 0;
    // four 64-point rounds per iteration: their 12 loads per lane are in flight together, the tests and the in-order appends
    // follow (a wave sweeps its quarter as a chain of dependent round trips otherwise)
    for (int base = lo; base < hi && cnt < sampled; base += 256) {
        float px[4], py[4], pz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = base + 64 * u + lane;
            const int kk = k < hi ? k : lo;                                  // (a valid address; the result is masked below)
            px[u] = pts[3 * kk]; py[u] = pts[3 * kk + 1]; pz[u] = pts[3 * kk + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = base + 64 * u + lane;
            bool in = false;
            if (k < hi) {
                const float x = px[u], y = py[u], z = pz[u];
                if (!(fabsf(x - cx) > 10.0f || fabsf(y - cy) > hh || fabsf(z - cz) > 10.0f)) {
                    const float xr = __fadd_rn(__fmul_rn(x - cx, cosa), __fmul_rn(z - cz, -sina));
                    const float zr = __fadd_rn(__fmul_rn(x - cx, sina), __fmul_rn(z - cz, cosa));
                    in = (xr >= -hl) & (xr <= hl) & (zr >= -hw) & (zr <= hw);
                }
            }
            const unsigned long long mask = __ballot(in);
            if (in) {
                const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                if (pos < sampled) mine[pos] = k;
            }
            cnt += __popcll(mask);
        }
    }
    if (lane == 0) s_cnt[wave] = min(cnt, sampled);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < RP_THREADS / 64; ++q) {
        const int cq = s_cnt[q];
        if (q < wave) before += cq;
        total += cq;
    }
    const int n_mine = min(cnt, sampled);
    for (int i = lane; i < n_mine; i += 64)
        if (before + i < sampled) s_sel[before + i] = mine[i];
    __syncthreads();
    return min(total, sampled);
}

// The same selection through the spatial groups of prcnn_point_groups (csrc/fps.hip): 256 group boxes are tested against the
// box's footprint (conservatively: circumscribed square in x / z, the y slab, 1 mm of slack for the f32 rounding of either
// side), the points of the groups that remain go through the SAME pt_in_box3d arithmetic, and the hits are sorted by original
// index -- what the index-order sweep produces.  Returns -1 when more than `cap` points are inside (a scene-sized box): the
// caller falls back to the sweep.  s_hits: cap ints, s_cand: pts_num / 64 ints, s_misc: 2 ints.
__device__ __forceinline__ int select_points_culled(int pts_num, int sampled, const float4 *__restrict__ pxyz,
                                                    const float4 *__restrict__ aabb, float bx0, float bx1, float bx2, float h, float w,
                                                    float l, float ry, int *s_sel, int *s_hits, int cap, int *s_cand, int *s_misc)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float cx = bx0, cz = bx2;
    const float cy = (float)((double)bx1 - (double)h / 2.0);
    const float cosa = cos_f32(ry), sina = sin_f32(ry);
    const float hh = h * 0.5f, hl = l * 0.5f, hw = w * 0.5f;
    const float rad = fminf(sqrtf(hl * hl + hw * hw) * 1.0001f, 10.0f) + 1e-3f, ys = hh + 1e-3f;
    const int groups = pts_num / 64;
    if (t < 2) s_misc[t] = 0;
    __syncthreads();
    for (int g = t; g < groups; g += RP_THREADS) {
        const float4 lo = aabb[2 * g], hi = aabb[2 * g + 1];
        const bool hit = !(lo.x > cx + rad || hi.x < cx - rad || lo.z > cz + rad || hi.z < cz - rad || lo.y > cy + ys || hi.y < cy - ys);
        if (hit) s_cand[atomicAdd(&s_misc[0], 1)] = g;
    }
    __syncthreads();
    const int ncand = s_misc[0];
    // four candidate groups of a wave per round: their loads are in flight together (one dependent round trip per group otherwise)
    for (int c0 = wave; c0 < ncand; c0 += 4 * (RP_THREADS / 64)) {
        float4 pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * (RP_THREADS / 64);
            pp[u] = pxyz[(long)s_cand[c < ncand ? c : c0] * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u * (RP_THREADS / 64) >= ncand) break;                 // wave-uniform
            const float4 p = pp[u];
            const float x = p.x, y = p.y, z = p.z;
            bool in = false;
            if (!(fabsf(x - cx) > 10.0f || fabsf(y - cy) > hh || fabsf(z - cz) > 10.0f)) {
                const float xr = __fadd_rn(__fmul_rn(x - cx, cosa), __fmul_rn(z - cz, -sina));
                const float zr = __fadd_rn(__fmul_rn(x - cx, sina), __fmul_rn(z - cz, cosa));
                in = (xr >= -hl) & (xr <= hl) & (zr >= -hw) & (zr <= hw);
            }
            const unsigned long long mask = __ballot(in);
            int base = 0;
            if (lane == 0 && mask) base = atomicAdd(&s_misc[1], __popcll(mask));
            base = __shfl(base, 0, 64);
            if (in) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (pos < cap) s_hits[pos] = __float_as_int(p.w);
            }
        }
    }
    __syncthreads();
    const int total = s_misc[1];
    // s_misc shares its words with the sweep's s_cnt -- every wave must have read `total` before a wave that falls back
    // starts writing them
    __syncthreads();
    if (total > cap) return -1;
    if (total <= RP_RANK_MAX) {
        // the usual case (54-106 points per RoI): a hit's place in index order = the number of hits with a smaller index (indices are
        // distinct) -- thread t counts them for hit t over the list in LDS, four per 16-byte broadcast read, and writes the hit
        // straight into s_sel.  ONE barrier instead of the bitonic network's log2(P) (log2(P) + 1) / 2 = 21-45 (7.7 of the kernel's
        // 40 us: profiles/r04_microbench.md, roipool phases)
        const int padded = (total + 3) & ~3;
        for (int i = total + t; i < padded; i += RP_THREADS) s_hits[i] = 0x7fffffff;
        __syncthreads();
        for (int i = t; i < total; i += RP_THREADS) {
            const int mine = s_hits[i];
            int rank = 0;
            if ((sampled & 3) == 0) {                                          // s_hits starts 9 * sampled ints into the LDS block: 16-byte aligned
                for (int q = 0; q < padded; q += 4) {
                    const int4 o = *reinterpret_cast<const int4 *>(s_hits + q);
                    rank += (o.x < mine) + (o.y < mine) + (o.z < mine) + (o.w < mine);
                }
            } else {
                for (int q = 0; q < total; ++q) rank += s_hits[q] < mine;
            }
            if (rank < sampled) s_sel[rank] = mine;
        }
        __syncthreads();
        return min(total, sampled);
    }
    int P = 64;
    while (P < total) P <<= 1;
    // the sort pads to a power of two, which must fit the `cap` ints of s_hits (sampled not a power of two, or < 16): otherwise fall
    // back to the sweep as well
    if (P > cap) return -1;
    for (int i = total + t; i < P; i += RP_THREADS) s_hits[i] = 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = t; i < P; i += RP_THREADS) {
                const int q = i ^ jj;
                if (q > i) {
                    const int a = s_hits[i], b = s_hits[q];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s_hits[i] = b; s_hits[q] = a; }
                }
            }
            __syncthreads();
        }
    const int cnt = min(total, sampled);
    for (int i = t; i < cnt; i += RP_THREADS) s_sel[i] = s_hits[i];
    __syncthreads();
    return cnt;
}

__global__ __launch_bounds__(RP_THREADS) void roipool3d_kernel(
    int pts_num, int boxes_num, int feat_len, int sampled, const float *__restrict__ xyz,
    const float *__restrict__ boxes3d, const float *__restrict__ pts_feature,
    float *__restrict__ pooled, int *__restrict__ empty_flag)
{
    extern __shared__ __align__(16) int rp_lds[];
    int *s_sel = rp_lds, *s_part = rp_lds + sampled, *s_cnt = rp_lds + 5 * sampled;

    const int box = blockIdx.x, b = blockIdx.y;
    const int t = threadIdx.x;
    const float *bx = boxes3d + ((long)b * boxes_num + box) * 7;
    const float *__restrict__ pts = xyz + (long)b * pts_num * 3;
    const int cnt = select_points(pts_num, sampled, pts, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], bx[6], s_sel, s_part, s_cnt);
    if (cnt == 0) {
        if (t == 0) empty_flag[(long)b * boxes_num + box] = 1;
        return;  // rows stay as the caller left them
    }

    const int width = 3 + feat_len;
    float *__restrict__ dst = pooled + ((long)b * boxes_num + box) * (long)sampled * width;
    const float *__restrict__ feat = pts_feature + (long)b * pts_num * feat_len;
    // element e = (slot s, column j); consecutive e are consecutive in the output
    const long elems = (long)sampled * width;
    for (long e = t; e < elems; e += RP_THREADS) {
        const int s = (int)(e / width);
        const int j = (int)(e - (long)s * width);
        const int k = s_sel[s < cnt ? s : s % cnt];  // wrap-around duplication :152-159
        dst[e] = j < 3 ? pts[3 * k + j] : feat[(long)k * feat_len + (j - 3)];
    }
}

// RCNN input assembly in the same pass (rcnn_net.py:139-163 with roipool3d_utils.py:7-28 and
// kitti_utils.py:150-160 enlarge_box3d): the box is enlarged here, the pooled coordinates are moved into the
// RoI's canonical frame (centre subtracted, rotated by the RoI heading about y), and the row is written in the
// layout the RCNN point MLPs consume:  [x', y', z', seg mask, depth, 0, 0, 0 | C features]  (C % 4 == 0, every
// 16-byte chunk aligned).  The caller does not pre-clear: empty boxes are written here too.  Selection is the same code
// as roipool3d_kernel, hence the same points.
__global__ __launch_bounds__(RP_THREADS) void roipool3d_canonical_kernel(
    int pts_num, int boxes_num, int feat_len, int sampled, float extra, float extra2, const float *__restrict__ xyz,
    const float *__restrict__ rois, const float *__restrict__ feats, const float *__restrict__ seg_mask,
    const float *__restrict__ depth, float4 *__restrict__ pooled, int *__restrict__ empty_flag, int *__restrict__ pooled_cnt,
    const float4 *__restrict__ pxyz, const float4 *__restrict__ aabb, float *__restrict__ xyz_out)
{
    extern __shared__ __align__(16) int rp_lds[];
    float4 *s_c01 = reinterpret_cast<float4 *>(rp_lds);               // [sampled][2]: the two leading chunks of every distinct row
    int *s_sel = rp_lds + 8 * sampled, *s_part = s_sel + sampled, *s_cnt = s_sel + 5 * sampled;
    int *s_cand = s_sel + 5 * sampled + 4;                              // pts_num / 64 ints when the spatial groups are given

    const int box = blockIdx.x, b = blockIdx.y;
    const int t = threadIdx.x;
    const float *bx = rois + ((long)b * boxes_num + box) * 7;
    const float rx = bx[0], ry_bottom = bx[1], rz = bx[2], heading = bx[6];
    const float *__restrict__ pts = xyz + (long)b * pts_num * 3;
    // enlarge_box3d: h, w, l += 2 * extra_width;  y += extra_width
    int cnt = -1;
    if (pxyz)
        cnt = select_points_culled(pts_num, sampled, pxyz + (long)b * pts_num, aabb + (long)b * (pts_num / 64) * 2, rx, ry_bottom + extra, rz,
                                   bx[3] + extra2, bx[4] + extra2, bx[5] + extra2, heading, s_sel, s_part, 4 * sampled, s_cand, s_cnt);
    if (cnt < 0)                                                         // no groups given, or a box holding more than 4 x sampled points
        cnt = select_points(pts_num, sampled, pts, rx, ry_bottom + extra, rz, bx[3] + extra2, bx[4] + extra2,
                            bx[5] + extra2, heading, s_sel, s_part, s_cnt);
    const int q4 = 2 + feat_len / 4;                                     // float4 chunks per row
    float4 *__restrict__ dst = pooled + ((long)b * boxes_num + box) * (long)sampled * q4;
    const long chunks = (long)sampled * q4;
    const float cosa = cosf(heading), sina = sinf(heading);             // torch.cos / torch.sin of the f32 heading
    // pooled_cnt (optional): number of DISTINCT rows of this box = min(#points in the box, sampled), at least 1.  Rows s >= cnt
    // are copies of row s % cnt (roipool3d_kernel.cu:152-159); with pooled_cnt given their 128 feature columns are written
    // only up to the next multiple of 64 rows -- the consumers (rcnn_point_mlp over the live tiles, SA1 over the distinct
    // rows) never read beyond -- while the coordinate / mask / depth chunks of ALL rows are written (FPS and the ball query
    // of SA1 run over all `sampled` points, copies included, exactly as the reference does).
    if (pooled_cnt && t == 0) pooled_cnt[(long)b * boxes_num + box] = max(cnt, 1);
    const int feat_rows = pooled_cnt ? ((max(cnt, 1) + 63) & ~63) : sampled;
    if (cnt == 0) {
        // the reference leaves zero rows and then applies the canonical transform to ALL rows (rcnn_net.py:147-156):
        // an empty box holds the image of the origin, zero features
        if (t == 0) empty_flag[(long)b * boxes_num + box] = 1;
        const float x = 0.f - rx, y = 0.f - ry_bottom, z = 0.f - rz;
        const float4 origin = make_float4(fmaf(z, -sina, __fmul_rn(x, cosa)), y, fmaf(z, cosa, __fmul_rn(x, sina)), 0.f);
        for (long e = t; e < chunks; e += RP_THREADS) {
            const int q = (int)(e % q4);
            if (q >= 2 && e / q4 >= feat_rows) continue;
            dst[e] = (q == 0) ? origin : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (xyz_out) {
            float *xo = xyz_out + ((long)b * boxes_num + box) * (long)sampled * 3;
            for (int e = t; e < sampled; e += RP_THREADS) { xo[3 * e] = origin.x; xo[3 * e + 1] = origin.y; xo[3 * e + 2] = origin.z; }
        }
        return;
    }
    if (t == 0) empty_flag[(long)b * boxes_num + box] = 0;
    const float4 *__restrict__ feat4 = reinterpret_cast<const float4 *>(feats + (long)b * pts_num * feat_len);
    const float *__restrict__ mk = seg_mask + (long)b * pts_num;
    const float *__restrict__ dp = depth + (long)b * pts_num;
    const int f4 = feat_len / 4;
    // The two leading chunks of a row (canonical coordinates + mask, depth) are computed ONCE per distinct row into LDS; every
    // output row s -- a distinct row or a wrap-around copy s % cnt -- takes them from there.  The 128 feature columns are read
    // from the points' feature rows for the rows below feat_rows only, eight independent 16-byte gathers per thread in flight
    // (the loop used to be one dependent LDS -> gather -> store chain per iteration with an integer division and a modulo in
    // front: 70 of the kernel's 90 us).  (Sending the first batch out in FRONT of the coordinate phase was tried: 48 instead of 40 us --
    // the phase's own dependent loads queue behind the eight gathers, and the barrier drains them anyway.)
    for (int s2 = t; s2 < cnt; s2 += RP_THREADS) {
        const int k = s_sel[s2];
        const float x = pts[3 * k] - rx, y = pts[3 * k + 1] - ry_bottom, z = pts[3 * k + 2] - rz;
        // rotate_pc_along_y_torch (kitti_utils.py:45-63) is a batched (1x2)@(2x2) matmul: the GEMM library
        // accumulates k = 0, 1 with fused multiply-adds, i.e. fma(z, r1, x * r0) -- reproduced here
        s_c01[2 * s2] = make_float4(fmaf(z, -sina, __fmul_rn(x, cosa)), y, fmaf(z, cosa, __fmul_rn(x, sina)), mk[k]);
        s_c01[2 * s2 + 1] = make_float4(dp[k], 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // chunks 0 / 1 of ALL rows: thread pair per row
    for (int e = t; e < 2 * sampled; e += RP_THREADS) {
        const int s2 = e >> 1, q = e & 1;
        const int src = s2 < cnt ? s2 : s2 % cnt;
        dst[(long)s2 * q4 + q] = s_c01[2 * src + q];
    }
    // (optional) the canonical coordinates once more as a dense (sampled, 3) cloud: what the RCNN's sampling and ball queries read --
    // the caller used to cut them out of the 544-byte rows with a strided copy, 30 us on the proposal stream
    if (xyz_out) {
        float *xo = xyz_out + ((long)b * boxes_num + box) * (long)sampled * 3;
        const float *sc = reinterpret_cast<const float *>(s_c01);
        for (int e = t; e < 3 * sampled; e += RP_THREADS) {
            const int s2 = e / 3, c = e - 3 * s2;
            const int src = s2 < cnt ? s2 : s2 % cnt;
            xo[e] = sc[8 * src + c];
        }
    }
    // feature chunks of the rows below feat_rows.  (Sending the first batch of gathers out in FRONT of the coordinate phase, to overlap its
    // loads and its barrier, was tried in round 4: the compiler keeps the eight rows in scratch across the barrier -- 42 instead of 35 us.)
    const float inv_f4 = 1.0f / (float)f4;
    const int nfeat = feat_rows * f4;
    for (int e0 = t; e0 < nfeat; e0 += 8 * RP_THREADS) {
        float4 v[8];
        int row[8], q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * RP_THREADS;
            const int ee = e < nfeat ? e : 0;
            row[u] = __float2int_rz(((float)ee + 0.5f) * inv_f4);          // ee / f4 (exact: ee + 0.5 is never within rounding of a multiple)
            q[u] = ee - row[u] * f4;
            const int src = row[u] < cnt ? row[u] : row[u] % cnt;
            v[u] = feat4[(long)s_sel[src] * f4 + q[u]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u * RP_THREADS < nfeat) dst[(long)row[u] * q4 + 2 + q[u]] = v[u];
    }
}

}  // namespace prcnn

using namespace prcnn;

extern "C" int prcnn_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                               int sampled_pts_num, const float *xyz, const float *boxes3d,
                               const float *pts_feature, float *pooled_features,
                               int *pooled_empty_flag, void *stream)
{
    PRCNN_REQUIRE(batch_size >= 0 && pts_num >= 0 && boxes_num >= 0 && feature_in_len >= 0 && sampled_pts_num >= 0,
                  "roipool3d: bad sizes");
    PRCNN_REQUIRE(sampled_pts_num <= RP_MAX_S, "roipool3d: sampled_pts_num=%d > %d unsupported", sampled_pts_num, RP_MAX_S);
    PRCNN_REQUIRE(batch_size <= 65535, "roipool3d: batch > 65535");
    if (batch_size == 0 || boxes_num == 0 || sampled_pts_num == 0) return PRCNN_OK;
    PRCNN_REQUIRE(boxes3d && pooled_features && pooled_empty_flag && (xyz || pts_num == 0) &&
                  (pts_feature || feature_in_len == 0 || pts_num == 0), "roipool3d: null pointer");
    dim3 grid(boxes_num, batch_size);
    const size_t lds = ((size_t)5 * sampled_pts_num + 4) * sizeof(int);
    hipLaunchKernelGGL(roipool3d_kernel, grid, dim3(RP_THREADS), lds, (hipStream_t)stream, pts_num, boxes_num,
                       feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,
                       pooled_empty_flag);
    return check_launch("roipool3d");
}


// Fast-path RCNN input assembly (see roipool3d_canonical_kernel).  xyz (b,n,3), rois (b,m,7) NOT enlarged,
// feats (b,n,c) point-major with c % 4 == 0, seg_mask / depth (b,n) -> pooled (b,m,sampled,8+c), empty (b,m) i32.
// xyz_out (optional, (b, m, sampled, 3)): the rows' canonical coordinates as dense clouds as well.
extern "C" int prcnn_roipool3d_canonical_xyz(int batch_size, int pts_num, int boxes_num, int feature_len, int sampled_pts_num,
                                             float pool_extra_width, const float *xyz, const float *rois, const float *feats,
                                             const float *seg_mask, const float *depth, float *pooled, int *pooled_empty_flag,
                                             int *pooled_cnt, const float *pxyz, const float *aabb, float *xyz_out, void *stream);
extern "C" int prcnn_roipool3d_canonical(int batch_size, int pts_num, int boxes_num, int feature_len, int sampled_pts_num,
                                         float pool_extra_width, const float *xyz, const float *rois, const float *feats,
                                         const float *seg_mask, const float *depth, float *pooled, int *pooled_empty_flag,
                                         int *pooled_cnt, const float *pxyz, const float *aabb, void *stream)
{
    return prcnn_roipool3d_canonical_xyz(batch_size, pts_num, boxes_num, feature_len, sampled_pts_num, pool_extra_width, xyz, rois, feats,
                                         seg_mask, depth, pooled, pooled_empty_flag, pooled_cnt, pxyz, aabb, nullptr, stream);
}

extern "C" int prcnn_roipool3d_canonical_xyz(int batch_size, int pts_num, int boxes_num, int feature_len, int sampled_pts_num,
                                             float pool_extra_width, const float *xyz, const float *rois, const float *feats,
                                             const float *seg_mask, const float *depth, float *pooled, int *pooled_empty_flag,
                                             int *pooled_cnt, const float *pxyz, const float *aabb, float *xyz_out, void *stream)
{
    PRCNN_REQUIRE((pxyz == nullptr) == (aabb == nullptr), "roipool3d_canonical: pxyz and aabb go together (prcnn_point_groups)");
    PRCNN_REQUIRE(!pxyz || (pts_num % 64 == 0 && pts_num <= 65536 && (((uintptr_t)pxyz | (uintptr_t)aabb) & 15) == 0),
                  "roipool3d_canonical: spatial groups need pts_num a multiple of 64, <= 65536");
    PRCNN_REQUIRE(batch_size >= 0 && pts_num >= 0 && boxes_num >= 0 && feature_len >= 0 && sampled_pts_num >= 0,
                  "roipool3d_canonical: bad sizes");
    PRCNN_REQUIRE(feature_len % 4 == 0, "roipool3d_canonical: feature length %d is not a multiple of 4", feature_len);
    PRCNN_REQUIRE(sampled_pts_num <= RP_MAX_S, "roipool3d_canonical: sampled_pts_num=%d > %d unsupported", sampled_pts_num, RP_MAX_S);
    PRCNN_REQUIRE(batch_size <= 65535, "roipool3d_canonical: batch > 65535");
    if (batch_size == 0 || boxes_num == 0 || sampled_pts_num == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rois && pooled && pooled_empty_flag && (pts_num == 0 || (xyz && seg_mask && depth && (feats || feature_len == 0))),
                  "roipool3d_canonical: null pointer");
    PRCNN_REQUIRE((((uintptr_t)pooled | (uintptr_t)feats) & 15) == 0, "roipool3d_canonical: 16-byte alignment required");
    dim3 grid(boxes_num, batch_size);
    const size_t lds = ((size_t)13 * sampled_pts_num + 4 + (pxyz ? pts_num / 64 : 0)) * sizeof(int);   // s_c01 | s_sel | s_part | s_cnt | s_cand
    if (lds > 64 * 1024) {
        const int rc = ensure_dynamic_lds((const void *)roipool3d_canonical_kernel, lds, "roipool3d_canonical");
        if (rc != PRCNN_OK) return rc;
    }
    hipLaunchKernelGGL(roipool3d_canonical_kernel, grid, dim3(RP_THREADS), lds, (hipStream_t)stream, pts_num, boxes_num,
                       feature_len, sampled_pts_num, pool_extra_width, (float)((double)pool_extra_width * 2.0), xyz, rois, feats,
                       seg_mask, depth, reinterpret_cast<float4 *>(pooled), pooled_empty_flag, pooled_cnt,
                       reinterpret_cast<const float4 *>(pxyz), reinterpret_cast<const float4 *>(aabb), xyz_out);
    return check_launch("roipool3d_canonical");
}
