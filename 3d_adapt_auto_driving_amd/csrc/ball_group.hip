// ball_group.hip -- ball query (K1), grouping (K2/K3), gather (K4/K5) and the fused
// QueryAndGroup path for gfx950.
//
// Reference behaviour restated (never copied): pointnet2_lib/pointnet2/src/ball_query_gpu.cu:9-45,
// group_points_gpu.cu:8-66, sampling_gpu.cu:8-63, pointnet2_utils.py:241-264.
//
// Ball query design (wave64, brute force but index-order exact):
//   * one lane owns one query centre for the whole scan; the cloud point is wave-uniform, so it
//     is fetched with SCALAR loads (s_load_dwordx4) and used as an SGPR operand of the VALU
//     distance ops -- no LDS or vector-memory traffic in the inner loop;
//   * a workgroup = NSEG waves that own the SAME 64 centres but disjoint, contiguous index
//     segments of the cloud (fills the chip when b*m/64 alone is < #SIMDs);
//   * each wave appends its hits (already in index order) to an LDS list laid out
//     [segment][slot][lane] (bank = lane -> conflict free) and stops early once all of its
//     lanes hold nsample hits;
//   * after one barrier the segment lists are concatenated in segment order = global index
//     order, truncated to nsample and back-filled with the first hit, and written with fully
//     coalesced stores.
#include "common.hpp"
#include <stdlib.h>

namespace prcnn {

template <int NSEG>
__global__ __launch_bounds__(64 * NSEG) void ball_query_kernel(
    int n, int m, float r2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx, int write_empty, const int *__restrict__ limit)
{
    extern __shared__ int lds[];  // [NSEG][nsample][64] hits, then [NSEG][64] counts
    int *counts = lds + NSEG * nsample * 64;

    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = c0 + lane;
    const bool valid = p < m;

    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (valid) {
        const float *c = new_xyz + ((long)b * m + p) * 3;
        cx = c[0]; cy = c[1]; cz = c[2];
    }
    const float *__restrict__ cloud = xyz + (long)b * n * 3;
    // limit (optional): only the first limit[b] points of the cloud are scanned (prcnn_ball_query_limit)
    const int n_scan = limit ? min(n, max(limit[b], 1)) : n;
    const int seg_len = (n_scan + NSEG - 1) / NSEG;
    const int k0 = seg * seg_len;
    const int k1 = min(n_scan, k0 + seg_len);

    int cnt = valid ? 0 : nsample;  // lanes past m never record anything
    int *myhits = lds + seg * nsample * 64 + lane;

    for (int kb = k0; kb < k1; kb += 64) {
        if (__all(cnt >= nsample)) break;
        // 64 points per round: ONE vector load per coordinate (lane j holds point kb + j), then point j reaches every lane as
        // an SGPR operand through v_readlane.  (Rounds 1-2 fetched every point with three scalar loads: their latency was most
        // of the loop on the 800 RoI clouds of a batch -- 91 us per call for 8 M distance tests.)
        const int kk = kb + lane;
        float vx = 0.f, vy = 0.f, vz = 0.f;
        if (kk < k1) { vx = cloud[3 * kk]; vy = cloud[3 * kk + 1]; vz = cloud[3 * kk + 2]; }
        const int nb = min(64, k1 - kb);
        for (int j = 0; j < nb; ++j) {
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx), j));
            const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy), j));
            const float z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz), j));
            const float d2 = sqdist3(cx, cy, cz, x, y, z);
            if (d2 < r2 && cnt < nsample) {
                myhits[cnt * 64] = kb + j;
                ++cnt;
            }
        }
    }
    counts[seg * 64 + lane] = valid ? cnt : 0;
    __syncthreads();

    // merge: element e = (centre cl, slot s); consecutive e -> consecutive idx addresses
    const int total_e = min(64, m - c0) * nsample;
    int *out = idx + ((long)b * m + c0) * nsample;
    for (int e = threadIdx.x; e < total_e; e += 64 * NSEG) {
        const int cl = e / nsample;
        const int s = e - cl * nsample;
        int tot = 0;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) tot += counts[g * 64 + cl];
        if (tot == 0) {
            if (write_empty) out[e] = 0;
            continue;
        }
        int want = s < tot ? s : 0;  // slots past the hit count repeat the first hit
        int v = 0;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) {
            const int cg = counts[g * 64 + cl];
            if (want >= 0 && want < cg) v = lds[(g * nsample + want) * 64 + cl];
            want -= cg;  // becomes negative once consumed
        }
        out[e] = v;
    }
}

// K2: out[b][c][slot] = points[b][c][idx[b][slot]]
__global__ __launch_bounds__(256) void group_points_kernel(
    int c, int n, long slots, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int ci = blockIdx.y;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= slots) return;
    const int k = idx[(long)b * slots + s];
    out[((long)b * c + ci) * slots + s] = points[((long)b * c + ci) * n + k];
}

// K3: grad_points[b][c][idx[b][slot]] += grad_out[b][c][slot]
__global__ __launch_bounds__(256) void group_points_grad_kernel(
    int c, int n, long slots, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points)
{
    const int b = blockIdx.z;
    const int ci = blockIdx.y;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= slots) return;
    const int k = idx[(long)b * slots + s];
    atomicAdd(grad_points + ((long)b * c + ci) * n + k, grad_out[((long)b * c + ci) * slots + s]);
}

// Fused grouping for QueryAndGroup: one thread per (centre, sample) slot walks the channels.
// Channel 0..2 = xyz[idx] - new_xyz (exact f32 subtract), 3.. = features[idx].
__global__ __launch_bounds__(256) void group_cat_kernel(
    int n, int m, int c, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, const float *__restrict__ features,
    const int *__restrict__ idx, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const long slots = (long)m * nsample;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= slots) return;
    const int p = (int)(s / nsample);
    const int k = idx[(long)b * slots + s];
    const float *pt = xyz + ((long)b * n + k) * 3;
    const float *ct = new_xyz + ((long)b * m + p) * 3;
    float *o = out + (long)b * (3 + c) * slots + s;
    o[0] = pt[0] - ct[0];
    o[slots] = pt[1] - ct[1];
    o[2 * slots] = pt[2] - ct[2];
    const float *f = features + (long)b * c * n + k;
    o += 3 * slots;
    for (int ci = 0; ci < c; ++ci) o[(long)ci * slots] = f[(long)ci * n];
}

// Channel-major grouping at HBM speed: the source rows of ROWS output channels of one scene are
// staged in LDS (ROWS * N floats), so the data-dependent gather is an LDS read (ds_read_b32, a few
// cycles even with bank conflicts) instead of 64 different cache lines per wave-load; idx is read
// with 16-byte loads, the (B, 3+C, M, ns) output is written with 16-byte stores on consecutive
// addresses.  Output channels 0..2 are xyz[idx] - centre, channels 3.. are features[idx]
// (pointnet2_utils.py:249-257).  grid = (row groups, slot chunks, B); block = 1024 threads.
// CAT = false: plain group_points (K2, group_points_gpu.cu:47-66) -- every output channel is a feature row, nothing is subtracted.
// CAT = true (round 4): a workgroup owns ROWS FEATURE rows (output channels 3 + ...) and, beside them, a 1/gridDim.x slice of the
// slots of the three COORDINATE channels, which it gathers straight from the (L2-resident, 196 KB) cloud -- 3 of 131 channels,
// 12 bytes per slot.  Before, the coordinate channels were row groups of their own: 131 x 8 = 1048 equal workgroups on 512
// resident slots = two full rounds and a third one of 24 workgroups (116 us against 98 us for the 128-channel group_points).
template <int ROWS, bool CAT = true>
__global__ __launch_bounds__(1024) void group_cat_lds_kernel(
    int n, int m, int c, int nsample, const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    const float *__restrict__ features, const int *__restrict__ idx, float *__restrict__ out)
{
    extern __shared__ float rows[];  // [ROWS][n]
    const int b = blockIdx.z;
    constexpr int NX = CAT ? 3 : 0;            // leading coordinate channels
    const int ch0 = NX + blockIdx.x * ROWS;
    const int cout = NX + c;
    const long slots = (long)m * nsample;
    const int t = threadIdx.x;

#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int ch = ch0 + r;
        if (ch >= cout) break;
        float *dst = rows + (long)r * n;
        const float *src = features + ((long)b * c + (ch - NX)) * n;
        for (int k = t; k < n; k += 1024) dst[k] = src[k];
    }
    const long quads = slots >> 2;
    const int4 *idx4 = reinterpret_cast<const int4 *>(idx + (long)b * slots);
    if (CAT) {
        // this workgroup's slice of the coordinate channels (before the barrier: it does not touch LDS)
        const long parts = (long)gridDim.x * gridDim.y, part = (long)blockIdx.y * gridDim.x + blockIdx.x;
        const long per = (quads + parts - 1) / parts;
        const long a0 = part * per, a1 = min(quads, a0 + per);
        const float *pts = xyz + (long)b * n * 3;
        const float *ctr = new_xyz + (long)b * m * 3;
        float *o = out + (long)b * cout * slots;
        for (long q = a0 + t; q < a1; q += 1024) {
            const int4 k = idx4[q];
            const long s0 = q << 2;
            const int kk[4] = {k.x, k.y, k.z, k.w};
            float v[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *p = pts + 3 * (long)kk[e];
                const float *cc = ctr + ((s0 + e) / nsample) * 3;
                v[0][e] = p[0] - cc[0]; v[1][e] = p[1] - cc[1]; v[2][e] = p[2] - cc[2];
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float4 *dst = reinterpret_cast<float4 *>(o + (long)ch * slots) + q;
                __builtin_nontemporal_store(v[ch][0], &dst->x);
                __builtin_nontemporal_store(v[ch][1], &dst->y);
                __builtin_nontemporal_store(v[ch][2], &dst->z);
                __builtin_nontemporal_store(v[ch][3], &dst->w);
            }
        }
        if (part == parts - 1)                             // tail slots (slots % 4)
            for (long sl = (quads << 2) + t; sl < slots; sl += 1024) {
                const int k = idx[(long)b * slots + sl];
                for (int ch = 0; ch < 3; ++ch) o[(long)ch * slots + sl] = pts[3 * (long)k + ch] - ctr[(sl / nsample) * 3 + ch];
            }
    }
    __syncthreads();

    // this block's share of the slots, in units of 4 consecutive slots
    const long per = (quads + gridDim.y - 1) / gridDim.y;
    const long q0 = (long)blockIdx.y * per;
    const long q1 = min(quads, q0 + per);
    constexpr int U = 4;   // quads in flight per thread: idx loads issued first, then gathers, then stores
    for (long qb = q0; qb < q1; qb += 1024 * U) {
        int4 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = qb + t + (long)u * 1024;
            k[u] = q < q1 ? idx4[q] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int ch = ch0 + r;
            if (ch >= cout) break;
            const float *row = rows + (long)r * n;
            float4 *dst = reinterpret_cast<float4 *>(out + ((long)b * cout + ch) * slots);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long q = qb + t + (long)u * 1024;
                if (q >= q1) continue;
                const float4 v = make_float4(row[k[u].x], row[k[u].y], row[k[u].z], row[k[u].w]);
                __builtin_nontemporal_store(v.x, &dst[q].x);
                __builtin_nontemporal_store(v.y, &dst[q].y);
                __builtin_nontemporal_store(v.z, &dst[q].z);
                __builtin_nontemporal_store(v.w, &dst[q].w);
            }
        }
    }
    // tail slots (slots % 4), handled by the last chunk
    if (blockIdx.y == gridDim.y - 1) {
        for (long sl = (quads << 2) + t; sl < slots; sl += 1024) {
            const int k = idx[(long)b * slots + sl];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int ch = ch0 + r;
                if (ch >= cout) break;
                out[((long)b * cout + ch) * slots + sl] = rows[(long)r * n + k];
            }
        }
    }
}

template <int ROWS, bool CAT = true>
static int launch_group_cat_lds(int b, int n, int m, int c, int nsample, const float *new_xyz, const float *xyz,
                                const float *features, const int *idx, float *out, hipStream_t st)
{
    const size_t lds = (size_t)ROWS * n * sizeof(float);
    if (lds > 64 * 1024) {
        const int rc = ensure_dynamic_lds((const void *)group_cat_lds_kernel<ROWS, CAT>, lds, "group_cat");
        if (rc != PRCNN_OK) return rc;
    }
    const int groups = ceil_div(c, ROWS);       // feature row groups (CAT: the coordinate channels ride along as slot slices)
    // enough blocks to fill 256 CUs a few times over; every chunk re-stages the rows, so keep chunks large
    int chunks = 1;
    const long slots = (long)m * nsample;
    while ((long)b * groups * chunks < 1024 && slots / (chunks * 2) >= 16384) chunks *= 2;
    if (const char *e = getenv("PRCNN_GROUP_CHUNKS")) chunks = atoi(e) > 0 ? atoi(e) : chunks;   // tuning knob
    dim3 grid(groups, chunks, b);
    hipLaunchKernelGGL((group_cat_lds_kernel<ROWS, CAT>), grid, dim3(1024), lds, st, n, m, c, nsample, new_xyz, xyz,
                       features, idx, out);
    return check_launch(CAT ? "query_and_group" : "group_points");
}

// K4 / K5
__global__ __launch_bounds__(256) void gather_points_kernel(
    int c, int n, int m, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out)
{
    const int b = blockIdx.z, ci = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= m) return;
    out[((long)b * c + ci) * m + p] = points[((long)b * c + ci) * n + idx[(long)b * m + p]];
}

__global__ __launch_bounds__(256) void gather_points_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points)
{
    const int b = blockIdx.z, ci = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= m) return;
    atomicAdd(grad_points + ((long)b * c + ci) * n + idx[(long)b * m + p],
              grad_out[((long)b * c + ci) * m + p]);
}

static int pick_nseg(int b, int n, int m, int nsample)
{
    const long groups = (long)b * ceil_div(m, 64);
    int nseg = 1;
    // aim for >= 2048 waves (2 per SIMD on 256 CUs), keep segments >= 256 points and the hit
    // lists within 128 KiB of LDS
    while (nseg < 8 && groups * nseg < 2048 && n / (nseg * 2) >= 256 &&
           (long)(nseg * 2) * nsample * 256 + (long)(nseg * 2) * 256 <= 128 * 1024)
        nseg *= 2;
    return nseg;
}

template <int NSEG>
static int launch_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                             const float *xyz, int *idx, int write_empty, hipStream_t st, const int *limit = nullptr)
{
    const size_t lds = ((size_t)NSEG * nsample * 64 + (size_t)NSEG * 64) * sizeof(int);
    if (lds > 64 * 1024) {
        const int rc = ensure_dynamic_lds((const void *)ball_query_kernel<NSEG>, lds, "ball_query");
        if (rc != PRCNN_OK) return rc;
    }
    dim3 grid(ceil_div(m, 64), b);
    hipLaunchKernelGGL(ball_query_kernel<NSEG>, grid, dim3(64 * NSEG), lds, st, n, m,
                       radius * radius, nsample, new_xyz, xyz, idx, write_empty, limit);
    return check_launch("ball_query");
}

int ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                    int *idx, int write_empty, hipStream_t st, int *used);   // ball_grid.hip
int ball_query_dense(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                     int *idx, int write_empty, hipStream_t st, int *used);  // ball_dense.hip

static int g_ball_query_mode = 0;   // 0 auto (bucket-sorted grid, wave per centre), 1 brute force only, 2 linked-list grid (round 1)

static int ball_query_dispatch(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                               const float *xyz, int *idx, int write_empty, hipStream_t st)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample > 0, "ball_query: bad sizes b=%d n=%d m=%d ns=%d", b, n, m, nsample);
    PRCNN_REQUIRE(nsample <= 256, "ball_query: nsample=%d > 256 unsupported", nsample);
    PRCNN_REQUIRE(b <= 65535, "ball_query: batch %d > 65535", b);
    if (b == 0 || m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(new_xyz && xyz && idx, "ball_query: null pointer");
    if (g_ball_query_mode == 0) {
        int used = 0;
        const int rc = ball_query_dense(b, n, m, radius, nsample, new_xyz, xyz, idx, write_empty, st, &used);
        if (rc != PRCNN_OK || used) return rc;
    }
    if (g_ball_query_mode != 1) {
        int used = 0;
        const int rc = ball_query_grid(b, n, m, radius, nsample, new_xyz, xyz, idx, write_empty, st, &used);
        if (rc != PRCNN_OK || used) return rc;
    }
    switch (pick_nseg(b, n, m, nsample)) {
        case 1: return launch_ball_query<1>(b, n, m, radius, nsample, new_xyz, xyz, idx, write_empty, st);
        case 2: return launch_ball_query<2>(b, n, m, radius, nsample, new_xyz, xyz, idx, write_empty, st);
        case 4: return launch_ball_query<4>(b, n, m, radius, nsample, new_xyz, xyz, idx, write_empty, st);
        default: return launch_ball_query<8>(b, n, m, radius, nsample, new_xyz, xyz, idx, write_empty, st);
    }
}

}  // namespace prcnn

using namespace prcnn;

// 0 = automatic (bucket-sorted hashed grid with a wave per centre for n >= 2048, ball_dense.hip; brute force otherwise),
// 1 = brute force only, 2 = the linked-list hashed grid of round 1 (ball_grid.hip, n >= 4096) instead of the bucket-sorted one
extern "C" int prcnn_set_ball_query_mode(int mode)
{
    PRCNN_REQUIRE(mode >= 0 && mode <= 2, "set_ball_query_mode: mode %d", mode);
    g_ball_query_mode = mode;
    return PRCNN_OK;
}

extern "C" int prcnn_ball_query(int b, int n, int m, float radius, int nsample,
                                const float *new_xyz, const float *xyz, int *idx, void *stream)
{
    return ball_query_dispatch(b, n, m, radius, nsample, new_xyz, xyz, idx, 0, (hipStream_t)stream);
}

// The same with EVERY slot written: an empty ball gets the zeros the reference's caller-side zero fill would hold (pointnet2_utils.py:218),
// so the engine need not clear the index tensor first (a fill launch per ball query).
extern "C" int prcnn_ball_query_full(int b, int n, int m, float radius, int nsample,
                                     const float *new_xyz, const float *xyz, int *idx, void *stream)
{
    return ball_query_dispatch(b, n, m, radius, nsample, new_xyz, xyz, idx, 1, (hipStream_t)stream);
}

// Ball query over clouds whose points k >= limit[cloud] are known to be COPIES of point k % limit[cloud] (RoI pooling's
// wrap-around fill, roipool3d_kernel.cu:152-159): only the first limit[cloud] points are scanned.  The idx rows differ from
// prcnn_ball_query's (slots the full scan fills with copies hold the first hit here) but name the same SET of distinct points
// per ball, which is all a max-pooled SA level sees.  Not part of the reference ABI: an engine-side shortcut.
extern "C" int prcnn_ball_query_limit(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                                      const int *limit, int *idx, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample > 0 && nsample <= 256 && b <= 65535, "ball_query_limit: bad sizes");
    if (b == 0 || m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(new_xyz && xyz && idx && limit, "ball_query_limit: null pointer");
    // one wave per 64 centres scans the (short) live part of its cloud
    // (write_empty = 1: every slot of idx is written -- an empty ball gets zeros, the value the reference's zero-filled tensor
    // holds -- so the caller need not clear 26 MB of indices first)
    return launch_ball_query<1>(b, n, m, radius, nsample, new_xyz, xyz, idx, 1, (hipStream_t)stream, limit);
}

extern "C" int prcnn_group_points(int b, int c, int n, int npoints, int nsample,
                                  const float *points, const int *idx, float *out, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "group_points: bad sizes");
    PRCNN_REQUIRE(b <= 65535 && c <= 65535, "group_points: b/c > 65535");
    const long slots = (long)npoints * nsample;
    if (b == 0 || c == 0 || slots == 0) return PRCNN_OK;
    PRCNN_REQUIRE(points && idx && out, "group_points: null pointer");
    // round 4: the channel rows staged in LDS (the kernel of the fused operator without its coordinate channels): the gather is an
    // LDS read and the (b, c, npoints, nsample) output leaves in 16-byte stores -- 0.35 ms -> see profiles/r04_dropin_ops.md for
    // C = 128, nsample = 32, b = 8 (the direct kernel below: one 4-byte gather per element through L2, 1.8 TB/s)
    if ((((uintptr_t)idx | (uintptr_t)out) & 15) == 0 && (slots & 3) == 0 && slots >= 2L * n && (long)n * 4 <= 128 * 1024) {
        hipStream_t st = (hipStream_t)stream;
        if ((long)n * 4 * 4 <= 64 * 1024) return launch_group_cat_lds<4, false>(b, n, npoints, c, nsample, nullptr, nullptr, points, idx, out, st);
        if ((long)n * 4 * 2 <= 64 * 1024) return launch_group_cat_lds<2, false>(b, n, npoints, c, nsample, nullptr, nullptr, points, idx, out, st);
        return launch_group_cat_lds<1, false>(b, n, npoints, c, nsample, nullptr, nullptr, points, idx, out, st);
    }
    dim3 grid(ceil_div(slots, 256), c, b);
    hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, slots, points, idx, out);
    return check_launch("group_points");
}

extern "C" int prcnn_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                       const float *grad_out, const int *idx, float *grad_points, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "group_points_grad: bad sizes");
    PRCNN_REQUIRE(b <= 65535 && c <= 65535, "group_points_grad: b/c > 65535");
    const long slots = (long)npoints * nsample;
    if (b == 0 || c == 0 || slots == 0) return PRCNN_OK;
    PRCNN_REQUIRE(grad_out && idx && grad_points, "group_points_grad: null pointer");
    dim3 grid(ceil_div(slots, 256), c, b);
    hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, slots, grad_out, idx, grad_points);
    return check_launch("group_points_grad");
}

extern "C" int prcnn_gather_points(int b, int c, int n, int npoints,
                                   const float *points, const int *idx, float *out, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "gather_points: bad sizes");
    PRCNN_REQUIRE(b <= 65535 && c <= 65535, "gather_points: b/c > 65535");
    if (b == 0 || c == 0 || npoints == 0) return PRCNN_OK;
    PRCNN_REQUIRE(points && idx && out, "gather_points: null pointer");
    dim3 grid(ceil_div(npoints, 256), c, b);
    hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, npoints, points, idx, out);
    return check_launch("gather_points");
}

extern "C" int prcnn_gather_points_grad(int b, int c, int n, int npoints,
                                        const float *grad_out, const int *idx, float *grad_points, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "gather_points_grad: bad sizes");
    PRCNN_REQUIRE(b <= 65535 && c <= 65535, "gather_points_grad: b/c > 65535");
    if (b == 0 || c == 0 || npoints == 0) return PRCNN_OK;
    PRCNN_REQUIRE(grad_out && idx && grad_points, "gather_points_grad: null pointer");
    dim3 grid(ceil_div(npoints, 256), c, b);
    hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, npoints, grad_out, idx, grad_points);
    return check_launch("gather_points_grad");
}

extern "C" int prcnn_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                                     const float *new_xyz, const float *xyz, const float *features,
                                     int *idx, float *out, void *stream)
{
    PRCNN_REQUIRE(c >= 0 && (c == 0 || features), "query_and_group: features missing for c=%d", c);
    PRCNN_REQUIRE(idx && out, "query_and_group: idx/out must be provided");
    PRCNN_REQUIRE(n > 0 || m == 0, "query_and_group: empty cloud");
    int rc = ball_query_dispatch(b, n, m, radius, nsample, new_xyz, xyz, idx, 1, (hipStream_t)stream);
    if (rc != PRCNN_OK || b == 0 || m == 0) return rc;
    // LDS row staging pays when every staged row is reused by many slots and fits in LDS
    const long slots = (long)m * nsample;
    const bool aligned = (((uintptr_t)idx | (uintptr_t)out) & 15) == 0 && (slots & 3) == 0;
    if (aligned && c > 0 && b <= 65535 && slots >= 4L * n && (long)n * 4 <= 128 * 1024) {
        hipStream_t st = (hipStream_t)stream;
        int force = 0;
        if (const char *e = getenv("PRCNN_GROUP_ROWS")) force = atoi(e);                     // tuning knob
        if (force == 1) return launch_group_cat_lds<1>(b, n, m, c, nsample, new_xyz, xyz, features, idx, out, st);
        if (force == 2 && (long)n * 4 * 2 <= 144 * 1024) return launch_group_cat_lds<2>(b, n, m, c, nsample, new_xyz, xyz, features, idx, out, st);
        // <= 64 KiB of staged rows per block keeps two blocks per CU: one stages while the other streams
        if ((long)n * 4 * 4 <= 64 * 1024) return launch_group_cat_lds<4>(b, n, m, c, nsample, new_xyz, xyz, features, idx, out, st);
        if ((long)n * 4 * 2 <= 64 * 1024) return launch_group_cat_lds<2>(b, n, m, c, nsample, new_xyz, xyz, features, idx, out, st);
        return launch_group_cat_lds<1>(b, n, m, c, nsample, new_xyz, xyz, features, idx, out, st);
    }
    dim3 grid(ceil_div((long)m * nsample, 256), b);
    hipLaunchKernelGGL(group_cat_kernel, grid, dim3(256), 0, (hipStream_t)stream, n, m, c, nsample,
                       new_xyz, xyz, features, idx, out);
    return check_launch("query_and_group");
}
