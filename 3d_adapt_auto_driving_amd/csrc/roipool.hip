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

namespace prcnn {

constexpr int RP_THREADS = 256;
constexpr int RP_MAX_S = 2048;

__global__ __launch_bounds__(RP_THREADS) void roipool3d_kernel(
    int pts_num, int boxes_num, int feat_len, int sampled, const float *__restrict__ xyz,
    const float *__restrict__ boxes3d, const float *__restrict__ pts_feature,
    float *__restrict__ pooled, int *__restrict__ empty_flag)
{
    __shared__ int s_sel[RP_MAX_S];
    __shared__ int s_wcnt[RP_THREADS / 64];

    const int box = blockIdx.x, b = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *bx = boxes3d + ((long)b * boxes_num + box) * 7;
    // pt_in_box3d :14-28; the trig pair and cy depend on the box only
    const float cx = bx[0], cz = bx[2], h = bx[3], w = bx[4], l = bx[5];
    const float cy = (float)((double)bx[1] - (double)h / 2.0);
    const float cosa = cos_f32(bx[6]), sina = sin_f32(bx[6]);
    const float hh = h * 0.5f, hl = l * 0.5f, hw = w * 0.5f;  // exact halves
    const float *__restrict__ pts = xyz + (long)b * pts_num * 3;

    __syncthreads();

    int total = 0;
    for (int base = 0; base < pts_num && total < sampled; base += RP_THREADS) {
        const int k = base + t;
        bool in = false;
        if (k < pts_num) {
            const float x = pts[3 * k], y = pts[3 * k + 1], z = pts[3 * k + 2];
            if (!(fabsf(x - cx) > 10.0f || fabsf(y - cy) > hh || fabsf(z - cz) > 10.0f)) {
                const float xr = __fadd_rn(__fmul_rn(x - cx, cosa), __fmul_rn(z - cz, -sina));
                const float zr = __fadd_rn(__fmul_rn(x - cx, sina), __fmul_rn(z - cz, cosa));
                in = (xr >= -hl) & (xr <= hl) & (zr >= -hw) & (zr <= hw);
            }
        }
        const unsigned long long mask = __ballot(in);
        if (lane == 0) s_wcnt[wave] = __popcll(mask);
        __syncthreads();
        int before = total;
#pragma unroll
        for (int q = 0; q < RP_THREADS / 64; ++q) {
            const int cq = s_wcnt[q];
            if (q < wave) before += cq;
            total += cq;
        }
        if (in) {
            const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < sampled) s_sel[pos] = k;
        }
        __syncthreads();
    }
    const int cnt = min(total, sampled);
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
    hipLaunchKernelGGL(roipool3d_kernel, grid, dim3(RP_THREADS), 0, (hipStream_t)stream, pts_num, boxes_num,
                       feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,
                       pooled_empty_flag);
    return check_launch("roipool3d");
}
