// interp.hip -- three_nn (K7), three_interpolate (K8) and its gradient (K9) for gfx950.
//
// Reference behaviour restated: pointnet2_lib/pointnet2/src/interpolate_gpu.cu:9-52, :77-97,
// :120-142.  three_nn keeps the three smallest squared distances with strict '<' (lowest index
// wins ties).  The reference compares an f32 candidate against double bests initialised to
// 1e40; every stored best is an f32 value, so f32 comparisons against +inf are equivalent and
// an untouched slot reads back as (float)1e40 = +inf either way.
//
// Design: one lane per unknown point; the known point is wave-uniform and comes in through
// scalar loads, so the inner loop is pure VALU (8 distance ops + the 3-deep insertion).
#include "common.hpp"
#include <math.h>
#include <stdlib.h>

namespace prcnn {

__global__ __launch_bounds__(256) void three_nn_kernel(
    int n, int m, const float *__restrict__ unknown, const float *__restrict__ known,
    float *__restrict__ dist2, int *__restrict__ idx, float *__restrict__ weight)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < n;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (valid) {
        const float *u = unknown + ((long)b * n + p) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    const float *__restrict__ kn = known + (long)b * m * 3;
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
#pragma unroll 4
    for (int k = 0; k < m; ++k) {
        const float d = sqdist3(ux, uy, uz, kn[3 * k], kn[3 * k + 1], kn[3 * k + 2]);
        // the reference's insertion (interpolate_gpu.cu:36-44: `if (d < best1) ... else if (d < best2) ... else if (d < best3)`) as selects:
        // the same strict comparisons, the same (distance, index) triples -- as branches the three arms diverge inside a wave and every
        // known point cost the sum of all three
        const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
        b3 = c2 ? b2 : (c3 ? d : b3); i3 = c2 ? i2 : (c3 ? k : i3);
        b2 = c1 ? b1 : (c2 ? d : b2); i2 = c1 ? i1 : (c2 ? k : i2);
        b1 = c1 ? d : b1;             i1 = c1 ? k : i1;
    }
    if (valid) {
        int *oi = idx + ((long)b * n + p) * 3;
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
        if (dist2) {
            float *od = dist2 + ((long)b * n + p) * 3;
            od[0] = b1; od[1] = b2; od[2] = b3;
        }
        if (weight) three_nn_weights(b1, b2, b3, weight + ((long)b * n + p) * 3);
    }
}

// out[b][c][p] = w0*f[i0] + w1*f[i1] + w2*f[i2]  (left to right, no fma)
__global__ __launch_bounds__(256) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int *ix = idx + ((long)b * n + p) * 3;
    const float *w = weight + ((long)b * n + p) * 3;
    const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    // blockIdx.y walks channel groups so that idx/weight are read once per 8 channels
    const int c0 = blockIdx.y * 8;
    const int c1 = min(c, c0 + 8);
    for (int ci = c0; ci < c1; ++ci) {
        const float *f = points + ((long)b * c + ci) * m;
        const float v = __fadd_rn(__fadd_rn(__fmul_rn(w0, f[i0]), __fmul_rn(w1, f[i1])), __fmul_rn(w2, f[i2]));
        out[((long)b * c + ci) * n + p] = v;
    }
}

// The same operator at HBM speed (round 4; the kernel above gathers 3 x 4 bytes per output element through L2: 1.0 TB/s on
// (8, 256, 4096) -> (8, 256, 16384)): the m-float rows of ROWS channels are staged in LDS, a thread owns 4 consecutive points --
// their 12 indices and 12 weights come in as six 16-byte loads, once for all ROWS channels -- and every channel leaves as one
// 16-byte store per thread.  Same operation order (w0 f0 + w1 f1, + w2 f2; no fma).  grid = (point chunks, channel groups, b).
template <int ROWS>
__global__ __launch_bounds__(1024) void three_interpolate_lds_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out)
{
    extern __shared__ float rows[];            // [ROWS][m]
    const int b = blockIdx.z, c0 = blockIdx.y * ROWS, t = threadIdx.x;
    const int nrows = min(ROWS, c - c0);
    {
        const float4 *src = reinterpret_cast<const float4 *>(points + ((long)b * c + c0) * m);   // consecutive channels are contiguous
        float4 *dst = reinterpret_cast<float4 *>(rows);
        const int quads = nrows * m / 4;
        for (int k = t; k < quads; k += 1024) dst[k] = src[k];
    }
    __syncthreads();
    const int quads = n >> 2;
    const int per = (quads + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * per, q1 = min(quads, q0 + per);
    const int4 *ix4 = reinterpret_cast<const int4 *>(idx + (long)b * n * 3);
    const float4 *w4 = reinterpret_cast<const float4 *>(weight + (long)b * n * 3);
    for (int q = q0 + t; q < q1; q += 1024) {
        const int4 ia = ix4[3 * q], ib = ix4[3 * q + 1], ic = ix4[3 * q + 2];          // points 4q .. 4q + 3: (i0 i1 i2)(i0 i1 i2)...
        const float4 wa = w4[3 * q], wb = w4[3 * q + 1], wc = w4[3 * q + 2];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (r >= nrows) break;
            const float *f = rows + r * m;
            float4 v;
            v.x = __fadd_rn(__fadd_rn(__fmul_rn(wa.x, f[ia.x]), __fmul_rn(wa.y, f[ia.y])), __fmul_rn(wa.z, f[ia.z]));
            v.y = __fadd_rn(__fadd_rn(__fmul_rn(wa.w, f[ia.w]), __fmul_rn(wb.x, f[ib.x])), __fmul_rn(wb.y, f[ib.y]));
            v.z = __fadd_rn(__fadd_rn(__fmul_rn(wb.z, f[ib.z]), __fmul_rn(wb.w, f[ib.w])), __fmul_rn(wc.x, f[ic.x]));
            v.w = __fadd_rn(__fadd_rn(__fmul_rn(wc.y, f[ic.y]), __fmul_rn(wc.z, f[ic.z])), __fmul_rn(wc.w, f[ic.w]));
            float4 *dst = reinterpret_cast<float4 *>(out + ((long)b * c + c0 + r) * n) + q;
            __builtin_nontemporal_store(v.x, &dst->x);
            __builtin_nontemporal_store(v.y, &dst->y);
            __builtin_nontemporal_store(v.z, &dst->z);
            __builtin_nontemporal_store(v.w, &dst->w);
        }
    }
}

template <int ROWS>
static int launch_three_interpolate_lds(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                        float *out, hipStream_t st)
{
    const size_t lds = (size_t)ROWS * m * sizeof(float);
    if (lds > 64 * 1024) {
        const int rc = ensure_dynamic_lds((const void *)three_interpolate_lds_kernel<ROWS>, lds, "three_interpolate");
        if (rc != PRCNN_OK) return rc;
    }
    const int groups = ceil_div(c, ROWS);
    int chunks = 1;                            // every chunk re-stages its rows: split the points only while the chip is not covered
    while ((long)b * groups * chunks < 512 && n / (chunks * 2) >= 8192) chunks *= 2;
    hipLaunchKernelGGL(three_interpolate_lds_kernel<ROWS>, dim3(chunks, groups, b), dim3(1024), lds, st, c, m, n, points, idx, weight, out);
    return check_launch("three_interpolate");
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points)
{
    const int b = blockIdx.z, ci = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int *ix = idx + ((long)b * n + p) * 3;
    const float *w = weight + ((long)b * n + p) * 3;
    const float g = grad_out[((long)b * c + ci) * n + p];
    float *dst = grad_points + ((long)b * c + ci) * m;
    atomicAdd(dst + ix[0], g * w[0]);
    atomicAdd(dst + ix[1], g * w[1]);
    atomicAdd(dst + ix[2], g * w[2]);
}

}  // namespace prcnn

namespace prcnn {
int three_nn_grid(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                  hipStream_t st, int *used, float *weight);   // three_nn_grid.hip
}
using namespace prcnn;

static int three_nn_any(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, float *weight,
                        void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "three_nn: bad sizes");
    PRCNN_REQUIRE(b <= 65535, "three_nn: batch > 65535");
    if (b == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(unknown && (dist2 || weight) && idx && (known || m == 0), "three_nn: null pointer");
    const bool brute_only = false;       // (round 6: switch PRCNN_THREE_NN_BRUTE removed; small levels take the scan by shape)
    if (!brute_only) {
        int used = 0;
        const int rc = three_nn_grid(b, n, m, unknown, known, dist2, idx, (hipStream_t)stream, &used, weight);
        if (rc != PRCNN_OK || used) return rc;
    }
    dim3 grid(ceil_div(n, 256), b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream, n, m, unknown, known, dist2, idx, weight);
    return check_launch("three_nn");
}

extern "C" int prcnn_three_nn(int b, int n, int m, const float *unknown, const float *known,
                              float *dist2, int *idx, void *stream)
{
    return three_nn_any(b, n, m, unknown, known, dist2, idx, nullptr, stream);
}

// three_nn + the inverse-distance weights PointnetFPModule.forward derives from it (pointnet2_modules.py:139-144) in the same
// kernel: idx (b,n,3) i32 and weight (b,n,3) f32 = r_k / (r_0 + r_1 + r_2), r_k = 1 / (sqrt(dist2_k) + 1e-8).
extern "C" int prcnn_three_nn_weights(int b, int n, int m, const float *unknown, const float *known,
                                      int *idx, float *weight, void *stream)
{
    PRCNN_REQUIRE(weight, "three_nn_weights: null pointer");
    return three_nn_any(b, n, m, unknown, known, nullptr, idx, weight, stream);
}

extern "C" int prcnn_three_interpolate(int b, int c, int m, int n, const float *points,
                                       const int *idx, const float *weight, float *out, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, "three_interpolate: bad sizes");
    PRCNN_REQUIRE(b <= 65535 && c <= 65535 * 8, "three_interpolate: b/c too large");
    if (b == 0 || c == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
    if ((((uintptr_t)points | (uintptr_t)idx | (uintptr_t)weight | (uintptr_t)out) & 15) == 0 && (n & 3) == 0 && (m & 3) == 0 &&
        n >= 2 * m && b <= 65535 && (long)m * 4 * 4 <= 128 * 1024) {
        // rows staged in LDS; 8 channels per workgroup while that keeps two workgroups per CU, else 4
        if ((long)m * 4 * 8 <= 64 * 1024) return launch_three_interpolate_lds<8>(b, c, m, n, points, idx, weight, out, (hipStream_t)stream);
        return launch_three_interpolate_lds<4>(b, c, m, n, points, idx, weight, out, (hipStream_t)stream);
    }
    dim3 grid(ceil_div(n, 256), ceil_div(c, 8), b);
    hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, m, n, points, idx, weight, out);
    return check_launch("three_interpolate");
}

extern "C" int prcnn_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                            const int *idx, const float *weight, float *grad_points, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, "three_interpolate_grad: bad sizes");
    PRCNN_REQUIRE(b <= 65535 && c <= 65535, "three_interpolate_grad: b/c too large");
    if (b == 0 || c == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(grad_out && idx && weight && grad_points, "three_interpolate_grad: null pointer");
    dim3 grid(ceil_div(n, 256), c, b);
    hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    return check_launch("three_interpolate_grad");
}
