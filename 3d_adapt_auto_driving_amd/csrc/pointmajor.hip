// pointmajor.hip -- point-major ("channels-last") forms of grouping, pooling and interpolation
// for gfx950: the layout the MI355X inference path keeps its activations in.
//
// The reference keeps features channel-major (B, C, N) and gathers 4-byte elements out of each
// channel row (group_points_gpu.cu:47-66, interpolate_gpu.cu:77-97): every lane of a wave touches
// a different cache line.  With features stored point-major (B, N, C) a neighbour is ONE
// contiguous C-vector, so the gather becomes 16-byte loads on consecutive addresses, the grouped
// tensor (B*M*ns rows x K columns) is written with consecutive 16-byte stores, it is directly the
// A operand of the per-group MLP GEMM (rows x K) @ (K x Cout) with the bias+ReLU epilogue fused
// into the GEMM, and the max over nsample reduces ns consecutive rows.
//
// Row layout of the grouped tensor:  [ f_0 .. f_{C-1} , 0-pad to C4 , dx, dy, dz, 0 ]  with
// C4 = round_up(C, 4): every 16-byte chunk is aligned; the MLP's first-layer weight is permuted
// (and zero-padded) to this column order once, on the host.  Values are the reference's
// (xyz[idx] - centre, features[idx]) -- only their position in the row differs.
#include "common.hpp"

namespace prcnn {

// one thread per 16-byte chunk of the grouped tensor
__global__ __launch_bounds__(256) void group_cat_pm_kernel(
    int n, int m, int c, int nsample, int chunks /* C4/4 + 1 */, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, const float *__restrict__ feat /* (b,n,c) */,
    const int *__restrict__ idx, float4 *__restrict__ out)
{
    const int b = blockIdx.y;
    const long slots = (long)m * nsample;
    const long total = slots * chunks;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long s = e / chunks;
        const int j = (int)(e - s * chunks);
        const int k = idx[(long)b * slots + s];
        float4 v;
        if (j == chunks - 1) {
            const int p = (int)(s / nsample);
            const float *pt = xyz + ((long)b * n + k) * 3;
            const float *ct = new_xyz + ((long)b * m + p) * 3;
            v = make_float4(pt[0] - ct[0], pt[1] - ct[1], pt[2] - ct[2], 0.f);
        } else {
            const float *f = feat + ((long)b * n + k) * c + 4 * j;
            if (4 * j + 4 <= c && (c & 3) == 0) {
                v = *reinterpret_cast<const float4 *>(f);
            } else {
                v.x = 4 * j + 0 < c ? f[0] : 0.f;
                v.y = 4 * j + 1 < c ? f[1] : 0.f;
                v.z = 4 * j + 2 < c ? f[2] : 0.f;
                v.w = 4 * j + 3 < c ? f[3] : 0.f;
            }
        }
        out[((long)b * slots) * chunks + e] = v;
    }
}

// First shared-MLP layer without materialising the grouped input.  The layer is linear before its
// ReLU, so  W1 [f_k ; x_k - c_p] + b1 = (W1f f_k + b1) + W1x (x_k - c_p):  the feature part
// P = F W1f^T + b1 is ONE GEMM over the N points of the cloud instead of over the M*ns grouped
// rows (16x fewer rows at M*ns = 8192, N = 512), and this kernel forms the grouped rows of the
// layer-1 OUTPUT directly:   out[slot][ch] = relu(P[idx[slot]][ch] + wx[ch]*dx + wy[ch]*dy + wz[ch]*dz)
// (the coordinate part is evaluated on the differences, so nothing cancels).  Thread per 16-byte
// chunk of the output row; P rows are contiguous C-vectors.
__global__ __launch_bounds__(256) void gather_affine_relu_pm_kernel(
    int n, int m, int c4 /* cout/4 */, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, const float4 *__restrict__ P /* (b,n,cout) */,
    const float4 *__restrict__ wxyz /* (3, cout) */, const int *__restrict__ idx, float4 *__restrict__ out)
{
    const int b = blockIdx.y;
    const long slots = (long)m * nsample;
    const long total = slots * c4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long s = e / c4;
        const int j = (int)(e - s * c4);
        const int k = idx[(long)b * slots + s];
        const int p = (int)(s / nsample);
        const float *pt = xyz + ((long)b * n + k) * 3;
        const float *ct = new_xyz + ((long)b * m + p) * 3;
        const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];
        const float4 base = P[((long)b * n + k) * c4 + j];
        const float4 wx = wxyz[j], wy = wxyz[c4 + j], wz = wxyz[2 * c4 + j];
        float4 v;
        v.x = fmaxf(base.x + wx.x * dx + wy.x * dy + wz.x * dz, 0.f);
        v.y = fmaxf(base.y + wx.y * dx + wy.y * dy + wz.y * dz, 0.f);
        v.z = fmaxf(base.z + wx.z * dx + wy.z * dy + wz.z * dz, 0.f);
        v.w = fmaxf(base.w + wx.w * dx + wy.w * dy + wz.w * dz, 0.f);
        out[((long)b * slots) * c4 + e] = v;
    }
}

// out[r][ch] = max_s in[(r*ns + s)][ch]; rows of `in` have `c` floats, rows of `out` have
// `out_stride` floats and the result goes to columns [out_col, out_col + c)
__global__ __launch_bounds__(256) void maxpool_pm_kernel(long rows_out, int ns, int c4 /* c/4 */,
                                                         const float4 *__restrict__ in, float *__restrict__ out,
                                                         int out_stride, int out_col)
{
    const long total = rows_out * c4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / c4;
        const int j = (int)(e - r * c4);
        const float4 *src = in + (r * ns) * c4 + j;
        float4 m = src[0];
        for (int s = 1; s < ns; ++s) {
            const float4 v = src[(long)s * c4];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        float *dst = out + r * out_stride + out_col + 4 * j;
        dst[0] = m.x; dst[1] = m.y; dst[2] = m.z; dst[3] = m.w;
    }
}

// out[b][p][out_col + ch] = w0*f[i0][ch] + w1*f[i1][ch] + w2*f[i2][ch]   (left to right, no fma)
__global__ __launch_bounds__(256) void three_interpolate_pm_kernel(
    int c4, int m, int n, const float4 *__restrict__ feat /* (b,m,c) */, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out, int out_stride, int out_col)
{
    const int b = blockIdx.y;
    const long total = (long)n * c4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int p = (int)(e / c4);
        const int j = (int)(e - (long)p * c4);
        const int *ix = idx + ((long)b * n + p) * 3;
        const float *w = weight + ((long)b * n + p) * 3;
        const float4 f0 = feat[((long)b * m + ix[0]) * c4 + j];
        const float4 f1 = feat[((long)b * m + ix[1]) * c4 + j];
        const float4 f2 = feat[((long)b * m + ix[2]) * c4 + j];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        float *dst = out + ((long)b * n + p) * out_stride + out_col + 4 * j;
        dst[0] = __fadd_rn(__fadd_rn(__fmul_rn(w0, f0.x), __fmul_rn(w1, f1.x)), __fmul_rn(w2, f2.x));
        dst[1] = __fadd_rn(__fadd_rn(__fmul_rn(w0, f0.y), __fmul_rn(w1, f1.y)), __fmul_rn(w2, f2.y));
        dst[2] = __fadd_rn(__fadd_rn(__fmul_rn(w0, f0.z), __fmul_rn(w1, f1.z)), __fmul_rn(w2, f2.z));
        dst[3] = __fadd_rn(__fadd_rn(__fmul_rn(w0, f0.w), __fmul_rn(w1, f1.w)), __fmul_rn(w2, f2.w));
    }
}

// FP module input in ONE pass (pointnet2_modules.py:139-151: interpolate, then torch.cat with the skip features):
// out[b][p] = [ w0 f[i0] + w1 f[i1] + w2 f[i2]  (c4 float4s, left to right, no fma) | skip[b][p] (s4 float4s) ]
__global__ __launch_bounds__(256) void three_interpolate_cat_pm_kernel(
    int c4, int s4, int m, int n, const float4 *__restrict__ feat /* (b,m,c) */, const int *__restrict__ idx,
    const float *__restrict__ weight, const float4 *__restrict__ skip /* (b,n,s) */, float4 *__restrict__ out /* (b,n,c+s) */)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y;
    const int row4 = c4 + s4;
    const long total = (long)n * row4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int p = (int)(e / row4);
        const int j = (int)(e - (long)p * row4);
        const long pt = (long)b * n + p;
        if (j >= c4) {
            out[pt * row4 + j] = skip[pt * s4 + (j - c4)];
            continue;
        }
        const int *ix = idx + pt * 3;
        const float *w = weight + pt * 3;
        const f32x4 *f = reinterpret_cast<const f32x4 *>(feat);
        const f32x4 f0 = f[((long)b * m + ix[0]) * c4 + j];
        const f32x4 f1 = f[((long)b * m + ix[1]) * c4 + j];
        const f32x4 f2 = f[((long)b * m + ix[2]) * c4 + j];
        // one rounding per operation (the file is compiled with -ffp-contract=off): v_pk_mul_f32 / v_pk_add_f32
        const f32x4 v = (w[0] * f0 + w[1] * f1) + w[2] * f2;
        reinterpret_cast<f32x4 *>(out)[pt * row4 + j] = v;
    }
}

static int grid_cap(long items)
{
    long g = (items + 255) / 256;
    const long cap = 256L * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace prcnn

using namespace prcnn;

// out (b, m*nsample, kpad) with kpad = round_up(c,4) + 4; features (b, n, c) point-major or NULL
extern "C" int prcnn_group_cat_pm(int b, int n, int m, int c, int nsample, const float *new_xyz,
                                  const float *xyz, const float *features, const int *idx, float *out,
                                  void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0 && nsample >= 0, "group_cat_pm: bad sizes");
    PRCNN_REQUIRE(b <= 65535, "group_cat_pm: batch > 65535");
    if (b == 0 || m == 0 || nsample == 0) return PRCNN_OK;
    PRCNN_REQUIRE(new_xyz && xyz && idx && out && (features || c == 0), "group_cat_pm: null pointer");
    PRCNN_REQUIRE(((uintptr_t)out & 15) == 0 && (c == 0 || ((uintptr_t)features & 15) == 0), "group_cat_pm: 16-byte alignment required");
    const int chunks = (c + 3) / 4 + 1;
    dim3 grid(grid_cap((long)m * nsample * chunks), b);
    hipLaunchKernelGGL(group_cat_pm_kernel, grid, dim3(256), 0, (hipStream_t)stream, n, m, c, nsample, chunks,
                       new_xyz, xyz, features, idx, (float4 *)out);
    return check_launch("group_cat_pm");
}

// P (b, n, cout) = per-point part of the first layer (bias included), wxyz (3, cout) = its xyz columns
// -> out (b, m*nsample, cout) = relu(P[idx] + wxyz . (xyz[idx] - centre)); cout % 4 == 0
extern "C" int prcnn_gather_affine_relu_pm(int b, int n, int m, int cout, int nsample, const float *new_xyz,
                                           const float *xyz, const float *P, const float *wxyz, const int *idx,
                                           float *out, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && cout >= 0 && nsample >= 0, "gather_affine_relu_pm: bad sizes");
    PRCNN_REQUIRE((cout & 3) == 0 && b <= 65535, "gather_affine_relu_pm: cout %% 4 != 0 or batch too large");
    if (b == 0 || m == 0 || nsample == 0 || cout == 0) return PRCNN_OK;
    PRCNN_REQUIRE(new_xyz && xyz && P && wxyz && idx && out, "gather_affine_relu_pm: null pointer");
    PRCNN_REQUIRE((((uintptr_t)P | (uintptr_t)wxyz | (uintptr_t)out) & 15) == 0, "gather_affine_relu_pm: 16-byte alignment required");
    dim3 grid(grid_cap((long)m * nsample * (cout / 4)), b);
    hipLaunchKernelGGL(gather_affine_relu_pm_kernel, grid, dim3(256), 0, (hipStream_t)stream, n, m, cout / 4, nsample,
                       new_xyz, xyz, (const float4 *)P, (const float4 *)wxyz, idx, (float4 *)out);
    return check_launch("gather_affine_relu_pm");
}

// in (rows_out*ns, c) -> out[r][out_col .. out_col+c) with row stride out_stride; c % 4 == 0
extern "C" int prcnn_maxpool_pm(long rows_out, int ns, int c, const float *in, float *out, int out_stride,
                                int out_col, void *stream)
{
    PRCNN_REQUIRE(rows_out >= 0 && ns > 0 && c >= 0 && out_stride >= c + out_col && out_col >= 0, "maxpool_pm: bad sizes");
    PRCNN_REQUIRE((c & 3) == 0, "maxpool_pm: channel count %d not a multiple of 4", c);
    if (rows_out == 0 || c == 0) return PRCNN_OK;
    PRCNN_REQUIRE(in && out && ((uintptr_t)in & 15) == 0, "maxpool_pm: null or misaligned pointer");
    hipLaunchKernelGGL(maxpool_pm_kernel, dim3(grid_cap(rows_out * (c / 4))), dim3(256), 0, (hipStream_t)stream,
                       rows_out, ns, c / 4, (const float4 *)in, out, out_stride, out_col);
    return check_launch("maxpool_pm");
}

// features (b, m, c) point-major, idx/weight (b, n, 3) -> out (b, n, out_stride)[..., out_col:out_col+c]
extern "C" int prcnn_three_interpolate_pm(int b, int c, int m, int n, const float *features, const int *idx,
                                          const float *weight, float *out, int out_stride, int out_col,
                                          void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0 && out_stride >= c + out_col && out_col >= 0, "three_interpolate_pm: bad sizes");
    PRCNN_REQUIRE((c & 3) == 0 && b <= 65535, "three_interpolate_pm: c %% 4 != 0 or batch too large");
    if (b == 0 || c == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(features && idx && weight && out && ((uintptr_t)features & 15) == 0, "three_interpolate_pm: null or misaligned pointer");
    dim3 grid(grid_cap((long)n * (c / 4)), b);
    hipLaunchKernelGGL(three_interpolate_pm_kernel, grid, dim3(256), 0, (hipStream_t)stream, c / 4, m, n,
                       (const float4 *)features, idx, weight, out, out_stride, out_col);
    return check_launch("three_interpolate_pm");
}

// features (b, m, c) point-major, idx/weight (b, n, 3), skip (b, n, c_skip) -> out (b, n, c + c_skip) = [interpolated | skip]
extern "C" int prcnn_three_interpolate_cat_pm(int b, int c, int m, int n, const float *features, const int *idx,
                                              const float *weight, const float *skip, int c_skip, float *out, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0 && c_skip >= 0, "three_interpolate_cat_pm: bad sizes");
    PRCNN_REQUIRE((c & 3) == 0 && (c_skip & 3) == 0 && b <= 65535, "three_interpolate_cat_pm: widths must be multiples of 4, batch <= 65535");
    if (b == 0 || n == 0 || c + c_skip == 0) return PRCNN_OK;
    PRCNN_REQUIRE(features && idx && weight && out && (skip || c_skip == 0), "three_interpolate_cat_pm: null pointer");
    PRCNN_REQUIRE((((uintptr_t)features | (uintptr_t)out | (uintptr_t)skip) & 15) == 0, "three_interpolate_cat_pm: 16-byte alignment required");
    dim3 grid(grid_cap((long)n * ((c + c_skip) / 4)), b);
    hipLaunchKernelGGL(three_interpolate_cat_pm_kernel, grid, dim3(256), 0, (hipStream_t)stream, c / 4, c_skip / 4, m, n,
                       (const float4 *)features, idx, weight, (const float4 *)skip, (float4 *)out);
    return check_launch("three_interpolate_cat_pm");
}
