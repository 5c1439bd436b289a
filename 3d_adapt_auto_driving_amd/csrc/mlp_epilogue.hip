// mlp_epilogue.hip -- epilogues of the per-group shared MLP (pointnet2_modules.py:37-53:
// [Conv2d 1x1 (+BN eval) + ReLU] x k, then max over nsample) for gfx950.
//
// The 1x1 convolutions are dense f32 GEMMs (they run on MFMA through rocBLAS/Tensile at
// ~145 TFLOP/s here); what the reference leaves on the table is the elementwise traffic around
// them: bias add, ReLU and the max-pool each re-stream the (B, C, npoint, nsample) activation.
//   * bias_relu_inplace: one read + one write instead of two of each;
//   * maxpool_bias_relu: the LAST layer's bias+ReLU commute with the max over nsample
//     (x -> relu(x + b) is monotone non-decreasing and rounding is monotone, so
//     max_s relu(fl(x_s + b)) == relu(fl(max_s x_s + b)) exactly): the raw GEMM output is read
//     ONCE, reduced over nsample with DPP/permute lane shuffles, and only (B, C, npoint) is
//     written.  HBM-bound; 16-byte loads, consecutive lanes on consecutive addresses.
#include "common.hpp"

namespace prcnn {

// x (outer, C, inner) in place: x = max(x + bias[c], 0); inner % 4 == 0 path is vectorised
__global__ __launch_bounds__(256) void bias_relu_vec4_kernel(long total4, int c, long inner4,
                                                             const float *__restrict__ bias, float4 *__restrict__ x)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int ch = (int)((i / inner4) % c);
        const float b = bias[ch];
        float4 v = x[i];
        v.x = fmaxf(v.x + b, 0.f); v.y = fmaxf(v.y + b, 0.f); v.z = fmaxf(v.z + b, 0.f); v.w = fmaxf(v.w + b, 0.f);
        x[i] = v;
    }
}

__global__ __launch_bounds__(256) void bias_relu_scalar_kernel(long total, int c, long inner,
                                                               const float *__restrict__ bias, float *__restrict__ x)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)((i / inner) % c);
        x[i] = fmaxf(x[i] + bias[ch], 0.f);
    }
}

// in (rows, ns) with row = (b*C + c)*npoint + p  ->  out[row] = relu(max_s in[row][s] + bias[c])
// LANES = ns/4 lanes share a row (each holds a float4), power of two <= 64.
template <int LANES>
__global__ __launch_bounds__(256) void maxpool_bias_relu_kernel(long rows, int c, int npoint,
                                                                const float *__restrict__ bias,
                                                                const float4 *__restrict__ in, float *__restrict__ out)
{
    constexpr int ROWS_PER_BLOCK = 256 / LANES;
    const int sub = threadIdx.x % LANES;
    for (long r0 = (long)blockIdx.x * ROWS_PER_BLOCK; r0 < rows; r0 += (long)gridDim.x * ROWS_PER_BLOCK) {
        const long row = r0 + threadIdx.x / LANES;
        float m = -INFINITY;
        if (row < rows) {
            const float4 v = in[row * LANES + sub];
            m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        }
#pragma unroll
        for (int s = 1; s < LANES; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
        if (sub == 0 && row < rows) {
            const int ch = (int)((row / npoint) % c);
            out[row] = fmaxf(m + bias[ch], 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void maxpool_bias_relu_generic_kernel(long rows, int c, int npoint, int ns,
                                                                        const float *__restrict__ bias,
                                                                        const float *__restrict__ in, float *__restrict__ out)
{
    for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long)gridDim.x * 256) {
        float m = -INFINITY;
        for (int s = 0; s < ns; ++s) m = fmaxf(m, in[row * ns + s]);
        out[row] = fmaxf(m + bias[(row / npoint) % c], 0.f);
    }
}

static int grid_for(long work_items)
{
    long g = (work_items + 255) / 256;
    const long cap = 256L * 16;  // 256 CUs x 16 resident 256-thread blocks, grid-stride beyond
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace prcnn

using namespace prcnn;

// x (outer, c, inner) f32, in place.  Not part of the reference ABI (it fuses what
// pytorch_utils.py's Conv2d -> (BN) -> ReLU chain does in separate passes).
extern "C" int prcnn_bias_relu_inplace(long outer, int c, long inner, const float *bias, float *x, void *stream)
{
    PRCNN_REQUIRE(outer >= 0 && c >= 0 && inner >= 0, "bias_relu_inplace: bad sizes");
    const long total = outer * c * inner;
    if (total == 0) return PRCNN_OK;
    PRCNN_REQUIRE(bias && x, "bias_relu_inplace: null pointer");
    if (inner % 4 == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(bias_relu_vec4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream,
                           total / 4, c, inner / 4, bias, (float4 *)x);
    else
        hipLaunchKernelGGL(bias_relu_scalar_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           total, c, inner, bias, x);
    return check_launch("bias_relu_inplace");
}

// in (b, c, npoint, nsample) raw last-layer GEMM output -> out (b, c, npoint)
extern "C" int prcnn_maxpool_bias_relu(int b, int c, int npoint, int nsample, const float *bias,
                                       const float *in, float *out, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && c >= 0 && npoint >= 0 && nsample > 0, "maxpool_bias_relu: bad sizes");
    const long rows = (long)b * c * npoint;
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(bias && in && out, "maxpool_bias_relu: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (nsample % 4 == 0) && ((uintptr_t)in & 15) == 0;
    const int lanes = nsample / 4;
#define LAUNCH(L) hipLaunchKernelGGL(maxpool_bias_relu_kernel<L>, dim3(grid_for(rows * L)), dim3(256), 0, st, \
                                     rows, c, npoint, bias, (const float4 *)in, out)
    if (vec && lanes == 1) LAUNCH(1);
    else if (vec && lanes == 2) LAUNCH(2);
    else if (vec && lanes == 4) LAUNCH(4);
    else if (vec && lanes == 8) LAUNCH(8);
    else if (vec && lanes == 16) LAUNCH(16);
    else if (vec && lanes == 32) LAUNCH(32);
    else if (vec && lanes == 64) LAUNCH(64);
    else
        hipLaunchKernelGGL(maxpool_bias_relu_generic_kernel, dim3(grid_for(rows)), dim3(256), 0, st, rows, c,
                           npoint, nsample, bias, in, out);
#undef LAUNCH
    return check_launch("maxpool_bias_relu");
}
