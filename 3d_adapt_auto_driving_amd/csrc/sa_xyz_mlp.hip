// sa_xyz_mlp.hip -- one whole set-abstraction scale on coordinates only (the first RPN SA level:
// QueryAndGroup with no input features -> 3-layer shared MLP -> max over nsample;
// pointnet2_modules.py:37-53 with pointnet2_utils.py:241-264) in ONE kernel.
//
// Why its own kernel: with C_in = 3 and widths of 16..64 the level is all activation traffic when it goes
// through GEMMs -- 1.5 M grouped rows x (32+32+64) floats written and read back (> 1 GB per batch of 8
// scenes) for 9 GFLOP of arithmetic.  Here a grouped row lives in ONE LANE from the gather to the max:
//   * lane = one (centre, sample) row; the ns rows of a centre are ns consecutive lanes of a wave;
//   * weights are wave-uniform: they come in through scalar loads and enter the FMAs as SGPR operands
//     (round 4: v_pk_fma_f32 v[2], s[2], v -- two output channels per instruction, see xyz_layer), so the inner loops are pure VALU
//     with no LDS and no vector loads;
//   * activations (C1 + C2 + C3 <= 160 values) stay in VGPRs, loops fully unrolled;
//   * max over nsample = xor-butterfly across the ns lanes of the centre; each lane then stores its share
//     of the C3 outputs.
// HBM traffic: idx (4 B/row), the gathered coordinates (L2-resident cloud), C3 floats per centre out.
// Bound: VALU f32 (2 * rows * (3 C1 + C1 C2 + C2 C3) flop) at the PACKED rate, 256 flop / clock / CU = the f32 MFMA's own peak (157 TFLOP/s):
// an MFMA version would gain nothing at f32 and pay a transpose through LDS between the layers; and VALU work overlaps with the MFMA
// kernels of the other streams on the same CU, MFMA work queues behind them.  The MLP arithmetic decides no index, so FMAs
// are used (the GEMM libraries it replaces do the same); results agree with the GEMM path to f32 rounding.
#include "common.hpp"

namespace prcnn {

// y[j] = fma(w[K-1][j], a[K-1], ... fma(w[0][j], a[0], b[j])) for j < N: the FMA chain over k of one row's layer, TWO output channels per
// v_pk_fma_f32 (the weights of a channel pair are an SGPR pair, the activation one VGPR used for both halves).  Each component sees exactly
// the scalar chain: same bits as N v_fmac_f32 per k, half the VALU instructions -- packed f32 is the full 256 flop / clock / CU of the
// chip's f32 rate (as much as the f32 MFMA), plain v_fmac_f32 half of it.
template <int K, int N>
__device__ __forceinline__ void xyz_layer(const float *__restrict__ w, const float *__restrict__ b, const float (&a)[K], float (&y)[N])
{
    static_assert(N % 2 == 0, "channel pairs");
    pk_f32x2 acc[N / 2];
#pragma unroll
    for (int j = 0; j < N / 2; ++j) acc[j] = (pk_f32x2){b[2 * j], b[2 * j + 1]};
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const pk_f32x2 s = {a[q], a[q]};
#pragma unroll
        for (int j = 0; j < N / 2; ++j)
            acc[j] = __builtin_elementwise_fma((pk_f32x2){w[q * N + 2 * j], w[q * N + 2 * j + 1]}, s, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < N / 2; ++j) { y[2 * j] = acc[j].x; y[2 * j + 1] = acc[j].y; }
}

template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(256) void sa_xyz_mlp_kernel(
    long rows, int n, int m, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int *__restrict__ idx, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, float *__restrict__ out, int out_stride, int out_col)
{
    static_assert(C3 % NS == 0, "every lane of a centre stores C3 / NS channels");
    const long row_raw = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = row_raw < rows ? row_raw : rows - 1;       // tail lanes recompute the last row, store nothing
    const long centre = row / NS;
    const int b = (int)(centre / m);
    const int k = idx[row];
    const float *p = xyz + ((long)b * n + k) * 3;
    const float *c = new_xyz + centre * 3;
    const float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];

    float a1[C1];
    {
        const float d3[3] = {dx, dy, dz};
        xyz_layer<3, C1>(w1, b1, d3, a1);
    }
#pragma unroll
    for (int j = 0; j < C1; ++j) a1[j] = fmaxf(a1[j], 0.f);

    float a2[C2];
    xyz_layer<C1, C2>(w2, b2, a1, a2);
#pragma unroll
    for (int j = 0; j < C2; ++j) a2[j] = fmaxf(a2[j], 0.f);

    float a3[C3];
    xyz_layer<C2, C3>(w3, b3, a2, a3);

    // ReLU commutes with max; reduce across the NS lanes of this centre
#pragma unroll
    for (int off = NS / 2; off >= 1; off >>= 1)
#pragma unroll
        for (int j = 0; j < C3; ++j) a3[j] = fmaxf(a3[j], __shfl_xor(a3[j], off));

    if (row_raw < rows) {
        constexpr int PER = C3 / NS;
        const int s = (int)(row % NS);
        float *o = out + centre * out_stride + out_col + s * PER;
#pragma unroll
        for (int j = 0; j < C3; ++j)                               // lane s keeps channels [s*PER, (s+1)*PER)
            if (j / PER == s) o[j - s * PER] = fmaxf(a3[j], 0.f);
    }
}

// ---- the same level over the DISTINCT rows only (prcnn_ball_pack lists; see sa_packed.hip).  lane = one packed row; the
// wave's 64 rows are one tile of the list.  The relu'd outputs go through LDS so that the pool runs with lane = channel:
// a serial pass over the tile's rows keeps a running max per channel and flushes it with ONE coalesced atomicMax per centre
// (outputs are >= 0; the output slice is zeroed by the caller of the kernel).  Per-row arithmetic is identical to
// sa_xyz_mlp_kernel, so the pooled result is bit-identical.
template <int C1, int C2, int C3>
__global__ __launch_bounds__(256) void sa_xyz_mlp_packed_kernel(
    int m, const unsigned int *__restrict__ hdr, const unsigned int *__restrict__ rowinfo, const float4 *__restrict__ rowdxyz,
    const int *__restrict__ tilecloud, const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
    const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3, float *__restrict__ out,
    int out_stride, int out_col)
{
    __shared__ float z[4][64 * (C3 + 1)];
    __shared__ int ctr[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * 4 + wv;                  // this wave's tile
    if (t >= (long)hdr[0]) return;                             // wave-uniform: no barrier below, LDS is private per wave
    const long row = t * 64 + lane;
    const float4 d = rowdxyz[row];
    const float dx = d.x, dy = d.y, dz = d.z;
    ctr[wv][lane] = tilecloud[t] * m + (int)(rowinfo[row] >> 16);

    float a1[C1];
    {
        const float d3[3] = {dx, dy, dz};
        xyz_layer<3, C1>(w1, b1, d3, a1);
    }
#pragma unroll
    for (int j = 0; j < C1; ++j) a1[j] = fmaxf(a1[j], 0.f);
    float a2[C2];
    xyz_layer<C1, C2>(w2, b2, a1, a2);
#pragma unroll
    for (int j = 0; j < C2; ++j) a2[j] = fmaxf(a2[j], 0.f);
    float a3[C3];
    xyz_layer<C2, C3>(w3, b3, a2, a3);
    float *zw = z[wv];
#pragma unroll
    for (int j = 0; j < C3; ++j) zw[lane * (C3 + 1) + j] = fmaxf(a3[j], 0.f);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // pool: lane -> channel (C3 = 64: all rows; C3 = 32: lane half hh takes rows [32 hh, 32 hh + 32))
    constexpr int ROWS_PER = C3 == 64 ? 64 : 32;
    const int ch = lane & (C3 - 1), hh = C3 == 64 ? 0 : (lane >> 5);
    const int *cc = ctr[wv];
    float cur = 0.f;
    for (int i = 0; i < ROWS_PER; ++i) {
        const int r = hh * ROWS_PER + i;
        cur = fmaxf(cur, zw[r * (C3 + 1) + ch]);
        const bool last = (i == ROWS_PER - 1) || (cc[r + 1] != cc[r]);
        if (last) {
            atomicMax(reinterpret_cast<int *>(out + (long)cc[r] * out_stride + out_col + ch), __float_as_int(cur));
            cur = 0.f;
        }
    }
}

}  // namespace prcnn

using namespace prcnn;

// the same level over a packed row list (prcnn_ball_pack of the level's index tensor): out[(b*m)][out_col .. +c3) is zeroed
// and receives the per-centre maxima.  max_tiles = b * ceil(m * nsample / 64).
extern "C" int prcnn_sa_xyz_mlp_packed(int b, int m, int c1, int c2, int c3, long max_tiles, const unsigned int *rowinfo,
                                       const float *rowdxyz, const int *tilecloud, const unsigned int *hdr, const float *w1,
                                       const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                                       float *out, int out_stride, int out_col, int out_is_zero, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && m >= 0 && max_tiles >= 0, "sa_xyz_mlp_packed: bad sizes");
    PRCNN_REQUIRE((c1 == 16 && c2 == 16 && c3 == 32) || (c1 == 32 && c2 == 32 && c3 == 64),
                  "sa_xyz_mlp_packed: unsupported shape c1=%d c2=%d c3=%d", c1, c2, c3);
    PRCNN_REQUIRE(out_stride >= out_col + c3 && out_col >= 0, "sa_xyz_mlp_packed: bad output slice");
    if ((long)b * m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rowinfo && rowdxyz && tilecloud && hdr && w1 && b1 && w2 && b2 && w3 && b3 && out, "sa_xyz_mlp_packed: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (!out_is_zero && hipMemset2DAsync(out + out_col, (size_t)out_stride * sizeof(float), 0, (size_t)c3 * sizeof(float), (size_t)b * m, st) != hipSuccess) {
        set_error("sa_xyz_mlp_packed: cannot zero the output slice");
        return PRCNN_ELAUNCH;
    }
    if (max_tiles == 0) return PRCNN_OK;
    const long grid = (max_tiles + 3) / 4;
    PRCNN_REQUIRE(grid <= 0x7fffffffL, "sa_xyz_mlp_packed: too many tiles");
    if (c3 == 32)
        hipLaunchKernelGGL((sa_xyz_mlp_packed_kernel<16, 16, 32>), dim3((unsigned)grid), dim3(256), 0, st, m, hdr, rowinfo,
                           (const float4 *)rowdxyz, tilecloud, w1, b1, w2, b2, w3, b3, out, out_stride, out_col);
    else
        hipLaunchKernelGGL((sa_xyz_mlp_packed_kernel<32, 32, 64>), dim3((unsigned)grid), dim3(256), 0, st, m, hdr, rowinfo,
                           (const float4 *)rowdxyz, tilecloud, w1, b1, w2, b2, w3, b3, out, out_stride, out_col);
    return check_launch("sa_xyz_mlp_packed");
}

// xyz (b,n,3), new_xyz (b,m,3), idx (b,m,nsample) -> out[(b*m rows)][out_col .. out_col + c3), row stride out_stride.
// w1 (>=3, c1) rows = x, y, z weights; w2 (c1, c2); w3 (c2, c3): k-major ("row = input channel"), BN folded.
// Every layer is followed by ReLU.  Supported: (c1,c2,c3,nsample) in {(16,16,32,16), (32,32,64,32)}.
extern "C" int prcnn_sa_xyz_mlp_supported(int c1, int c2, int c3, int nsample)
{
    return (c1 == 16 && c2 == 16 && c3 == 32 && nsample == 16) || (c1 == 32 && c2 == 32 && c3 == 64 && nsample == 32);
}

extern "C" int prcnn_sa_xyz_mlp(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *new_xyz,
                                const float *xyz, const int *idx, const float *w1, const float *b1, const float *w2,
                                const float *b2, const float *w3, const float *b3, float *out, int out_stride,
                                int out_col, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "sa_xyz_mlp: bad sizes");
    PRCNN_REQUIRE(prcnn_sa_xyz_mlp_supported(c1, c2, c3, nsample),
                  "sa_xyz_mlp: unsupported shape c1=%d c2=%d c3=%d nsample=%d", c1, c2, c3, nsample);
    PRCNN_REQUIRE(out_stride >= out_col + c3 && out_col >= 0, "sa_xyz_mlp: bad output slice");
    const long rows = (long)b * m * nsample;
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(n > 0, "sa_xyz_mlp: empty cloud");
    PRCNN_REQUIRE(new_xyz && xyz && idx && w1 && b1 && w2 && b2 && w3 && b3 && out, "sa_xyz_mlp: null pointer");
    const long grid = (rows + 255) / 256;
    PRCNN_REQUIRE(grid <= 0x7fffffffL, "sa_xyz_mlp: too many rows");
    hipStream_t st = (hipStream_t)stream;
    if (nsample == 16)
        hipLaunchKernelGGL((sa_xyz_mlp_kernel<16, 16, 32, 16>), dim3((unsigned)grid), dim3(256), 0, st, rows, n, m, xyz, new_xyz,
                           idx, w1, b1, w2, b2, w3, b3, out, out_stride, out_col);
    else
        hipLaunchKernelGGL((sa_xyz_mlp_kernel<32, 32, 64, 32>), dim3((unsigned)grid), dim3(256), 0, st, rows, n, m, xyz, new_xyz,
                           idx, w1, b1, w2, b2, w3, b3, out, out_stride, out_col);
    return check_launch("sa_xyz_mlp");
}
