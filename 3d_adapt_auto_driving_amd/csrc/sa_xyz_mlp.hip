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
    // where a new centre begins, as ONE word (round 5): the row values are independent loads of an unrolled loop and a boundary is a bit
    // test on a constant position (C3 = 64: a scalar branch; C3 = 32: the two lane halves walk their own 32 rows, the test is per lane).
    // (The loop used to read the row's value, then two entries of the centre list, compare and branch, row after row, not unrolled.)
    const int myc = cc[lane], prevc = cc[lane ? lane - 1 : 0];
    const unsigned long long start = __ballot(lane == 0 || myc != prevc);
    const unsigned long long sh = C3 == 64 ? start : (start >> (hh * ROWS_PER));
    float cur = 0.f;
#pragma unroll
    for (int i = 0; i < ROWS_PER; ++i) {
        const int r = hh * ROWS_PER + i;
        cur = fmax_raw(cur, zw[r * (C3 + 1) + ch]);
        const bool last = (i == ROWS_PER - 1) || ((sh >> (i + 1)) & 1ull) != 0;
        if (last) {
            atomicMax(reinterpret_cast<int *>(out + (long)cc[r] * out_stride + out_col + ch), __float_as_int(cur));
            cur = 0.f;
        }
    }
}


// ---- the wider scale (3 -> 32 -> 32 -> 64) with layers 2 and 3 on the matrix cores (round 5) -------------------------------------
// The VALU form above streams every weight through a scalar register: 213 s_load_dwordx16 per 64-row tile, 13.6 KB that each of
// a CU's waves asks the scalar cache for again -- at the packed-f32 rate that is most of the cache's bandwidth, and on LiDAR-shaped
// scenes (1.2 M rows in this scale) the kernel ran at 0.37 of that rate (150 us for 16 clouds; more resident waves made it SLOWER).
// Here the weights sit in registers as the MFMA's A operand and the product is taken transposed, out^T = W^T act^T:
//   A: lane (c = lane & 31, h = lane >> 5) holds W[2 s + h][32 cb + c]            (a channel block's weights of k-step s)
//   B: lane (j = lane & 31, h) holds act[row j of the 32-row block][2 s + h]
//   D: lane (j, h), register r holds out[row j][32 cb + (r & 3) + 8 (r >> 2) + 4 h] -- the ROW stays in its lane.
// B operands: layer 2's input has all 32 channels of the lane's own row (the VALU layer 1): X = act[2 s], Y = act[2 s + 1], and ONE
// v_permlane32_swap (X's upper lanes <-> Y's lower lanes) turns them into the operand of rows 0-31 and that of rows 32-63.  Layer 3's
// input is layer 2's D: channels 2 s and 2 s + 1 sit in the same lane half; swapping registers r(2 s), r(2 s + 1) gives the operand of
// step s in one result and that of step s + 2 (channels 2 s + 4, 2 s + 5, held by the other half) in the other: 8 swaps per block.
// v_mfma_f32_32x32x2_f32 is bitwise fma(A[i][0], B[0][j], .) then fma(A[i][1], B[1][j], .) (oracle/mlp_oracle.c), the accumulators
// start from the bias: the chain of xyz_layer, k ascending -- same bits as the VALU form (tests/test_gpu_packed.py, the shadow run).
// Persistent waves (the weights are loaded once), one 64-row tile at a time; pooling as above through a 64 x 68 LDS tile per wave.
typedef float xm_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void sa_xyz_mlp_packed_mfma_kernel(
    int m, const unsigned int *__restrict__ hdr, const unsigned int *__restrict__ rowinfo, const float4 *__restrict__ rowdxyz,
    const int *__restrict__ tilecloud, const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
    const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3, float *__restrict__ out,
    int out_stride, int out_col)
{
    constexpr int C1 = 32, C2 = 32, C3 = 64, LD = 68;
    __shared__ __align__(16) float z[4][64 * LD];
    __shared__ int ctr[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long tiles = (long)hdr[0];
    const long nw = (long)gridDim.x * 4;
    long t = (long)blockIdx.x * 4 + wv;
    if (t >= tiles) return;                                    // wave-uniform: no workgroup barrier below, LDS is private per wave
    float wa2[16], wa3[2][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        wa2[s] = w2[(2 * s + h) * C2 + c];
        wa3[0][s] = w3[(2 * s + h) * C3 + c];
        wa3[1][s] = w3[(2 * s + h) * C3 + 32 + c];
    }
    xm_f32x16 bias2, bias3[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch_r = (r & 3) + 8 * (r >> 2) + 4 * h;
        bias2[r] = b2[ch_r]; bias3[0][r] = b3[ch_r]; bias3[1][r] = b3[32 + ch_r];
    }
    float *zw = z[wv];
    int *cc = ctr[wv];
    for (; t < tiles; t += nw) {
        const long row = t * 64 + lane;
        const float4 d = rowdxyz[row];
        cc[lane] = tilecloud[t] * m + (int)(rowinfo[row] >> 16);
        float a1[C1];
        {
            const float d3[3] = {d.x, d.y, d.z};
            xyz_layer<3, C1>(w1, b1, d3, a1);
        }
#pragma unroll
        for (int j = 0; j < C1; ++j) a1[j] = fmaxf(a1[j], 0.f);
        // ---- layer 2: D2[rb] (32 channels x 32 rows) for the row blocks rb = 0 (lanes 0-31's rows), 1 (lanes 32-63's rows)
        xm_f32x16 d2[2] = {bias2, bias2};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(a1[2 * s]), __float_as_int(a1[2 * s + 1]), false, false);
            d2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[s], __int_as_float(sw[0]), d2[0], 0, 0, 0);
            d2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[s], __int_as_float(sw[1]), d2[1], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { d2[0][r] = fmaxf(d2[0][r], 0.f); d2[1][r] = fmaxf(d2[1][r], 0.f); }
        // ---- layer 3: D3[rb][cb]
        xm_f32x16 d3[2][2] = {{bias3[0], bias3[1]}, {bias3[0], bias3[1]}};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float bop[16];                                     // the B operand of every k-step of this row block
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (((s >> 1) & 1) == 0) {                     // channels 2 s, 2 s + 1 live in the lower lanes; the swap also yields step s + 2
                    const int r0 = ((2 * s) & 3) + 4 * ((2 * s) >> 3), r1 = ((2 * s + 1) & 3) + 4 * ((2 * s + 1) >> 3);
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(d2[rb][r0]), __float_as_int(d2[rb][r1]), false, false);
                    bop[s] = __int_as_float(sw[0]);
                    bop[s + 2] = __int_as_float(sw[1]);
                }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                d3[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa3[0][s], bop[s], d3[rb][0], 0, 0, 0);
                d3[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa3[1][s], bop[s], d3[rb][1], 0, 0, 0);
            }
        }
        // ---- ReLU, rows to LDS ([row][channel], four consecutive channels per store), pool with lane = channel
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = make_float4(fmaxf(d3[rb][cb][4 * q], 0.f), fmaxf(d3[rb][cb][4 * q + 1], 0.f),
                                                 fmaxf(d3[rb][cb][4 * q + 2], 0.f), fmaxf(d3[rb][cb][4 * q + 3], 0.f));
                    *reinterpret_cast<float4 *>(zw + (32 * rb + c) * LD + 32 * cb + 8 * q + 4 * h) = v;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int myc = cc[lane], prevc = cc[lane ? lane - 1 : 0];
        const unsigned long long start = __ballot(lane == 0 || myc != prevc);
        float cur = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            cur = fmax_raw(cur, zw[i * LD + lane]);
            const bool last = (i == 63) || ((start >> (i + 1)) & 1ull) != 0;
            if (last) {
                atomicMax(reinterpret_cast<int *>(out + (long)cc[i] * out_stride + out_col + lane), __float_as_int(cur));
                cur = 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();                       // the next tile overwrites the LDS tile and the centre list
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
    else {
        // (round 6: the VALU form of this scale, sa_xyz_mlp_packed_kernel<32, 32, 64>, and its switch PRCNN_XYZ_MFMA are gone:
        //  0.37 of the packed-f32 rate on LiDAR-shaped scenes against the matrix-core form below, same bits)
        const long cap = 2L * mfma_grid_cap();                 // persistent waves: the weights are loaded once per wave
        hipLaunchKernelGGL(sa_xyz_mlp_packed_mfma_kernel, dim3((unsigned)(grid < cap ? grid : cap)), dim3(256), 0, st, m, hdr, rowinfo,
                           (const float4 *)rowdxyz, tilecloud, w1, b1, w2, b2, w3, b3, out, out_stride, out_col);
    }
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
