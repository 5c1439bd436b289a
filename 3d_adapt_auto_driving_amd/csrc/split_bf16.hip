// EXPERIMENT (round 6): a per-point layer act(A @ W + bias) on the bf16 matrix cores with f32-class accuracy.
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at the f32 VECTOR rate (64 FLOP / clock / SIMD) and v_mfma_f32_32x32x16_bf16 sixteen times
// faster.  A float splits exactly into three bf16 pieces, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid):
// both differences are exact in f32, and 3 x 8 significand bits cover the 24 of a float), so a product is
//     x w = hi_x hi_w + (hi_x mid_w + mid_x hi_w) + (hi_x lo_w + lo_x hi_w + mid_x mid_w) + O(2^-32 |x w|),
// six bf16 MFMAs with f32 accumulation per k-step of 16 instead of eight f32 MFMAs of k = 2: 6 x 32 against 8 x 64 matrix-pipe
// cycles, 0.375 of the f32 form's.  The weights are split once (prcnn_split_weights_bf16x3, into the B-operand order of the
// instruction), the activations in registers on their way from HBM into the A operand.  NOT the arithmetic of the product path: the
// f32 kernels of csrc/packed_layer.hip compute fma chains that oracle/mlp_oracle.c restates bit for bit; this one agrees with them
// to ~1e-7 relative (profiles/r06_split_bf16.md) and stays behind the numerics switch PRCNN_SPLIT_BF16 (default off).
//
// Kernel: a workgroup of 4 waves owns 256 rows x 128 columns, a wave 64 rows (two 32-row blocks) x 4 column blocks of 32: 128
// accumulator registers.  Per k-step a lane loads the 8 consecutive floats of its row that the instruction wants from it (k = 8 (lane
// >> 5) .. + 7), splits them, and runs 4 column blocks x 2 row blocks x 6 MFMAs; the B pieces (16 bytes per lane and piece) come straight
// from L2 / L1 -- the four waves of a workgroup read the same 1 KB at the same time.
#include "common.hpp"

namespace prcnn {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)      // (bf16(a), bf16(b)) round-to-nearest-even, a in the low half
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

union Frag {
    unsigned u[4];
    bf16x8 v;
    uint4 q;
};

// x[0..7] -> hi / mid / lo fragments (8 bf16 each)
__device__ __forceinline__ void split8(const float (&x)[8], Frag &h, Frag &m, Frag &l)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned ph = cvt_pk_bf16(a, b);
        const float ra = a - bf_lo(ph), rb = b - bf_hi(ph);           // exact
        const unsigned pm = cvt_pk_bf16(ra, rb);
        const float ra2 = ra - bf_lo(pm), rb2 = rb - bf_hi(pm);       // exact
        h.u[i] = ph; m.u[i] = pm; l.u[i] = cvt_pk_bf16(ra2, rb2);
    }
}

// W (K, N) f32 row-major -> wsplit[((ks * (N / 32) + cb) * 3 + piece) * 64 + lane] (16 bytes each): the B operand of k-step ks
// (k = 16 ks + 8 (lane >> 5) + j), column block cb (n = 32 cb + (lane & 31))
__global__ __launch_bounds__(256) void split_weights_kernel(int K, int N, const float *__restrict__ W, uint4 *__restrict__ out)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)(K / 16) * (N / 32) * 64;
    if (g >= total) return;
    const int lane = (int)(g & 63);
    const long blk = g >> 6;
    const int cb = (int)(blk % (N / 32)), ks = (int)(blk / (N / 32));
    const int n = 32 * cb + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = W[(long)(k0 + j) * N + n];
    Frag h, m, l;
    split8(x, h, m, l);
    out[(blk * 3 + 0) * 64 + lane] = h.q;
    out[(blk * 3 + 1) * 64 + lane] = m.q;
    out[(blk * 3 + 2) * 64 + lane] = l.q;
}

__global__ __launch_bounds__(256, 2) void rows_layer_bf16x3_kernel(long rows, int K, int N, int n_store, const float *__restrict__ A, long lda,
                                                                 const uint4 *__restrict__ wsplit, const float *__restrict__ bias, int relu,
                                                                 float *__restrict__ out, long ldo)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * 256 + w * 64;
    const int cb0 = blockIdx.y * 4;                                    // first of this workgroup's four 32-column blocks
    const int ncb = N / 32;
    if (r0 >= rows) return;
    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;
    const float *arow[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        long r = r0 + rb * 32 + (lane & 31);
        r = r < rows ? r : rows - 1;                                   // ragged end: a valid row, its results are not stored
        arow[rb] = A + r * lda + 8 * (lane >> 5);
    }
    const int nks = K / 16;
    float4 nx[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        nx[rb][0] = *reinterpret_cast<const float4 *>(arow[rb]);
        nx[rb][1] = *reinterpret_cast<const float4 *>(arow[rb] + 4);
    }
    // B pieces one column block ahead (two register sets alternating): the loads of block cb + 1 -- or of block 0 of the next k-step --
    // are in flight while the twelve MFMAs of block cb run.  (Measured, profiles/r06_split_bf16.md: the same pieces staged through LDS once
    // per workgroup, two 24-KB buffers and a barrier per 32 k, are SLOWER -- 155 against 138 us at 32768 x 512 x 512.)
    const uint4 *wp = wsplit + (long)cb0 * 3 * 64 + lane;
    const long wstep = (long)ncb * 3 * 64;
    uint4 bq[2][3];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) bq[0][s_] = wp[s_ * 64];
    for (int ks = 0; ks < nks; ++ks) {
        Frag ah[2], am[2], al[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float x[8] = {nx[rb][0].x, nx[rb][0].y, nx[rb][0].z, nx[rb][0].w, nx[rb][1].x, nx[rb][1].y, nx[rb][1].z, nx[rb][1].w};
            split8(x, ah[rb], am[rb], al[rb]);
        }
        const int kn = ks + 1 < nks ? ks + 1 : ks;                     // (the last step prefetches itself again: no branch in the loop)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            nx[rb][0] = *reinterpret_cast<const float4 *>(arow[rb] + 16 * kn);
            nx[rb][1] = *reinterpret_cast<const float4 *>(arow[rb] + 16 * kn + 4);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int cur = cb & 1, nxt = cur ^ 1;
            const uint4 *np_ = cb < 3 ? wp + (long)ks * wstep + (cb + 1) * 3 * 64 : wp + (long)kn * wstep;
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) bq[nxt][s_] = np_[s_ * 64];
            Frag bh, bm, bl;
            bh.q = bq[cur][0]; bm.q = bq[cur][1]; bl.q = bq[cur][2];
            // the six products of both row blocks, small terms first; the two row blocks' chains are independent
            f32x16 c0 = acc[0][cb], c1 = acc[1][cb];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0].v, bh.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[1].v, bh.v, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0].v, bl.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1].v, bl.v, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0].v, bm.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1].v, bm.v, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0].v, bh.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1].v, bh.v, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0].v, bm.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1].v, bm.v, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0].v, bh.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1].v, bh.v, c1, 0, 0, 0);
            acc[0][cb] = c0; acc[1][cb] = c1;
        }
    }
    const bool full = r0 + 64 <= rows;                                 // wave-uniform: a whole tile stores without a per-lane condition
#pragma unroll                                                         // (a store under a per-lane condition makes the compiler wait for the store before it)
    for (int cb = 0; cb < 4; ++cb) {
        const int n = 32 * (cb0 + cb) + (lane & 31);
        const float bv = bias ? bias[n] : 0.f;
        if (32 * (cb0 + cb) >= n_store) continue;                       // (n_store is a multiple of 32 or ends the matrix: checked per lane below)
        const bool col_ok = n < n_store;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float *o = out + (r0 + rb * 32 + 4 * (lane >> 5)) * ldo + n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[rb][cb][i] + bv;
                if (relu) v = fmaxf(v, 0.f);
                const int dr = (i & 3) + 8 * (i >> 2);
                if (full) {
                    if (col_ok) o[(long)dr * ldo] = v;
                } else if (col_ok && r0 + rb * 32 + 4 * (lane >> 5) + dr < rows) {
                    o[(long)dr * ldo] = v;
                }
            }
        }
    }
}
}  // namespace prcnn
using namespace prcnn;

extern "C" int prcnn_split_weights_bf16x3(int K, int N, const float *W, void *out, void *stream)
{
    PRCNN_REQUIRE(K > 0 && N > 0 && K % 16 == 0 && N % 128 == 0, "split_weights_bf16x3: K=%d N=%d (K %% 16, N %% 128)", K, N);
    PRCNN_REQUIRE(W && out, "split_weights_bf16x3: null pointer");
    const long total = (long)(K / 16) * (N / 32) * 64;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, N, W, (uint4 *)out);
    return check_launch("split_weights_bf16x3");
}

extern "C" int prcnn_rows_layer_bf16x3(long rows, int K, int N, int n_store, const float *A, long lda, const void *wsplit, const float *bias,
                                       int relu, float *out, long ldo, void *stream)
{
    PRCNN_REQUIRE(rows >= 0 && K > 0 && K % 32 == 0 && N % 128 == 0 && n_store > 0 && n_store <= N, "rows_layer_bf16x3: rows=%ld K=%d N=%d n_store=%d (K %% 32, N %% 128)", rows, K, N, n_store);
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(A && wsplit && out && lda % 4 == 0 && ((uintptr_t)A & 15) == 0, "rows_layer_bf16x3: null / misaligned operand (lda %% 4, A 16-byte aligned)");
    hipLaunchKernelGGL(rows_layer_bf16x3_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)(N / 128)), dim3(256), 0, (hipStream_t)stream,
                       rows, K, N, n_store, A, lda, (const uint4 *)wsplit, bias, relu, out, ldo);
    return check_launch("rows_layer_bf16x3");
}
