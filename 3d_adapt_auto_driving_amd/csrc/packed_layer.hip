// packed_layer.hip -- shared-MLP layers of ANY width over row lists whose length lives on the device.
//
// The fused kernels of sa_mlp_fused.hip / sa_packed.hip keep a level's weights in registers, which caps the widths at
// 128-128-256.  The deeper RPN levels (pointrcnn/lib/config.py:58-61: 128-196-256, 256-256-512, 256-384-512) and the
// RCNN's GroupAll level (256-256-512) do not fit, so they run layer by layer on the SAME packed row lists
// (prcnn_ball_pack: only the distinct rows of every group):
//
//   packed_gather_affine_kernel   A1[r] = relu(P[point(r)] + Wxyz . (xyz[point(r)] - centre(r)))         (layer 1)
//   packed_layer_kernel<false>    Y[r]  = relu(A[r] @ W + b)                                              (layer 2)
//   packed_layer_kernel<true>     out[centre] = max over the centre's rows of relu(A[r] @ W + b)          (layer 3 + pool)
//
// One workgroup = one 64-row tile x one 128-column block; K runs in 128-deep panels: the panel of W goes to registers
// (64 per lane, same lane mapping as sa_mlp_fused.hip), the panel of A to LDS, 128 v_mfma_f32_32x32x2_f32 per panel.
// The grid is sized for the worst case (every ball full); the real tile count is read from the pack header and the
// surplus workgroups exit at once.  Widths are zero-padded to multiples of 128 by the caller (exact: relu(0) = 0 meets
// zero weight rows).  The same kernel with a host-side row count serves the per-point GEMMs (P = features @ W1f + b1).
// Summation order: panels in order, inside a panel k = s, s + 64 for s = 0..63 -- oracle/mlp_oracle.c
// orc_rows_layer_mfma reproduces it bit for bit.
#include "common.hpp"
#include <cstdlib>
#include "segmax.hpp"

namespace prcnn {

unsigned int *next_ticket(hipStream_t st);    // sa_mlp_fused.hip: a zeroed, self-resetting 16-word ticket record of the stream's ring

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PL_ROWS = 64;
constexpr int PL_LD = 128 + 4;

// One layer problem / one gather problem; a launch carries up to PL_MAX_BATCH of them (blockIdx.z picks): the independent
// scales of an MSG level run side by side in ONE launch instead of one small latency-bound launch each.
constexpr int PL_MAX_BATCH = 4;
struct PLProblem {
    const unsigned int *hdr; long rows_host; int K, N; const float *A; long lda; const float *W; const float *bias; int do_relu;
    float *out; long ldo; const unsigned int *rowinfo; const int *tilecloud; int m, out_col, n_store;
    int lds_pool;               // SEGMAX: tiles with many centres pool through the dead LDS tile (segmax.hpp); needs 16-byte aligned output rows
    // interpolation addend (prcnn_packed_layer_interp, round 3): out[r] = relu((A[r] @ W + bias) + ((w0 G[i0] + w1 G[i1]) + w2 G[i2])),
    // G (clouds, add_m, N-wide rows, leading dimension add_ldg), idx / weight (rows, 3), row r belongs to cloud r / add_n
    const float *addG; const int *addIdx; const float *addW; int add_n, add_m; long add_ldg;
};
struct PLBatch { PLProblem p[PL_MAX_BATCH]; };
struct PGProblem {
    int n, c1; const unsigned int *hdr; const float4 *rowdxyz; const float4 *P; const float4 *wxyz; const unsigned int *rowinfo;
    const int *tilecloud; float4 *out;
};
struct PGBatch { PGProblem p[PL_MAX_BATCH]; };

__global__ __launch_bounds__(256) void packed_gather_affine_kernel(const PGBatch bt)
{
    const PGProblem &pb = bt.p[blockIdx.z];
    const int n = pb.n, c1 = pb.c1;
    const unsigned int *__restrict__ hdr = pb.hdr;
    const float4 *__restrict__ rowdxyz = pb.rowdxyz;
    const float4 *__restrict__ P = pb.P;
    const float4 *__restrict__ wxyz = pb.wxyz;
    const unsigned int *__restrict__ rowinfo = pb.rowinfo;
    const int *__restrict__ tilecloud = pb.tilecloud;
    float4 *__restrict__ out = pb.out;
    const long t = blockIdx.x;
    if (t >= (long)hdr[0]) return;
    const int cloud = tilecloud[t];
    const int q4 = c1 / 4;                                 // float4 chunks per row
    const long pbase = (long)cloud * n;
    for (int e = threadIdx.x; e < PL_ROWS * q4; e += 256) {
        const int row = e / q4, q = e - row * q4;
        const int k = (int)(rowinfo[t * PL_ROWS + row] & 0xffffu);
        const float4 d = rowdxyz[t * PL_ROWS + row];
        const float dx = d.x, dy = d.y, dz = d.z;
        const float4 base = P[(pbase + k) * q4 + q];
        const float4 wx = wxyz[q], wy = wxyz[q4 + q], wz = wxyz[2 * q4 + q];
        const float4 v = affine_relu4(base, wx, wy, wz, dx, dy, dz);
        out[(t * PL_ROWS + row) * q4 + q] = v;
    }
}

#define PL_LOAD_W(dst, kk)                                                                              \
    _Pragma("unroll") for (int s = 0; s < 64; ++s)                                                      \
        dst[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, lane_off, (unsigned int)((kk) + s) * row_bytes, 0));
#define PL_LOAD_A(dst, kk)                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                     \
        long g = t * PL_ROWS + r0 + 8 * i;                                                              \
        if (g >= rows) g = rows - 1; /* ragged last tile (host-count mode): recompute the last row */   \
        dst[i] = *reinterpret_cast<const f32x4 *>(A + g * lda + (kk) + 4 * chunk);                     \
    }
#define PL_STORE_A(src)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                       \
        *reinterpret_cast<f32x4 *>(tile + (r0 + 8 * i) * PL_LD + 4 * chunk) = src[i];
// s_waitcnt vmcnt(0) (expcnt / lgkmcnt left alone): said explicitly so that the compiler's wait-count bookkeeping knows nothing
// older than the prefetch issued next is outstanding, and puts no wait between that prefetch and the MFMAs that hide it
#define PL_VM_DRAIN __builtin_amdgcn_s_waitcnt(0x0F70);
// ---- one 128-deep panel = 16 k-groups of 8 MFMAs ---------------------------------------------------------------------
// A wave issues in order, and while it streams MFMAs the other waves of its SIMD get next to no VALU / address-generation
// slots (s_memtime stamps, profiles/r02_stage_stamps.md: two co-resident workgroups take turns, the time of a launch is the
// SUM of every wave's MFMA time and of its non-MFMA issue time).  So everything that is not an MFMA is placed INSIDE the
// MFMA stream of the wave that needs it, where the matrix pipe is busy anyway, and costs no VALU:
//   * the A operands of k-group g+1 are read from LDS before the MFMAs of group g;
//   * PREFETCH: the next panel comes in behind this one -- its 8 rows of A as buffer loads (scalar row offsets; rows past
//     the end read as zero through the buffer's bounds check, no per-lane clamping) in groups 0-7, written to the OTHER LDS
//     tile in groups 8-15; its 64 weights six per group in groups 0-10.  Nothing is left for the end of the stage but
//     one barrier.
#define PL_GROUP_MFMA_HEAD(wf)                                                                          \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wf[4 * g + 0], acc0, 0, 0, 0);                    \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wf[4 * g + 0], acc1, 0, 0, 0);
#define PL_GROUP_MFMA_TAIL(wf)                                                                          \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wf[4 * g + 1], acc0, 0, 0, 0);                    \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, wf[4 * g + 1], acc1, 0, 0, 0);                    \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wf[4 * g + 2], acc0, 0, 0, 0);                    \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, wf[4 * g + 2], acc1, 0, 0, 0);                    \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wf[4 * g + 3], acc0, 0, 0, 0);                    \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, wf[4 * g + 3], acc1, 0, 0, 0);
#define PL_STAGE(T, wf)                                                                                 \
    {                                                                                                   \
        const float *a0p = (T) + j * PL_LD + 64 * h, *a1p = (T) + (32 + j) * PL_LD + 64 * h;            \
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p);                                               \
        f32x4 a1 = *reinterpret_cast<const f32x4 *>(a1p);                                               \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                \
            f32x4 n0 = a0, n1 = a1;                                                                     \
            if (g < 15) {                                                                               \
                n0 = *reinterpret_cast<const f32x4 *>(a0p + 4 * (g + 1));                               \
                n1 = *reinterpret_cast<const f32x4 *>(a1p + 4 * (g + 1));                               \
            }                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            PL_GROUP_MFMA_HEAD(wf)                                                                      \
            PL_GROUP_MFMA_TAIL(wf)                                                                      \
            a0 = n0; a1 = n1;                                                                           \
        }                                                                                               \
    }
// (the scalar offsets of the 64 weight loads and the 8 row loads RUN -- one s_add each, opaque to the optimiser: written as products
//  of constants with row_bytes the compiler hoists all 72 of them into SGPRs of their own, and a kernel with anything else to keep in
//  scalar registers -- the persistent one below -- spills them into VGPR lanes: 248 v_readlane / v_writelane inside the MFMA stream)
#define PL_SADD(x, y) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(y) : "scc");
#define PL_STAGE_PREFETCH_X(T, TN, wf, wn, kn, ARS, LOFF)                                               \
    {                                                                                                   \
        const float *a0p = (T) + j * PL_LD + 64 * h, *a1p = (T) + (32 + j) * PL_LD + 64 * h;            \
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p);                                               \
        f32x4 a1 = *reinterpret_cast<const f32x4 *>(a1p);                                               \
        f32x4 ar[8];                                                                                    \
        unsigned int wo_ = (unsigned int)(kn) * row_bytes, ao_ = (unsigned int)(kn) * 4u;               \
        const unsigned int a8_ = 8u * a_row_bytes;                                                      \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                \
            f32x4 n0 = a0, n1 = a1;                                                                     \
            if (g < 15) {                                                                               \
                n0 = *reinterpret_cast<const f32x4 *>(a0p + 4 * (g + 1));                               \
                n1 = *reinterpret_cast<const f32x4 *>(a1p + 4 * (g + 1));                               \
            }                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            PL_GROUP_MFMA_HEAD(wf)                                                                      \
            if (g < 8) {                                                                                \
                ar[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((ARS), a_lane, ao_, 0)); \
                PL_SADD(ao_, a8_)                                                                       \
            } else                                                                                      \
                *reinterpret_cast<f32x4 *>((TN) + (r0 + 8 * (g - 8)) * PL_LD + 4 * chunk) = ar[g - 8];  \
            _Pragma("unroll") for (int q = 0; q < 6; ++q)                                               \
                if (6 * g + q < 64) {                                                                   \
                    wn[6 * g + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, (LOFF), wo_, 0)); \
                    PL_SADD(wo_, row_bytes)                                                             \
                }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            PL_GROUP_MFMA_TAIL(wf)                                                                      \
            a0 = n0; a1 = n1;                                                                           \
        }                                                                                               \
    }
#define PL_STAGE_PREFETCH(T, TN, wf, wn, kn) PL_STAGE_PREFETCH_X(T, TN, wf, wn, kn, ars, lane_off)

// the same stage with the NEXT ROW TILE of a K = 128 layer coming in behind it (packed_layer_stream_kernel: the weights stay):
// 8 rows per thread as buffer loads at scalar offset `soff` in k-groups 0-7, written to the other LDS tile in k-groups 8-15
#define PL_STAGE_ROWS(T, TN, wf, soff)                                                                  \
    {                                                                                                   \
        const float *a0p = (T) + j * PL_LD + 64 * h, *a1p = (T) + (32 + j) * PL_LD + 64 * h;            \
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p);                                               \
        f32x4 a1 = *reinterpret_cast<const f32x4 *>(a1p);                                               \
        f32x4 ar[8];                                                                                    \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                \
            f32x4 n0_ = a0, n1_ = a1;                                                                   \
            if (g < 15) {                                                                               \
                n0_ = *reinterpret_cast<const f32x4 *>(a0p + 4 * (g + 1));                              \
                n1_ = *reinterpret_cast<const f32x4 *>(a1p + 4 * (g + 1));                              \
            }                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            PL_GROUP_MFMA_HEAD(wf)                                                                      \
            if (g < 8)                                                                                  \
                ar[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                \
                    ars, a_lane, (soff) + (unsigned int)(8 * g) * a_row_bytes, 0));                     \
            else                                                                                        \
                *reinterpret_cast<f32x4 *>((TN) + (r0 + 8 * (g - 8)) * PL_LD + 4 * chunk) = ar[g - 8];  \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            PL_GROUP_MFMA_TAIL(wf)                                                                      \
            a0 = n0_; a1 = n1_;                                                                         \
        }                                                                                               \
    }

// act(acc + bias): SEGMAX = false -> out[r][n0 + ..] for the tile's rows (rows >= `rows` are not stored), staged through
// the (dead) LDS tile for coalesced 16-byte stores; SEGMAX = true -> segmented max over the tile's rows by centre, atomicMax
// into out[centre][out_col + n0 + ..]
template <bool SEGMAX>
__device__ __forceinline__ void pl_epilogue(const f32x16 &acc0, const f32x16 &acc1, float *tile, int *ctr, long t, long rows, int n0,
                                            const float *__restrict__ bias, int do_relu, float *__restrict__ out, long ldo,
                                            const unsigned int *__restrict__ rowinfo, const int *__restrict__ tilecloud, int m,
                                            int out_col, int n_store, int lds_pool, const PLProblem *add = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
    const int chunk = tid & 31, r0 = tid >> 5;
    const float bcol = bias[n0 + 32 * w + j];
    if constexpr (SEGMAX) {
        if (tid < PL_ROWS) {
            const unsigned int info = rowinfo[t * PL_ROWS + tid];
            ctr[tid] = tilecloud[t] * m + (int)(info >> 16);
        }
        __syncthreads();
        const int myc = ctr[lane], prevc = ctr[lane ? lane - 1 : 0];
        const unsigned long long start = __ballot(lane == 0 || myc != prevc);
        if (!lds_pool || __popcll(start) <= PK_LDS_MIN) {
            pk_segmented_max(acc0, acc1, ctr, start, h, out, (int)ldo, out_col + n0 + 32 * w + j, bcol);
        } else {
            // many centres in the tile: row-major through the panel tile, which is dead (every wave is past the barrier above)
            const float4 bias4 = *reinterpret_cast<const float4 *>(bias + n0 + 4 * chunk);
            pk_park(acc0, acc1, tile, PL_LD, 32 * w + j, h);
            lds_barrier();
            pk_segmented_max_lds(tile, PL_LD, ctr, tid, out, (int)ldo, out_col + n0, bias4);
        }
    } else {
        const bool with_add = add && add->addG;
        __syncthreads();                                   // the panel tile is dead: stage the results through it
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v0 = acc0[r] + bcol, v1 = acc1[r] + bcol;
            tile[row * PL_LD + 32 * w + j] = (do_relu && !with_add) ? fmaxf(v0, 0.f) : v0;
            tile[(32 + row) * PL_LD + 32 * w + j] = (do_relu && !with_add) ? fmaxf(v1, 0.f) : v1;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = r0 + 8 * i;
            const long g = t * PL_ROWS + row;
            if (g < rows) {
                // n_store < N: only the first n_store columns exist in `out` (a head's last layer: N padded to 128 for the
                // MFMA tiles, 1 / 46 / 76 real outputs); 16-byte stores when the row layout allows them
                const int c0 = n0 + 4 * chunk;
                float4 v = *reinterpret_cast<const float4 *>(tile + row * PL_LD + 4 * chunk);
                if (with_add) {
                    // + the interpolated coarse-level product, then the activation: (w0 g0 + w1 g1) + w2 g2 per component with one
                    // rounding per operation (the order of three_interpolate, interpolate_gpu.cu:92-94), added to (acc + bias)
                    const int *ix = add->addIdx + g * 3;
                    const float *wv = add->addW + g * 3;
                    const long cb = (g / add->add_n) * (long)add->add_m;
                    const float4 g0 = *reinterpret_cast<const float4 *>(add->addG + (cb + ix[0]) * add->add_ldg + c0);
                    const float4 g1 = *reinterpret_cast<const float4 *>(add->addG + (cb + ix[1]) * add->add_ldg + c0);
                    const float4 g2 = *reinterpret_cast<const float4 *>(add->addG + (cb + ix[2]) * add->add_ldg + c0);
                    const float w0 = wv[0], w1 = wv[1], w2 = wv[2];
                    v.x = __fadd_rn(v.x, __fadd_rn(__fadd_rn(__fmul_rn(w0, g0.x), __fmul_rn(w1, g1.x)), __fmul_rn(w2, g2.x)));
                    v.y = __fadd_rn(v.y, __fadd_rn(__fadd_rn(__fmul_rn(w0, g0.y), __fmul_rn(w1, g1.y)), __fmul_rn(w2, g2.y)));
                    v.z = __fadd_rn(v.z, __fadd_rn(__fadd_rn(__fmul_rn(w0, g0.z), __fmul_rn(w1, g1.z)), __fmul_rn(w2, g2.z)));
                    v.w = __fadd_rn(v.w, __fadd_rn(__fadd_rn(__fmul_rn(w0, g0.w), __fmul_rn(w1, g1.w)), __fmul_rn(w2, g2.w)));
                    if (do_relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                }
                if (c0 + 4 <= n_store && (ldo & 3) == 0) {
                    *reinterpret_cast<float4 *>(out + g * ldo + c0) = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < n_store) out[g * ldo + c0 + q] = e[q];
                }
            }
        }
    }
}

// K = 128: one panel, nothing to prefetch; 3 workgroups per CU cover each other's loads
template <bool SEGMAX>
__global__ __launch_bounds__(256, 2) void packed_layer_kernel(
    const PLBatch bt)
{
    const PLProblem &pb = bt.p[blockIdx.z];
    const unsigned int *__restrict__ hdr = pb.hdr;
    const long rows_host = pb.rows_host;
    const int K = pb.K, N = pb.N;
    const float *__restrict__ A = pb.A;
    const long lda = pb.lda;
    const float *__restrict__ W = pb.W;
    const float *__restrict__ bias = pb.bias;
    const int do_relu = pb.do_relu;
    float *__restrict__ out = pb.out;
    const long ldo = pb.ldo;
    const unsigned int *__restrict__ rowinfo = pb.rowinfo;
    const int *__restrict__ tilecloud = pb.tilecloud;
    const int m = pb.m, out_col = pb.out_col, n_store = pb.n_store;
    __shared__ float tile[PL_ROWS * PL_LD];
    __shared__ int ctr[PL_ROWS];
    const long t = blockIdx.x;
    const long rows = hdr ? (long)hdr[0] * PL_ROWS : rows_host;
    const int n0 = blockIdx.y * 128;
    if (t * PL_ROWS >= rows || n0 >= n_store) return;          // (a batched launch is sized for its largest problem)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
    const int chunk = tid & 31, r0 = tid >> 5;
    f32x16 acc0 = {0}, acc1 = {0};
    // weights through a buffer resource: one 32-bit lane offset in a VGPR, the row offset of each load in a scalar register
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, K * N * 4, 0x00020000);
    const unsigned int row_bytes = (unsigned int)N * 4u;
    const unsigned int lane_off = ((unsigned int)(64 * h) * (unsigned int)N + (unsigned int)(n0 + 32 * w + j)) * 4u;
    for (int k0 = 0; k0 < K; k0 += 128) {
        float wf[64];
        f32x4 ar[8];
        PL_LOAD_W(wf, k0)
        PL_LOAD_A(ar, k0)
        if (k0) __syncthreads();                           // every wave has finished reading the previous panel
        PL_STORE_A(ar)
        __syncthreads();
        PL_STAGE(tile, wf)
    }
    pl_epilogue<SEGMAX>(acc0, acc1, tile, ctr, t, rows, n0, bias, do_relu, out, ldo, rowinfo, tilecloud, m, out_col, n_store, pb.lds_pool, &pb);
}

// K >= 256: two LDS tiles, two weight register sets, every panel fetched behind the MFMAs of the one before it
// (PL_STAGE_PREFETCH); the k order of every dot product is the same as in the one-panel kernel (panels in sequence).
template <bool SEGMAX>
__global__ __launch_bounds__(256, 2) void packed_layer_pipe_kernel(
    const PLBatch bt)
{
    const PLProblem &pb = bt.p[blockIdx.z];
    const unsigned int *__restrict__ hdr = pb.hdr;
    const long rows_host = pb.rows_host;
    const int K = pb.K, N = pb.N;
    const float *__restrict__ A = pb.A;
    const long lda = pb.lda;
    const float *__restrict__ W = pb.W;
    const float *__restrict__ bias = pb.bias;
    const int do_relu = pb.do_relu;
    float *__restrict__ out = pb.out;
    const long ldo = pb.ldo;
    const unsigned int *__restrict__ rowinfo = pb.rowinfo;
    const int *__restrict__ tilecloud = pb.tilecloud;
    const int m = pb.m, out_col = pb.out_col, n_store = pb.n_store;
    __shared__ float tiles[2 * PL_ROWS * PL_LD];
    __shared__ int ctr[PL_ROWS];
    const long t = blockIdx.x;
    const long rows = hdr ? (long)hdr[0] * PL_ROWS : rows_host;
    const int n0 = blockIdx.y * 128;
    if (t * PL_ROWS >= rows || n0 >= n_store) return;          // (a batched launch is sized for its largest problem)
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    f32x16 acc0 = {0}, acc1 = {0};
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, K * N * 4, 0x00020000);
    const unsigned int row_bytes = (unsigned int)N * 4u;
    const unsigned int lane_off = ((unsigned int)(64 * h) * (unsigned int)N + (unsigned int)(n0 + 32 * w + j)) * 4u;
    // this tile's rows of A as their own buffer: rows past the end (a ragged last tile) are out of its range and read as 0
    const long left = rows - t * PL_ROWS;
    const unsigned int a_row_bytes = (unsigned int)lda * 4u;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(A + t * PL_ROWS * lda), 0, (int)((left < PL_ROWS ? left : PL_ROWS) * (long)a_row_bytes), 0x00020000);
    const unsigned int a_lane = (unsigned int)r0 * a_row_bytes + 16u * chunk;
    float wa[64], wb[64];
    {
        PL_LOAD_W(wa, 0)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<f32x4 *>(tiles + (r0 + 8 * i) * PL_LD + 4 * chunk) =
                __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, a_lane, (unsigned int)(8 * i) * a_row_bytes, 0));
    }
    const int np = K >> 7;
    for (int p = 0; p < np; ++p) {
        PL_VM_DRAIN                                        // the panel about to be used is complete (weights; rows are in LDS)
        lds_barrier();                                     // ... and published; the other tile is free
        const float *T = tiles + (p & 1) * (PL_ROWS * PL_LD);
        float *TN = tiles + ((p + 1) & 1) * (PL_ROWS * PL_LD);
        if (p + 1 < np) {
            PL_STAGE_PREFETCH(T, TN, wa, wb, (p + 1) * 128)
#pragma unroll
            for (int s = 0; s < 64; ++s) wa[s] = wb[s];
        } else {
            PL_STAGE(T, wa)
        }
    }
    if constexpr (!SEGMAX) {
        if (n0 + 128 <= n_store && !pb.addG) {
            // whole 128-column blocks, no addend: straight from the accumulators as buffer stores into the tile's own rows of `out`
            // (packed_layer_persist_kernel's epilogue: no LDS staging, no barrier; rows past the end are outside the buffer) -- round 5:
            // the staged epilogue was 4.5-14 k of a workgroup's 95-120 k cycles in the launches of at most 512 items this kernel keeps
            const float bcol = bias[n0 + 32 * w + j];
            const unsigned int o_row_bytes = (unsigned int)ldo * 4u;
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(out + t * PL_ROWS * ldo), 0, (int)((left < PL_ROWS ? left : PL_ROWS) * (long)o_row_bytes), 0x00020000);
            const unsigned int o_lane = (unsigned int)(4 * h) * o_row_bytes + (unsigned int)(n0 + 32 * w + j) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned int so = (unsigned int)((r & 3) + 8 * (r >> 2)) * o_row_bytes;
                const float v0 = acc0[r] + bcol, v1 = acc1[r] + bcol;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(do_relu ? fmaxf(v0, 0.f) : v0), ors, o_lane, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(do_relu ? fmaxf(v1, 0.f) : v1), ors, o_lane, so + 32u * o_row_bytes, 0);
            }
            return;
        }
    }
    pl_epilogue<SEGMAX>(acc0, acc1, tiles, ctr, t, rows, n0, bias, do_relu, out, ldo, rowinfo, tilecloud, m, out_col, n_store, pb.lds_pool, &pb);
}

// ---- K >= 256 layers, PERSISTENT (round 4).  packed_layer_pipe_kernel pays per 64 x 128 tile what profiles/layer_k_sweep.py measures as
// the intercept of time over K: 9.4 us per round of workgroups -- dispatch, an exposed round trip for the first panel (64 weights per
// lane + 8 rows), the LDS-staged epilogue with two barriers, the drain of its stores -- against 7.8 us of MFMAs per panel: 0.55 of
// the MFMA peak at K = 256, 0.67 at K = 512 (0.88 is the slope).  Here a workgroup draws (row tile, column block) items from a ticket
// and never leaves the panel pipeline: the FIRST panel of the next item is fetched behind the MFMAs of the last panel of the running
// one (same PL_STAGE_PREFETCH, other buffer resources), and the epilogue needs no LDS and no barrier -- every lane stores its 32
// results straight from the accumulators (a wave's store instruction covers two 128-byte row segments).  Per output element the
// MFMA sequence, the bias add and the ReLU are those of the one-tile kernels: same bits.  Host-count mode, N == n_store, no pooling,
// no interpolation addend.
template <bool ADD>
__global__ __launch_bounds__(256, 2) void packed_layer_persist_kernel(
    long rows, int K, int N, const float *__restrict__ A, long lda, const float *__restrict__ W, const float *__restrict__ bias,
    int do_relu, float *__restrict__ out, long ldo, unsigned int *__restrict__ ticket, const PLProblem addp)
{
    __shared__ float tiles[2 * PL_ROWS * PL_LD];
    __shared__ unsigned int s_item[2];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    const unsigned int col_blocks = (unsigned int)(N >> 7);
    const unsigned int n_tiles = (unsigned int)((rows + PL_ROWS - 1) / PL_ROWS);
    const unsigned int n_items = n_tiles * col_blocks;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, K * N * 4, 0x00020000);
    const unsigned int row_bytes = (unsigned int)N * 4u;
    const unsigned int a_row_bytes = (unsigned int)lda * 4u;
    const unsigned int a_lane = (unsigned int)r0 * a_row_bytes + 16u * chunk;
    const int np = K >> 7;
    // the rows of a tile as their own buffer: rows past the end (a ragged last tile) read as 0
    auto tile_rsrc = [&](unsigned int t) __attribute__((always_inline)) {
        const long left = rows - (long)t * PL_ROWS;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(A + (long)t * PL_ROWS * lda), 0,
                                                 (int)((left < PL_ROWS ? left : PL_ROWS) * (long)a_row_bytes), 0x00020000);
    };
    auto loff = [&](unsigned int cb) __attribute__((always_inline)) {
        return ((unsigned int)(64 * h) * (unsigned int)N + (cb * 128u + (unsigned int)(32 * w + j))) * 4u;
    };
    // the first item is the workgroup's index (no round trip in front of the first loads; the host launches no more workgroups than
    // items), the later ones are drawn: gridDim.x + ticket
    unsigned int item = blockIdx.x;
    // items in column-block-minor order: neighbours in the draw order share their rows of A
    unsigned int t = item / col_blocks, cb = item - t * col_blocks;
    __amdgpu_buffer_rsrc_t ars = tile_rsrc(t);
    unsigned int lane_off = loff(cb);
    float wa[64], wb[64];
    {
        PL_LOAD_W(wa, 0)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<f32x4 *>(tiles + (r0 + 8 * i) * PL_LD + 4 * chunk) =
                __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, a_lane, (unsigned int)(8 * i) * a_row_bytes, 0));
    }
    int pp = 0;                                              // panels staged so far: the LDS tile alternates across items too
    int slot = 1;
    PL_VM_DRAIN                                              // the first item's first panel has arrived
    while (true) {
        f32x16 acc0 = {0}, acc1 = {0};
        unsigned int next = 0xffffffffu, tn = 0, cbn = 0, drawn = 0;
        const float bcol = bias[(int)cb * 128 + 32 * w + j];    // (arrives behind the first panel: waited for by the drain of the second)
        for (int p = 0; p < np; ++p, ++pp) {
            // the panel about to be used is complete (weights in registers, rows in LDS): panel 0 of an item was waited for at the end
            // of the item before it (or above) -- no wait here, so that the stores of that item's results stay in flight behind this
            // panel's MFMAs (vmcnt counts loads and stores alike)
            if (p > 0) {
                PL_VM_DRAIN
                if (p == 1 && tid == 0) s_item[slot] = drawn;
            }
            lds_barrier();                                   // ... and published; the other tile is free
            // the NEXT item is drawn here: the atomic's round trip hides behind this panel's MFMAs, its value is read at the last panel
            if (p == 0 && tid == 0) drawn = gridDim.x + atomicAdd(ticket, 1u);
            const float *T = tiles + (pp & 1) * (PL_ROWS * PL_LD);
            float *TN = tiles + ((pp + 1) & 1) * (PL_ROWS * PL_LD);
            if (p + 1 < np) {
                PL_STAGE_PREFETCH(T, TN, wa, wb, (p + 1) * 128)
#pragma unroll
                for (int s2 = 0; s2 < 64; ++s2) wa[s2] = wb[s2];
            } else {
                next = s_item[slot];                         // (np >= 2: published before this panel's barrier)
                if (next < n_items) {
                    tn = next / col_blocks; cbn = next - tn * col_blocks;
                    const __amdgpu_buffer_rsrc_t ars_n = tile_rsrc(tn);
                    const unsigned int loff_n = loff(cbn);
                    PL_STAGE_PREFETCH_X(T, TN, wa, wb, 0, ars_n, loff_n)
#pragma unroll
                    for (int s2 = 0; s2 < 64; ++s2) wa[s2] = wb[s2];
                    PL_VM_DRAIN                              // (fetched behind 128 MFMAs: normally there already)
                } else {
                    PL_STAGE(T, wa)
                }
            }
        }
        // epilogue straight from the accumulators: lane (column j of its wave's 32, rows (r & 3) + 8 (r >> 2) + 4 h [+ 32]), as buffer
        // stores into the tile's own rows of `out` -- rows past the end of a ragged last tile are outside the buffer and dropped by its
        // bounds check: no per-lane compare, no 64-bit address arithmetic, and above all no wait in front of a store (a store under a
        // per-lane condition made the compiler wait for vmcnt(0) -- i.e. for the store before it -- 32 times per item)
        if constexpr (ADD) {
            // ADD: + the interpolated coarse-level product (prcnn_packed_layer_interp) -- the staged epilogue of the one-tile kernels, through
            // the LDS tile of the panel just consumed (the next item's first panel sits in the OTHER tile)
            float *Tlast = tiles + ((pp - 1) & 1) * (PL_ROWS * PL_LD);
            pl_epilogue<false>(acc0, acc1, Tlast, nullptr, (long)t, rows, (int)cb * 128, bias, do_relu, out, ldo, nullptr, nullptr, 0, 0, N, 0, &addp);
            __syncthreads();
        } else {
            const long left = rows - (long)t * PL_ROWS;
            const unsigned int o_row_bytes = (unsigned int)ldo * 4u;
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(out + (long)t * PL_ROWS * ldo), 0, (int)((left < PL_ROWS ? left : PL_ROWS) * (long)o_row_bytes), 0x00020000);
            const unsigned int o_lane = (unsigned int)(4 * h) * o_row_bytes + (cb * 128u + (unsigned int)(32 * w + j)) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned int so = (unsigned int)((r & 3) + 8 * (r >> 2)) * o_row_bytes;
                const float v0 = acc0[r] + bcol, v1 = acc1[r] + bcol;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(do_relu ? fmaxf(v0, 0.f) : v0), ors, o_lane, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(do_relu ? fmaxf(v1, 0.f) : v1), ors, o_lane, so + 32u * o_row_bytes, 0);
            }
        }
        if (next >= n_items) break;
        t = tn; cb = cbn;
        ars = tile_rsrc(t);
        lane_off = loff(cb);
        slot ^= 1;
    }
    if (tid == 0) ticket_release(ticket);
}

// ---- K = 128 layers over many row tiles (the per-point parts of the SA levels: 10^5 rows, one panel): PERSISTENT workgroups.
// With one tile per workgroup every tile paid its own 64 weight loads (64 KB per workgroup from L2), an exposed round trip for
// its rows and an epilogue nobody overlapped: 55-59 TFLOP/s (profiles/r02_microbench.md).  Here a workgroup keeps its 128 x 32
// weight slices in registers and strides over the row tiles; the next tile's rows arrive behind the MFMAs of the running one
// (PL_STAGE_ROWS, two LDS tiles).  Per row the arithmetic is that of packed_layer_kernel: same bits.  Host-count mode, no pooling.
__global__ __launch_bounds__(256, 2) void packed_layer_stream_kernel(long rows, int N, const float *__restrict__ A, long lda,
                                                                     const float *__restrict__ W, const float *__restrict__ bias,
                                                                     int do_relu, float *__restrict__ out, long ldo, int n_store,
                                                                     const PLProblem addp /* .addG != NULL: the interpolation addend */)
{
    __shared__ float tiles[2 * PL_ROWS * PL_LD];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    const int n0 = blockIdx.y * 128;
    const long ntiles = (rows + PL_ROWS - 1) / PL_ROWS;
    long t = blockIdx.x;
    if (t >= ntiles || n0 >= n_store) return;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, 128 * N * 4, 0x00020000);
    const unsigned int row_bytes = (unsigned int)N * 4u;
    const unsigned int lane_off = ((unsigned int)(64 * h) * (unsigned int)N + (unsigned int)(n0 + 32 * w + j)) * 4u;
    float wf[64];
    PL_LOAD_W(wf, 0)
    // the whole matrix as one buffer: rows past the end (the ragged last tile) are out of its range and read as zero
    const unsigned int a_row_bytes = (unsigned int)lda * 4u;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)(rows * (long)a_row_bytes), 0x00020000);
    const unsigned int a_lane = (unsigned int)r0 * a_row_bytes + 16u * chunk;
    {
        const unsigned int soff = (unsigned int)(t * PL_ROWS) * a_row_bytes;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<f32x4 *>(tiles + (r0 + 8 * i) * PL_LD + 4 * chunk) =
                __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, a_lane, soff + (unsigned int)(8 * i) * a_row_bytes, 0));
    }
    // round 4: the bias column of this workgroup is loaded once; no wait at the top of an iteration (the rows of the tile were waited for
    // when they went to LDS) -- so the results of the tile before stay in flight behind this tile's MFMAs -- and the epilogue writes
    // straight from the accumulators as buffer stores (see packed_layer_persist_kernel) when whole 128-column blocks are stored
    const float bcol = bias[n0 + 32 * w + j];
    const bool direct = n0 + 128 <= n_store && !addp.addG;
    const unsigned int o_row_bytes = (unsigned int)ldo * 4u;
    const unsigned int o_lane = (unsigned int)(4 * h) * o_row_bytes + (unsigned int)(n0 + 32 * w + j) * 4u;
    PL_VM_DRAIN
    for (int it = 0;; ++it) {
        lds_barrier();                                     // this tile's rows are published; the other tile is free
        float *T = tiles + (it & 1) * (PL_ROWS * PL_LD);
        float *TN = tiles + ((it + 1) & 1) * (PL_ROWS * PL_LD);
        const long tn = t + gridDim.x;
        f32x16 acc0 = {0}, acc1 = {0};
        if (tn < ntiles) {
            const unsigned int soff = (unsigned int)(tn * PL_ROWS) * a_row_bytes;
            PL_STAGE_ROWS(T, TN, wf, soff)
        } else {
            PL_STAGE(T, wf)
        }
        if (direct) {
            const long left = rows - t * PL_ROWS;
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(out + t * PL_ROWS * ldo), 0, (int)((left < PL_ROWS ? left : PL_ROWS) * (long)o_row_bytes), 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned int so = (unsigned int)((r & 3) + 8 * (r >> 2)) * o_row_bytes;
                const float v0 = acc0[r] + bcol, v1 = acc1[r] + bcol;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(do_relu ? fmaxf(v0, 0.f) : v0), ors, o_lane, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(do_relu ? fmaxf(v1, 0.f) : v1), ors, o_lane, so + 32u * o_row_bytes, 0);
            }
        } else {
            pl_epilogue<false>(acc0, acc1, T, nullptr, t, rows, n0, bias, do_relu, out, ldo, nullptr, nullptr, 0, 0, n_store, 0, &addp);
            __syncthreads();                               // (its staging through T is over before the next stage prefetches into it)
        }
        if (tn >= ntiles) break;
        t = tn;
    }
}

// ---- 32-row tiles: the layers that would give fewer than 256 workgroups of 64 rows (FP3, SA4's per-point parts, the RCNN
// heads: a few thousand rows, many panels).  Their launch time is one workgroup's serial panel chain; halving the rows of a
// tile halves the MFMAs of a stage (one accumulator per wave: 64 MFMAs, each waiting for the one before it -- the pipe's
// latency equals its occupancy for this instruction) and doubles the workgroups.  Per row the k order is that of the 64-row
// kernels: same bits.  Host-count mode only.
#define PL32_GROUP(wf, BODY)                                                                            \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wf[4 * g + 0], acc, 0, 0, 0);                       \
    BODY                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wf[4 * g + 1], acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wf[4 * g + 2], acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wf[4 * g + 3], acc, 0, 0, 0);
__global__ __launch_bounds__(256, 2) void packed_layer_pipe32_kernel(const PLBatch bt)
{
    // (round 4: up to PL_MAX_BATCH independent problems per launch, blockIdx.z picks -- the two branches of the RCNN head run side by
    //  side instead of one 20-us launch after the other on the feature stream; the grid is sized for the largest problem)
    const PLProblem &pb = bt.p[blockIdx.z];
    const long rows = pb.rows_host;
    const int K = pb.K, N = pb.N;
    const float *__restrict__ A = pb.A;
    const long lda = pb.lda;
    const float *__restrict__ W = pb.W;
    const float *__restrict__ bias = pb.bias;
    const int do_relu = pb.do_relu;
    float *__restrict__ out = pb.out;
    const long ldo = pb.ldo;
    const int n_store = pb.n_store;
    constexpr int R = 32;
    __shared__ float tiles[2 * R * PL_LD];
    const long t = blockIdx.x;
    const int n0 = blockIdx.y * 128;
    if (t * R >= rows || n0 >= n_store) return;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    f32x16 acc = {0};
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, K * N * 4, 0x00020000);
    const unsigned int row_bytes = (unsigned int)N * 4u;
    const unsigned int lane_off = ((unsigned int)(64 * h) * (unsigned int)N + (unsigned int)(n0 + 32 * w + j)) * 4u;
    const long left = rows - t * R;
    const unsigned int a_row_bytes = (unsigned int)lda * 4u;
    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc((void *)(A + t * R * lda), 0, (int)((left < R ? left : R) * (long)a_row_bytes), 0x00020000);
    const unsigned int a_lane = (unsigned int)r0 * a_row_bytes + 16u * chunk;
    float wa[64], wb[64];
    {
        PL_LOAD_W(wa, 0)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4 *>(tiles + (r0 + 8 * i) * PL_LD + 4 * chunk) =
                __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, a_lane, (unsigned int)(8 * i) * a_row_bytes, 0));
    }
    const int np = K >> 7;
    for (int p = 0; p < np; ++p) {
        PL_VM_DRAIN
        lds_barrier();
        const float *T = tiles + (p & 1) * (R * PL_LD);
        float *TN = tiles + ((p + 1) & 1) * (R * PL_LD);
        const float *ap = T + j * PL_LD + 64 * h;
        f32x4 a = *reinterpret_cast<const f32x4 *>(ap);
        if (p + 1 < np) {
            const int kn = (p + 1) * 128;
            f32x4 ar[4];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                f32x4 nx = a;
                if (g < 15) nx = *reinterpret_cast<const f32x4 *>(ap + 4 * (g + 1));
                __builtin_amdgcn_sched_barrier(0);
                PL32_GROUP(wa,
                    if (g < 4)
                        ar[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            ars, a_lane, (unsigned int)(8 * g) * a_row_bytes + (unsigned int)kn * 4u, 0));
                    else if (g >= 8 && g < 12)
                        *reinterpret_cast<f32x4 *>(TN + (r0 + 8 * (g - 8)) * PL_LD + 4 * chunk) = ar[g - 8];
                    _Pragma("unroll") for (int q = 0; q < 6; ++q)
                        if (6 * g + q < 64)
                            wb[6 * g + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                wrs, lane_off, (unsigned int)(kn + 6 * g + q) * row_bytes, 0));)
                a = nx;
            }
#pragma unroll
            for (int s = 0; s < 64; ++s) wa[s] = wb[s];
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                f32x4 nx = a;
                if (g < 15) nx = *reinterpret_cast<const f32x4 *>(ap + 4 * (g + 1));
                __builtin_amdgcn_sched_barrier(0);
                PL32_GROUP(wa, )
                a = nx;
            }
        }
    }
    const float bcol = bias[n0 + 32 * w + j];
    __syncthreads();                                       // the panel tiles are dead: stage the results through one
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[r] + bcol;
        tiles[row * PL_LD + 32 * w + j] = do_relu ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 8 * i;
        const long g = t * R + row;
        if (g < rows) {
            const int c0 = n0 + 4 * chunk;
            const float4 v = *reinterpret_cast<const float4 *>(tiles + row * PL_LD + 4 * chunk);
            if (c0 + 4 <= n_store && (ldo & 3) == 0) {
                *reinterpret_cast<float4 *>(out + g * ldo + c0) = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c0 + q < n_store) out[g * ldo + c0 + q] = e[q];
            }
        }
    }
}

// out[r][0..n) = A[r][0..K) @ W + bias for n <= 4 output columns (the 1-wide last layer of the classification heads): a
// GEMV per output, no MFMA tile to fill.  32 lanes per row: lane l accumulates k = l, l + 32, ... as one fma chain (from 0),
// the 32 partial sums are added in an xor butterfly (16, 8, 4, 2, 1), then the bias.  oracle/mlp_oracle.c orc_rows_dot
// restates that order.
__global__ __launch_bounds__(256) void rows_dot_kernel(long rows, int K, int n, const float *__restrict__ A, long lda,
                                                       const float *__restrict__ W, const float *__restrict__ bias,
                                                       float *__restrict__ out, long ldo)
{
    const long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l = threadIdx.x & 31;
    const long rr = r < rows ? r : rows - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = l; k < K; k += 32) {
        const float a = A[rr * lda + k];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < n) acc[c] = fmaf(a, W[(long)k * n + c], acc[c]);
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __fadd_rn(acc[c], __shfl_xor(acc[c], d, 32));
    if (l == 0 && r < rows)
        for (int c = 0; c < n; ++c) out[r * ldo + c] = __fadd_rn(acc[c], bias[c]);
}

}  // namespace prcnn

using namespace prcnn;

// (round 6: the A/B switches PRCNN_PL_PIPE / PRCNN_PL_STREAM / PRCNN_PL_PERSIST / PRCNN_SEGMAX_LDS are gone -- the forms they selected are
// chosen by shape below, each still reached by the shapes it serves: tests/test_gpu_packed.py)
static bool pipe_enabled() { return true; }

static long stream_min() { static const long v = getenv("PRCNN_PL_STREAM_MIN") ? atol(getenv("PRCNN_PL_STREAM_MIN")) : 512; return v; }
static long persist_min() { static const long v = getenv("PRCNN_PL_PERSIST_MIN") ? atol(getenv("PRCNN_PL_PERSIST_MIN")) : 256; return v; }
static long stream_cap() { static const long v = getenv("PRCNN_PL_STREAM_CAP") ? atol(getenv("PRCNN_PL_STREAM_CAP")) : 512; return v; }
static bool stream_enabled() { return true; }
static bool segmax_lds_enabled() { return true; }

extern "C" int prcnn_rows_dot(long rows, int K, int n, const float *A, long lda, const float *W, const float *bias, float *out,
                              long ldo, void *stream)
{
    PRCNN_REQUIRE(rows >= 0 && K > 0 && n >= 1 && n <= 4 && lda >= K && ldo >= n, "rows_dot: bad sizes (n <= 4)");
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(A && W && bias && out, "rows_dot: null pointer");
    PRCNN_REQUIRE((rows + 7) / 8 <= 0x7fffffffL, "rows_dot: too many rows");
    hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, rows, K, n, A, lda, W,
                       bias, out, ldo);
    return check_launch("rows_dot");
}

// A1 (max_tiles*64, c1) = relu(P[point] + wxyz . rowdxyz) for every packed row (prcnn_ball_pack: rowdxyz = xyz[point] - centre);
// P (b,n,c1), wxyz (3,c1), c1 % 4 == 0.  Up to 4 problems (the scales of one MSG level) share one launch.
extern "C" int prcnn_packed_gather_affine_batch(int nprob, const prcnn_gather_problem *pr, void *stream)
{
    PRCNN_REQUIRE(nprob >= 0 && nprob <= PL_MAX_BATCH && (nprob == 0 || pr), "packed_gather_affine: 0..%d problems", PL_MAX_BATCH);
    PGBatch bt;
    int k = 0;
    long grid_x = 0;
    for (int i = 0; i < nprob; ++i) {
        const prcnn_gather_problem &q = pr[i];
        PRCNN_REQUIRE(q.b >= 0 && q.n >= 0 && q.max_tiles >= 0 && q.c1 > 0 && q.c1 % 4 == 0, "packed_gather_affine: bad sizes");
        if (q.max_tiles == 0) continue;
        PRCNN_REQUIRE(q.P && q.wxyz && q.rowinfo && q.rowdxyz && q.tilecloud && q.hdr && q.out, "packed_gather_affine: null pointer");
        PRCNN_REQUIRE((((uintptr_t)q.P | (uintptr_t)q.wxyz | (uintptr_t)q.out) & 15) == 0, "packed_gather_affine: 16-byte alignment required");
        PRCNN_REQUIRE(q.max_tiles <= 0x7fffffffL, "packed_gather_affine: too many tiles");
        bt.p[k++] = PGProblem{q.n, q.c1, q.hdr, (const float4 *)q.rowdxyz, (const float4 *)q.P, (const float4 *)q.wxyz, q.rowinfo,
                              q.tilecloud, (float4 *)q.out};
        if (q.max_tiles > grid_x) grid_x = q.max_tiles;
    }
    if (k == 0) return PRCNN_OK;
    hipLaunchKernelGGL(packed_gather_affine_kernel, dim3((unsigned)grid_x, 1, k), dim3(256), 0, (hipStream_t)stream, bt);
    return check_launch("packed_gather_affine");
}

extern "C" int prcnn_packed_gather_affine(int b, int n, int c1, long max_tiles, const float *P, const float *wxyz,
                                          const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud,
                                          const unsigned int *hdr, float *out, void *stream)
{
    const prcnn_gather_problem q = {b, n, c1, max_tiles, P, wxyz, rowinfo, rowdxyz, tilecloud, hdr, out};
    return prcnn_packed_gather_affine_batch(1, &q, stream);
}

// Layer problems (plain: out[r][0..n_store) = act(A[r][0..K) @ W + bias)[0..n_store); segmax: last layer of a level + max pool,
// out[(b*m)][out_col .. out_col + N) = max over each centre's packed rows of relu(A[r] @ W + bias) through atomicMax into a zeroed
// slice).  K and N multiples of 128, W (K,N) k-major, n_store <= N (the caller pads a narrow last layer's weights to N = 128 and
// asks for its real width).  Row count: hdr != NULL -> hdr[0] * 64 rows (a packed list; max_tiles sizes the grid), else `rows`
// (host count).  Up to 4 problems that land on the same kernel share ONE launch (grid.z); others are launched one by one.
namespace {
enum PLClass { PL_ONE, PL_PIPE, PL_PIPE32, PL_STREAM };
}
extern "C" int prcnn_packed_layer_batch(int nprob, const prcnn_layer_problem *pr, int segmax, void *stream)
{
    PRCNN_REQUIRE(nprob >= 0 && nprob <= PL_MAX_BATCH && (nprob == 0 || pr), "packed_layer: 0..%d problems", PL_MAX_BATCH);
    hipStream_t st = (hipStream_t)stream;
    PLBatch bt;
    PLClass cls[PL_MAX_BATCH];
    long tiles_of[PL_MAX_BATCH];
    int blocks_of[PL_MAX_BATCH], src[PL_MAX_BATCH];
    int k = 0;
    for (int i = 0; i < nprob; ++i) {
        const prcnn_layer_problem &q = pr[i];
        PRCNN_REQUIRE(q.K > 0 && q.N > 0 && q.K % 128 == 0 && q.N % 128 == 0, "packed_layer: K=%d, N=%d must be multiples of 128", q.K, q.N);
        PRCNN_REQUIRE(q.lda >= q.K && q.lda % 4 == 0, "packed_layer: bad leading dimension of A");
        long tiles;
        int n_store;
        if (segmax) {
            PRCNN_REQUIRE(q.b >= 0 && q.m >= 0 && q.max_tiles >= 0 && q.out_col >= 0 && q.ldo >= q.out_col + q.N, "packed_layer_segmax: bad layout");
            if ((long)q.b * q.m == 0) continue;
            PRCNN_REQUIRE(q.A && q.W && q.bias && q.rowinfo && q.tilecloud && q.hdr && q.out, "packed_layer_segmax: null pointer");
            PRCNN_REQUIRE(((uintptr_t)q.A & 15) == 0 && q.max_tiles <= 0x7fffffffL, "packed_layer_segmax: alignment / size");
            if (!q.out_is_zero && hipMemset2DAsync(q.out + q.out_col, (size_t)q.ldo * sizeof(float), 0, (size_t)q.N * sizeof(float),
                                                    (size_t)q.b * q.m, st) != hipSuccess) {
                set_error("packed_layer_segmax: cannot zero the output slice");
                return PRCNN_ELAUNCH;
            }
            tiles = q.max_tiles;
            n_store = q.N;
        } else {
            PRCNN_REQUIRE(q.n_store >= 1 && q.n_store <= q.N && q.ldo >= q.n_store, "packed_layer: n_store=%d outside 1..N (or ldo too small)", q.n_store);
            tiles = q.hdr ? q.max_tiles : (q.rows + PL_ROWS - 1) / PL_ROWS;
            PRCNN_REQUIRE(tiles >= 0 && tiles <= 0x7fffffffL && q.rows >= 0, "packed_layer: bad row count");
            n_store = q.n_store;
            if (tiles > 0) {
                PRCNN_REQUIRE(q.A && q.W && q.bias && q.out, "packed_layer: null pointer");
                PRCNN_REQUIRE(((uintptr_t)q.A & 15) == 0 && (((uintptr_t)q.out & 15) == 0 || (q.ldo & 3) != 0), "packed_layer: 16-byte alignment required");
            }
        }
        if (tiles == 0) continue;
        const int col_blocks = (n_store + 127) / 128;      // column blocks that hold nothing to store are not launched
        const bool pipe = q.K >= 256 && pipe_enabled();
        cls[k] = (!segmax && !q.hdr && pipe && tiles * col_blocks < 256) ? PL_PIPE32 : (pipe ? PL_PIPE : PL_ONE);
        // one panel, many row tiles, rows counted on the host: persistent workgroups (few items: one tile per workgroup)
        if (cls[k] == PL_ONE && !segmax && !q.hdr && q.K == 128 && tiles * col_blocks >= stream_min() && stream_enabled() &&
            q.rows * q.lda * 4 < (1L << 31))
            cls[k] = PL_STREAM;
        tiles_of[k] = tiles; blocks_of[k] = col_blocks; src[k] = i;
        bt.p[k] = PLProblem{q.hdr, q.rows, q.K, q.N, q.A, q.lda, q.W, q.bias, segmax ? 1 : q.relu, q.out, q.ldo,
                            segmax ? q.rowinfo : nullptr, segmax ? q.tilecloud : nullptr, segmax ? q.m : 0, segmax ? q.out_col : 0, n_store,
                            (segmax && segmax_lds_enabled() && (((uintptr_t)q.out | (uintptr_t)q.bias) & 15) == 0 && (q.ldo & 3) == 0 &&
                             (q.out_col & 3) == 0) ? 1 : 0,
                            nullptr, nullptr, nullptr, 0, 0, 0};
        ++k;
    }
    if (k == 0) return PRCNN_OK;
    // one plain K >= 256 layer over rows counted on the host, whole 128-column blocks stored: the persistent pipeline (round 4;
    // few items: one tile per workgroup as in round 3, same bits)
    const bool persist = true;
    if (persist && k == 1 && cls[0] == PL_PIPE && !segmax && !bt.p[0].hdr && bt.p[0].n_store == bt.p[0].N && !pr[src[0]].hdr &&
        tiles_of[0] * blocks_of[0] > stream_cap() && tiles_of[0] * blocks_of[0] >= persist_min()) {   // (items <= workgroups: nothing to pipeline)
        const PLProblem &q = bt.p[0];
        const long items = tiles_of[0] * blocks_of[0];
        const long cap = stream_cap();
        unsigned int *ticket = next_ticket(st);
        if (!ticket) { set_error("packed_layer: cannot set up the item ticket"); return PRCNN_ELAUNCH; }
        hipLaunchKernelGGL(packed_layer_persist_kernel<false>, dim3((unsigned)(items < cap ? items : cap)), dim3(256), 0, st, q.rows_host, q.K, q.N,
                           q.A, q.lda, q.W, q.bias, q.do_relu, q.out, q.ldo, ticket, PLProblem{});
        return check_launch("packed_layer");
    }
    bool together = cls[0] != PL_STREAM;
    for (int i = 1; i < k; ++i) together = together && cls[i] == cls[0];
    if (together && cls[0] == PL_PIPE32) {
        long gx = 0;
        int gy = 0;
        for (int r = 0; r < k; ++r) {
            const long t32 = (bt.p[r].rows_host + 31) / 32;
            if (t32 > gx) gx = t32;
            if (blocks_of[r] > gy) gy = blocks_of[r];
        }
        hipLaunchKernelGGL(packed_layer_pipe32_kernel, dim3((unsigned)gx, gy, k), dim3(256), 0, st, bt);
        return check_launch("packed_layer");
    }
    for (int i = 0; i < k; ++i) {
        PLBatch one;
        long gx = tiles_of[i];
        int gy = blocks_of[i], gz = 1;
        if (together) {
            for (int r = 1; r < k; ++r) { if (tiles_of[r] > gx) gx = tiles_of[r]; if (blocks_of[r] > gy) gy = blocks_of[r]; }
            gz = k;
        } else {
            one.p[0] = bt.p[i];
        }
        const PLBatch &arg = together ? bt : one;
        if (cls[i] == PL_STREAM) {
            const prcnn_layer_problem &q = pr[src[i]];
            const long cap = stream_cap() / gy > 0 ? stream_cap() / gy : 1;  // two resident workgroups per CU over all column blocks
            hipLaunchKernelGGL(packed_layer_stream_kernel, dim3((unsigned)(tiles_of[i] < cap ? tiles_of[i] : cap), gy), dim3(256), 0, st,
                               q.rows, q.N, q.A, q.lda, q.W, q.bias, q.relu, q.out, q.ldo, q.n_store, PLProblem{});
        } else if (cls[i] == PL_PIPE32) {
            const prcnn_layer_problem &q = pr[src[i]];
            hipLaunchKernelGGL(packed_layer_pipe32_kernel, dim3((unsigned)((q.rows + 31) / 32), gy), dim3(256), 0, st, one);
        } else if (segmax) {
            auto kern = cls[i] == PL_PIPE ? packed_layer_pipe_kernel<true> : packed_layer_kernel<true>;
            hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy, gz), dim3(256), 0, st, arg);
        } else {
            auto kern = cls[i] == PL_PIPE ? packed_layer_pipe_kernel<false> : packed_layer_kernel<false>;
            hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy, gz), dim3(256), 0, st, arg);
        }
        const int rc = check_launch(segmax ? "packed_layer_segmax" : "packed_layer");
        if (rc != PRCNN_OK || together) return rc;
    }
    return PRCNN_OK;
}

extern "C" int prcnn_packed_layer(const unsigned int *hdr, long rows, long max_tiles, int K, int N, int n_store, const float *A,
                                  long lda, const float *W, const float *bias, int relu, float *out, long ldo, void *stream)
{
    prcnn_layer_problem q = {};
    q.hdr = hdr; q.rows = rows; q.max_tiles = max_tiles; q.K = K; q.N = N; q.n_store = n_store; q.A = A; q.lda = lda; q.W = W;
    q.bias = bias; q.relu = relu; q.out = out; q.ldo = ldo;
    return prcnn_packed_layer_batch(1, &q, 0, stream);
}

extern "C" int prcnn_packed_layer_segmax(int b, int m, long max_tiles, int K, int N, const float *A, long lda, const float *W,
                                         const float *bias, const unsigned int *rowinfo, const int *tilecloud,
                                         const unsigned int *hdr, float *out, int out_stride, int out_col, int out_is_zero, void *stream)
{
    prcnn_layer_problem q = {};
    q.hdr = hdr; q.max_tiles = max_tiles; q.K = K; q.N = N; q.n_store = N; q.A = A; q.lda = lda; q.W = W; q.bias = bias; q.relu = 1;
    q.out = out; q.ldo = out_stride; q.b = b; q.m = m; q.rowinfo = rowinfo; q.tilecloud = tilecloud; q.out_col = out_col;
    q.out_is_zero = out_is_zero;
    return prcnn_packed_layer_batch(1, &q, 1, stream);
}


// The first layer of a feature-propagation module WITHOUT the interpolated tensor (round 3).  The reference interpolates the coarse
// level's features to the fine level, concatenates the skip features and applies the layer (pointnet2_modules.py:139-156):
// relu(W [interp(f) | skip] + b).  The layer is linear in front of its ReLU and the interpolation is a weighted sum, so
//     W_a interp(f) = interp(W_a f):   G = f @ W_a at the COARSE level (a quarter of the rows), then
//     out[r] = relu((skip[r] @ W_b + b) + ((w0 G[i0] + w1 G[i1]) + w2 G[i2]))     -- this entry, the addend inside the layer's epilogue.
// A (rows, K) = the skip features, W (K, N) = W_b, G (clouds * m_known rows, N wide, leading dimension ldg), idx / weight (rows, 3) from
// three_nn, row r in cloud r / n_per_cloud.  Not the reference's order of operations (the products are summed in another
// association): results agree to ~1e-7 relative, inside BASELINE's 1e-4 box tolerance; oracle/ext_cpu.py restates THIS order.
extern "C" int prcnn_packed_layer_interp(long rows, int K, int N, const float *A, long lda, const float *W, const float *bias, int relu,
                                         float *out, long ldo, int n_per_cloud, int m_known, const float *G, long ldg, const int *idx,
                                         const float *weight, void *stream)
{
    PRCNN_REQUIRE(rows >= 0 && K > 0 && N > 0 && K % 128 == 0 && N % 128 == 0, "packed_layer_interp: K=%d, N=%d must be multiples of 128", K, N);
    PRCNN_REQUIRE(lda >= K && lda % 4 == 0 && ldo >= N && ldo % 4 == 0 && ldg >= N && ldg % 4 == 0, "packed_layer_interp: bad leading dimensions");
    PRCNN_REQUIRE(n_per_cloud >= 1 && m_known >= 1 && rows % n_per_cloud == 0, "packed_layer_interp: rows must be whole clouds");
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(A && W && bias && out && G && idx && weight, "packed_layer_interp: null pointer");
    PRCNN_REQUIRE((((uintptr_t)A | (uintptr_t)out | (uintptr_t)G) & 15) == 0, "packed_layer_interp: 16-byte alignment required");
    const long tiles = (rows + PL_ROWS - 1) / PL_ROWS;
    PRCNN_REQUIRE(tiles <= 0x7fffffffL, "packed_layer_interp: too many rows");
    PLBatch bt;
    bt.p[0] = PLProblem{nullptr, rows, K, N, A, lda, W, bias, relu, out, ldo, nullptr, nullptr, 0, 0, N, 0,
                        G, idx, weight, n_per_cloud, m_known, ldg};
    // round 4: the persistent forms (weights resident / next tile fetched behind the running one) when there is more than one round of
    // workgroups to run; same epilogue, same bits
    const bool persist = true;
    const long items = tiles * (N / 128);
    hipStream_t st = (hipStream_t)stream;
    if (persist && K == 128 && stream_enabled() && items >= stream_min() && rows * lda * 4 < (1L << 31)) {
        const long cap = stream_cap() / (N / 128) > 0 ? stream_cap() / (N / 128) : 1;
        hipLaunchKernelGGL(packed_layer_stream_kernel, dim3((unsigned)(tiles < cap ? tiles : cap), N / 128), dim3(256), 0, st, rows, N, A, lda,
                           W, bias, relu, out, ldo, N, bt.p[0]);
        return check_launch("packed_layer_interp");
    }
    if (persist && K >= 256 && pipe_enabled() && items > stream_cap() && items >= persist_min()) {
        const long cap = stream_cap();
        unsigned int *ticket = next_ticket(st);
        if (!ticket) { set_error("packed_layer_interp: cannot set up the item ticket"); return PRCNN_ELAUNCH; }
        hipLaunchKernelGGL(packed_layer_persist_kernel<true>, dim3((unsigned)(items < cap ? items : cap)), dim3(256), 0, st, rows, K, N, A, lda,
                           W, bias, relu, out, ldo, ticket, bt.p[0]);
        return check_launch("packed_layer_interp");
    }
    auto kern = (K >= 256 && pipe_enabled()) ? packed_layer_pipe_kernel<false> : packed_layer_kernel<false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, N / 128, 1), dim3(256), 0, st, bt);
    return check_launch("packed_layer_interp");
}
