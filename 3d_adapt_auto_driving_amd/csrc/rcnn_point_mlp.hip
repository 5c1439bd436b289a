// rcnn_point_mlp.hip -- the per-point MLP chain at the entrance of the RCNN (lib/net/rcnn_net.py:139-163:
// xyz_up_layer on [x', y', z', mask, depth], concatenation with the RPN features, merge_down_layer) fused with the
// per-point part of the first SA level's layer 1, in TWO hand-written f32 MFMA kernels:
//
//   rcnn_xyz_up_kernel :  X = relu(relu(in5 @ Wu1 + bu1) @ Wu2 + bu2)                       (rows x 128)
//   rcnn_merge_p_kernel:  P = relu([X | F] @ Wm + bm) @ Wp + bp = relu(X @ Wm_a + F @ Wm_b + bm) @ Wp + bp
//
// where F are the 128 RPN features of the pooled row and P is what csrc/sa_mlp_fused.hip gathers for SA1
// ("linear before ReLU": W1 [f ; x - c] + b1 = (W1f f + b1) + W1x (x - c)).  The library path did this with four
// GEMMs and a 420 MB concatenation, all of them persistent-grid kernels that stretch by 40-70 % when the geometry
// stream co-runs; here nothing but X goes through HBM between the pooled rows and P, `merged` never exists, and the
// workgroups are short-lived (ticketed tiles) like those of sa_mlp_fused.
//
// Same MFMA mapping as sa_mlp_fused.hip: tile = 64 rows, 4 waves, wave w owns output columns [32w, 32w+32), its
// 128 x 32 weight panels live in VGPRs (64 per panel), activations in LDS rows of 132 floats, K split across the
// two lane halves so one ds_read_b128 feeds four v_mfma_f32_32x32x2_f32.
#include "common.hpp"
#include <stdint.h>

namespace prcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PM_C = 128;
constexpr int PM_ROWS = 64;
constexpr int PM_LD = PM_C + 4;
constexpr int PM_TILES_PER_WG = 8;

// acc0 / acc1 (row halves 0-31 / 32-63) += tile[64][128] @ panel, panel register s = W[s + 64h][32w + j]
__device__ __forceinline__ void mfma_panel(const float *tile, const float (&wf)[64], f32x16 &acc0, f32x16 &acc1, int j, int h)
{
    const float *a0p = tile + j * PM_LD + 64 * h;
    const float *a1p = tile + (32 + j) * PM_LD + 64 * h;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float4 a0 = *reinterpret_cast<const float4 *>(a0p + 4 * g);
        const float4 a1 = *reinterpret_cast<const float4 *>(a1p + 4 * g);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wf[4 * g + 0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wf[4 * g + 0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wf[4 * g + 1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, wf[4 * g + 1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wf[4 * g + 2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, wf[4 * g + 2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wf[4 * g + 3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, wf[4 * g + 3], acc1, 0, 0, 0);
    }
}

__device__ __forceinline__ void load_panel(float (&wf)[64], const float *__restrict__ wt, int w, int j, int h)
{
#pragma unroll
    for (int s = 0; s < 64; ++s) wf[s] = wt[(long)(s + 64 * h) * PM_C + 32 * w + j];
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- kernel 1: the two xyz_up layers.  rows (R, ld): columns 0..7 = [x', y', z', mask, depth, 0, 0, 0]
__global__ __launch_bounds__(256, 2) void rcnn_xyz_up_kernel(
    long tiles, int ld, const float *__restrict__ rows, const float4 *__restrict__ wu1 /* (8,128) k-major */,
    const float4 *__restrict__ bu1, const float *__restrict__ wu2 /* (128,128) k-major */, const float *__restrict__ bu2,
    float *__restrict__ xout /* (R,128) */, unsigned int *__restrict__ ticket)
{
    __shared__ float lds[PM_ROWS * PM_LD + 4];
    float *A1 = lds;
    unsigned int *slot = reinterpret_cast<unsigned int *>(lds + PM_ROWS * PM_LD);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
    float wf[64];
    load_panel(wf, wu2, w, j, h);
    const float bias2 = bu2[32 * w + j];
    const int chunk = tid & 31;
    float4 k1[5];                                                   // layer-1 weights of this thread's 4 channels
#pragma unroll
    for (int k = 0; k < 5; ++k) k1[k] = wu1[k * 32 + chunk];
    const float4 b1 = bu1[chunk];

    // tickets as in sa_mlp_fused.hip: the next tile's ticket is drawn one tile ahead and never on the last tile served
    unsigned int *slot2 = slot;                                     // [2], double-buffered
    if (tid == 0) slot2[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = slot2[0];
    for (int served = 0; served < PM_TILES_PER_WG && t < tiles; ++served) {
        const bool more = served + 1 < PM_TILES_PER_WG;
        if (tid == 0) slot2[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 5) + 8 * i;
            const float *src = rows + (t * PM_ROWS + row) * ld;
            const float4 p0 = *reinterpret_cast<const float4 *>(src);
            const float d = src[4];
            float4 v;
            v.x = fmaxf(b1.x + k1[0].x * p0.x + k1[1].x * p0.y + k1[2].x * p0.z + k1[3].x * p0.w + k1[4].x * d, 0.f);
            v.y = fmaxf(b1.y + k1[0].y * p0.x + k1[1].y * p0.y + k1[2].y * p0.z + k1[3].y * p0.w + k1[4].y * d, 0.f);
            v.z = fmaxf(b1.z + k1[0].z * p0.x + k1[1].z * p0.y + k1[2].z * p0.z + k1[3].z * p0.w + k1[4].z * d, 0.f);
            v.w = fmaxf(b1.w + k1[0].w * p0.x + k1[1].w * p0.y + k1[2].w * p0.z + k1[3].w * p0.w + k1[4].w * d, 0.f);
            *reinterpret_cast<float4 *>(A1 + row * PM_LD + 4 * chunk) = v;
        }
        __syncthreads();
        const long t_next = slot2[(served + 1) & 1];
        f32x16 acc0 = {0}, acc1 = {0};
        mfma_panel(A1, wf, acc0, acc1, j, h);
        float *o = xout + t * PM_ROWS * PM_C + 32 * w + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, h);
            o[(long)row * PM_C] = fmaxf(acc0[r] + bias2, 0.f);
            o[(long)(32 + row) * PM_C] = fmaxf(acc1[r] + bias2, 0.f);
        }
        __syncthreads();                                            // every wave has read A1 before the next builder writes it
        t = t_next;
    }
}

// ---- kernel 2: merge_down (K = 256 as two 128-panels) + the per-point part of SA1's first layer
__global__ __launch_bounds__(256, 1) void rcnn_merge_p_kernel(
    long tiles, int ld, int fcol, const float *__restrict__ xfeat /* (R,128) */, const float *__restrict__ rows /* (R,ld) */,
    const float *__restrict__ wma, const float *__restrict__ wmb, const float *__restrict__ bm,
    const float *__restrict__ wp, const float *__restrict__ bp, float *__restrict__ pout /* (R,128) */,
    unsigned int *__restrict__ ticket)
{
    extern __shared__ float dyn[];                                  // X tile, F tile, Y tile: 3 x 64 x 132 floats (+ ticket slot)
    float *X = dyn, *F = dyn + PM_ROWS * PM_LD, *Y = dyn + 2 * PM_ROWS * PM_LD;
    unsigned int *slot = reinterpret_cast<unsigned int *>(dyn + 3 * PM_ROWS * PM_LD);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
    float wa[64], wb[64], wq[64];
    load_panel(wa, wma, w, j, h);
    load_panel(wb, wmb, w, j, h);
    load_panel(wq, wp, w, j, h);
    const float biasm = bm[32 * w + j], biasp = bp[32 * w + j];
    const int chunk = tid & 31;

    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = slot[0];
    for (int served = 0; served < PM_TILES_PER_WG && t < tiles; ++served) {
        const bool more = served + 1 < PM_TILES_PER_WG;
        if (tid == 0) slot[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 5) + 8 * i;
            const long g = t * PM_ROWS + row;
            *reinterpret_cast<float4 *>(X + row * PM_LD + 4 * chunk) =
                *reinterpret_cast<const float4 *>(xfeat + g * PM_C + 4 * chunk);
            *reinterpret_cast<float4 *>(F + row * PM_LD + 4 * chunk) =
                *reinterpret_cast<const float4 *>(rows + g * ld + fcol + 4 * chunk);
        }
        __syncthreads();
        const long t_next = slot[(served + 1) & 1];
        {
            f32x16 acc0 = {0}, acc1 = {0};
            mfma_panel(X, wa, acc0, acc1, j, h);
            mfma_panel(F, wb, acc0, acc1, j, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, h);
                Y[row * PM_LD + 32 * w + j] = fmaxf(acc0[r] + biasm, 0.f);
                Y[(32 + row) * PM_LD + 32 * w + j] = fmaxf(acc1[r] + biasm, 0.f);
            }
        }
        __syncthreads();
        {
            f32x16 acc0 = {0}, acc1 = {0};
            mfma_panel(Y, wq, acc0, acc1, j, h);
            float *o = pout + t * PM_ROWS * PM_C + 32 * w + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, h);
                o[(long)row * PM_C] = acc0[r] + biasp;              // linear: SA1 adds the coordinate part, then ReLU
                o[(long)(32 + row) * PM_C] = acc1[r] + biasp;
            }
        }
        // X / F are rewritten by the next builder: every wave passed the barrier after the first layer; Y is rewritten
        // only after the next tile's barrier, which all waves reach after this layer
        t = t_next;
    }
}

unsigned int *next_ticket(hipStream_t st);    // sa_mlp_fused.hip

}  // namespace prcnn

using namespace prcnn;

// rows (r, ld) f32 = the pooled RCNN input rows [x',y',z',mask,depth,0,0,0 | 128 features at column fcol] (r % 64 == 0);
// wu1 (8,128), wu2 (128,128), wm (256,128) = [Wm_a ; Wm_b], wp (128,128): k-major, BN folded; xfeat (r,128) scratch for X;
// p (r,128) = relu([X | F] wm + bm) wp + bp.
extern "C" int prcnn_rcnn_point_mlp(long r, int ld, int fcol, const float *rows, const float *wu1, const float *bu1,
                                    const float *wu2, const float *bu2, const float *wm, const float *bm, const float *wp,
                                    const float *bp, float *xfeat, float *p, void *stream)
{
    PRCNN_REQUIRE(r >= 0 && r % PM_ROWS == 0, "rcnn_point_mlp: %ld rows is not a multiple of %d", r, PM_ROWS);
    PRCNN_REQUIRE(ld >= 8 && ld % 4 == 0 && fcol >= 8 && fcol % 4 == 0 && fcol + PM_C <= ld,
                  "rcnn_point_mlp: bad row layout ld=%d fcol=%d", ld, fcol);
    if (r == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rows && wu1 && bu1 && wu2 && bu2 && wm && bm && wp && bp && xfeat && p, "rcnn_point_mlp: null pointer");
    PRCNN_REQUIRE((((uintptr_t)rows | (uintptr_t)wu1 | (uintptr_t)bu1 | (uintptr_t)xfeat | (uintptr_t)p) & 15) == 0,
                  "rcnn_point_mlp: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    const long tiles = r / PM_ROWS;
    const int grid = (int)((tiles + PM_TILES_PER_WG - 1) / PM_TILES_PER_WG);
    unsigned int *t1 = next_ticket(st), *t2 = next_ticket(st);
    if (!t1 || !t2) { set_error("rcnn_point_mlp: cannot set up the tile tickets"); return PRCNN_ELAUNCH; }
    hipLaunchKernelGGL(rcnn_xyz_up_kernel, dim3(grid), dim3(256), 0, st, tiles, ld, rows, (const float4 *)wu1,
                       (const float4 *)bu1, wu2, bu2, xfeat, t1);
    int rc = check_launch("rcnn_point_mlp(xyz_up)");
    if (rc != PRCNN_OK) return rc;
    const size_t lds = (size_t)(3 * PM_ROWS * PM_LD + 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)rcnn_merge_p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(rcnn_merge_p_kernel, dim3(grid), dim3(256), lds, st, tiles, ld, fcol, xfeat, rows, wm,
                       wm + (size_t)PM_C * PM_C, bm, wp, bp, p, t2);
    return check_launch("rcnn_point_mlp(merge)");
}
