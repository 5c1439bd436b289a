// sa_mlp_fused.hip -- one kernel for a whole set-abstraction MLP on grouped points:
//
//     gather -> layer 1 (affine + ReLU) -> layer 2 (GEMM + bias + ReLU) -> layer 3 (GEMM + bias + ReLU)
//            -> max over the nsample neighbours
//
// i.e. QueryAndGroup's grouping + SharedMLP + F.max_pool2d of pointnet2_modules.py:37-53 for one
// scale, with NOTHING of the (B, C, npoint, nsample) activations ever written to HBM.  The reference
// streams that tensor through HBM ~6 times per layer; the library-GEMM path of net/fast_infer.py still
// writes and re-reads it once per layer (3.35 GB per layer at the RCNN SA1 size, B = 8).
//
// Mapping to CDNA4 (gfx950, wave64):
//   * tile = one query centre = 64 grouped rows; workgroup = 4 waves serving 8 tiles drawn from a ticket
//     counter, then retiring.
//   * layers 2 and 3 are dense f32 contractions -> v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered
//     fma chain).  Wave w owns output columns [32w, 32w+32) (+128 for the second half when C3 = 256)
//     for BOTH 32-row halves of the tile, so its B operand (the weight columns) never changes:
//     it is loaded ONCE into VGPRs (64 registers per 128x32 weight panel) and stays there for the whole
//     kernel.  Only the activations go through LDS.
//   * the K dimension is split between the two lane halves: lanes 0-31 supply k = s, lanes 32-63
//     k = s + 64 at MFMA step s.  A lane's four consecutive steps are then 16 contiguous bytes of
//     an activation row -> one ds_read_b128 feeds four MFMAs per row half; rows are padded to 132
//     floats so the 16-lane b128 groups cover all 64 banks (conflict-free).
//   * layer 1 is linear before its ReLU: W1 [f ; x - c] + b1 = (W1f f + b1) + W1x (x - c).  The per-point
//     part P arrives precomputed (one small GEMM over the N points of the cloud); the tile builder
//     gathers P rows with 16-byte loads, adds the 3-term coordinate part, applies ReLU and writes the
//     A operand of layer 2 straight into LDS.
//   * epilogues stay in registers: bias + ReLU on the accumulators; layer 2's result goes back to LDS
//     as layer 3's A operand; layer 3's result is max-reduced over the tile's 64 rows (16 accumulator
//     registers x 2 row halves per lane, then one cross-half shuffle) and 32 lanes store 128 bytes.
#include "common.hpp"
#include <map>
#include <mutex>
#include <stdlib.h>

namespace prcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SA_C = 128;          // C1 = C2 (layer-1 and layer-2 widths)
constexpr int SA_NS = 64;          // grouped rows per tile (= nsample)
constexpr int SA_LD = SA_C + 4;    // LDS row stride in floats
// A workgroup retires after this many tiles (its 128-256 KiB of weight registers are re-read from L2 by
// the next one): short-lived workgroups keep CUs turning over, so kernels of other streams (the
// geometry pass of the next batch) are not locked out for the whole launch as they would be by a
// persistent grid.
constexpr int SA_TILES_PER_WG = 8;

// C3 = 128: two workgroups per CU (2 waves per SIMD, <= 256 registers each) so that one workgroup's tile
// builder / epilogues run beside the other's MFMAs.  (C3 = 256 would need ~440 registers in this form -> one wave per
// SIMD; it has its own eight-wave kernel below.)
template <int C3>
__global__ __launch_bounds__(256, (C3 == 128 ? 2 : 1)) void sa_mlp_fused_kernel(
    int n, int m, long tiles, const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    const float4 *__restrict__ P /* (b,n,128) */, const float4 *__restrict__ wxyz /* (3,128) */,
    const int *__restrict__ idx /* (b,m,64) */, const float *__restrict__ w2t /* (128,128) k-major */,
    const float *__restrict__ b2, const float *__restrict__ w3t /* (128,C3) k-major */,
    const float *__restrict__ b3, float *__restrict__ out, int out_stride, int out_col,
    unsigned int *__restrict__ ticket, int tiles_per_wg)
{
    constexpr int NCT = C3 / 128;                 // column tiles of layer 3 per wave
    __shared__ float lds[2 * SA_NS * SA_LD + 4];  // A1 tile and Y1 tile, 64 x 132 each (+ two tile tickets)
    float *A1 = lds, *Y1 = lds + SA_NS * SA_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    // ---- weights -> registers, once.  B[k][col] for k = s + 64h, col = 32w + j (+128ct)
    float wf2[64], wf3[NCT][64];
#pragma unroll
    for (int s = 0; s < 64; ++s) wf2[s] = w2t[(long)(s + 64 * h) * SA_C + 32 * w + j];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int s = 0; s < 64; ++s) wf3[ct][s] = w3t[(long)(s + 64 * h) * C3 + 128 * ct + 32 * w + j];
    const float bias2 = b2[32 * w + j];
    float bias3[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bias3[ct] = b3[128 * ct + 32 * w + j];

    // tile builder: thread owns 16-byte chunk (tid & 31) of rows (tid >> 5) + 8i
    const int chunk = tid & 31;
    const float4 wx = wxyz[chunk], wy = wxyz[32 + chunk], wz = wxyz[64 + chunk];

    // Tiles are handed out through a ticket counter, not a static stride: a workgroup that starts late
    // (e.g. its CU was busy with another stream's kernel) simply takes fewer tiles instead of
    // stretching the whole launch.  The ticket and the neighbour indices of the NEXT tile are fetched one
    // tile ahead, so the builder's only exposed latency is the gather of the P rows themselves.
    unsigned int *slot = reinterpret_cast<unsigned int *>(lds + 2 * SA_NS * SA_LD);   // [2] double-buffered
    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = slot[0];
    int kidx[8];
    if (t < tiles) {
#pragma unroll
        for (int i = 0; i < 8; ++i) kidx[i] = idx[t * SA_NS + (tid >> 5) + 8 * i];
    }
    for (int served = 0; served < tiles_per_wg && t < tiles; ++served) {
        const long b = t / m;
        const float *ct3 = new_xyz + t * 3;
        const float cx = ct3[0], cy = ct3[1], cz = ct3[2];
        // draw the next ticket now; it is read after this tile's first barrier.  Not on the last tile this
        // workgroup serves: a ticket drawn and not served would be a tile nobody computes.
        const bool more = served + 1 < tiles_per_wg;
        if (tid == 0) slot[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
        // ---- layer 1 into LDS (fused multiply-adds: this arithmetic decides no index, the GEMM it feeds uses FMAs too)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 5) + 8 * i;
            const int k = kidx[i];
            const float *pt = xyz + (b * n + k) * 3;
            const float dx = pt[0] - cx, dy = pt[1] - cy, dz = pt[2] - cz;
            const float4 base = P[(b * n + k) * (SA_C / 4) + chunk];
            float4 v;
            v.x = fmaxf(fmaf(wz.x, dz, fmaf(wy.x, dy, fmaf(wx.x, dx, base.x))), 0.f);
            v.y = fmaxf(fmaf(wz.y, dz, fmaf(wy.y, dy, fmaf(wx.y, dx, base.y))), 0.f);
            v.z = fmaxf(fmaf(wz.z, dz, fmaf(wy.z, dy, fmaf(wx.z, dx, base.z))), 0.f);
            v.w = fmaxf(fmaf(wz.w, dz, fmaf(wy.w, dy, fmaf(wx.w, dx, base.w))), 0.f);
            *reinterpret_cast<float4 *>(A1 + row * SA_LD + 4 * chunk) = v;
        }
        __syncthreads();
        const long t_next = slot[(served + 1) & 1];
        if (t_next < tiles) {     // neighbour indices of the next tile, in flight during the MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) kidx[i] = idx[t_next * SA_NS + (tid >> 5) + 8 * i];
        }

        // ---- layer 2: Y1[64][32w..32w+32) = relu(A1 @ W2 + b2)
        {
            f32x16 acc0 = {0}, acc1 = {0};
            const float *a0p = A1 + j * SA_LD + 64 * h;
            const float *a1p = A1 + (32 + j) * SA_LD + 64 * h;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a0 = *reinterpret_cast<const float4 *>(a0p + 4 * g);
                const float4 a1 = *reinterpret_cast<const float4 *>(a1p + 4 * g);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wf2[4 * g + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wf2[4 * g + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wf2[4 * g + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, wf2[4 * g + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wf2[4 * g + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, wf2[4 * g + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wf2[4 * g + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, wf2[4 * g + 3], acc1, 0, 0, 0);
            }
            // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                Y1[row * SA_LD + 32 * w + j] = fmaxf(acc0[r] + bias2, 0.f);
                Y1[(32 + row) * SA_LD + 32 * w + j] = fmaxf(acc1[r] + bias2, 0.f);
            }
        }
        __syncthreads();

        // ---- layer 3 + max over the 64 rows
        {
            const float *a0p = Y1 + j * SA_LD + 64 * h;
            const float *a1p = Y1 + (32 + j) * SA_LD + 64 * h;
            f32x16 acc[NCT][2];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) { acc[ct][0] = (f32x16){0}; acc[ct][1] = (f32x16){0}; }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a0 = *reinterpret_cast<const float4 *>(a0p + 4 * g);
                const float4 a1 = *reinterpret_cast<const float4 *>(a1p + 4 * g);
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[q], wf3[ct][4 * g + q], acc[ct][0], 0, 0, 0);
                        acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[q], wf3[ct][4 * g + q], acc[ct][1], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float mx = fmaxf(acc[ct][0][0], acc[ct][1][0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(acc[ct][0][r], acc[ct][1][r]));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));           // the other 4-row groups live in the other lane half
                // relu(max + b) == max(relu(. + b)): bias and ReLU are monotone
                if (h == 0) out[t * out_stride + out_col + 128 * ct + 32 * w + j] = fmaxf(mx + bias3[ct], 0.f);
            }
        }
        // the next tile's builder overwrites A1 only; every wave has left layer 2 (barrier above), and
        // Y1 is rewritten only after the next tile's barriers, which all waves reach after layer 3.
        // Ticket slots alternate: slot[(served+1)&1] is written before this tile's first barrier and read
        // after it; it is rewritten two tiles later, after two more barriers.
        t = t_next;
    }
    if (tid == 0) ticket_release(ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}

// ---- C3 = 256 with EIGHT waves per workgroup (one workgroup per CU, two waves per SIMD).
// The four-wave form above needs ~440 registers per lane at C3 = 256 (three weight panels), i.e. one wave per SIMD, and
// everything that is not an MFMA (tile builder, epilogues, barriers) then runs with the matrix pipe idle.  Here the three
// panels are spread over eight waves: wave w owns column panel (w & 3) of layer 2 for row half (w >> 2), and column panel
// (w & 3) of column tile (w >> 2) of layer 3 for both row halves -- 64 + 64 weight registers per lane, two waves per SIMD,
// the non-MFMA phases are shared by twice as many threads.  Each SIMD still sees two independent accumulator chains.
__global__ __launch_bounds__(512, 1) void sa_mlp_fused256_kernel(
    int n, int m, long tiles, const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    const float4 *__restrict__ P, const float4 *__restrict__ wxyz, const int *__restrict__ idx,
    const float *__restrict__ w2t, const float *__restrict__ b2, const float *__restrict__ w3t /* (128,256) */,
    const float *__restrict__ b3, float *__restrict__ out, int out_stride, int out_col,
    unsigned int *__restrict__ ticket, int tiles_per_wg)
{
    __shared__ float lds[2 * SA_NS * SA_LD + 4];
    float *A1 = lds, *Y1 = lds + SA_NS * SA_LD;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, wp = w & 3, wg = w >> 2;     // panel, group (row half / column tile)
    const int j = lane & 31, h = lane >> 5;

    float wf2[64], wf3[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) wf2[s] = w2t[(long)(s + 64 * h) * SA_C + 32 * wp + j];
#pragma unroll
    for (int s = 0; s < 64; ++s) wf3[s] = w3t[(long)(s + 64 * h) * 256 + 128 * wg + 32 * wp + j];
    const float bias2 = b2[32 * wp + j], bias3 = b3[128 * wg + 32 * wp + j];

    // tile builder: thread owns 16-byte chunk (tid & 31) of rows (tid >> 5) + 16 i, i < 4
    const int chunk = tid & 31;
    const float4 wx = wxyz[chunk], wy = wxyz[32 + chunk], wz = wxyz[64 + chunk];

    unsigned int *slot = reinterpret_cast<unsigned int *>(lds + 2 * SA_NS * SA_LD);
    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = slot[0];
    int kidx[4];
    if (t < tiles) {
#pragma unroll
        for (int i = 0; i < 4; ++i) kidx[i] = idx[t * SA_NS + (tid >> 5) + 16 * i];
    }
    for (int served = 0; served < tiles_per_wg && t < tiles; ++served) {
        const long b = t / m;
        const float *ct3 = new_xyz + t * 3;
        const float cx = ct3[0], cy = ct3[1], cz = ct3[2];
        const bool more = served + 1 < tiles_per_wg;
        if (tid == 0) slot[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 5) + 16 * i;
            const int k = kidx[i];
            const float *pt = xyz + (b * n + k) * 3;
            const float dx = pt[0] - cx, dy = pt[1] - cy, dz = pt[2] - cz;
            const float4 base = P[(b * n + k) * (SA_C / 4) + chunk];
            float4 v;
            v.x = fmaxf(fmaf(wz.x, dz, fmaf(wy.x, dy, fmaf(wx.x, dx, base.x))), 0.f);
            v.y = fmaxf(fmaf(wz.y, dz, fmaf(wy.y, dy, fmaf(wx.y, dx, base.y))), 0.f);
            v.z = fmaxf(fmaf(wz.z, dz, fmaf(wy.z, dy, fmaf(wx.z, dx, base.z))), 0.f);
            v.w = fmaxf(fmaf(wz.w, dz, fmaf(wy.w, dy, fmaf(wx.w, dx, base.w))), 0.f);
            *reinterpret_cast<float4 *>(A1 + row * SA_LD + 4 * chunk) = v;
        }
        __syncthreads();
        const long t_next = slot[(served + 1) & 1];
        if (t_next < tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) kidx[i] = idx[t_next * SA_NS + (tid >> 5) + 16 * i];
        }

        // ---- layer 2: rows [32 wg, 32 wg + 32) x columns [32 wp, 32 wp + 32)
        {
            f32x16 acc = {0};
            const float *ap = A1 + (32 * wg + j) * SA_LD + 64 * h;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a = *reinterpret_cast<const float4 *>(ap + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wf2[4 * g + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wf2[4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wf2[4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wf2[4 * g + 3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wg + (r & 3) + 8 * (r >> 2) + 4 * h;
                Y1[row * SA_LD + 32 * wp + j] = fmaxf(acc[r] + bias2, 0.f);
            }
        }
        __syncthreads();

        // ---- layer 3: all 64 rows x columns [128 wg + 32 wp, +32), then max over the rows
        {
            const float *a0p = Y1 + j * SA_LD + 64 * h;
            const float *a1p = Y1 + (32 + j) * SA_LD + 64 * h;
            f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a0 = *reinterpret_cast<const float4 *>(a0p + 4 * g);
                const float4 a1 = *reinterpret_cast<const float4 *>(a1p + 4 * g);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wf3[4 * g + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wf3[4 * g + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wf3[4 * g + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, wf3[4 * g + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wf3[4 * g + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, wf3[4 * g + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wf3[4 * g + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, wf3[4 * g + 3], acc1, 0, 0, 0);
            }
            float mx = fmaxf(acc0[0], acc1[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(acc0[r], acc1[r]));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (h == 0) out[t * out_stride + out_col + 128 * wg + 32 * wp + j] = fmaxf(mx + bias3, 0.f);
        }
        t = t_next;
    }
    if (tid == 0) ticket_release(ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}

}  // namespace prcnn

namespace prcnn {
// Tile-ticket words: a ring of 128 records of 16 words per (device, stream) ([0] draw counter, [1] workgroups done, [2..9] per-XCD draw
// counters of the kernels that partition their tiles by XCD, common.hpp XcdTickets) (slot 6 of the scratch cache), zeroed when the stream's
// ring is created (scratch_for's `fresh` flag).  One launch uses one pair and leaves it at zero (common.hpp ticket_release: the launch's last workgroup resets it),
// so nothing is filled in front of a launch -- eagerly or inside a captured hipGraph, whose replays find the pair clean as well.
// (Round 1 zeroed one word per launch: a 5 us fill kernel in front of each ticketed launch; round 2 one memset per 256 launches
// and, under capture, a memset node per launch.)  Launches of other streams never touch the ring.
constexpr unsigned TICKET_RING = 128;
static std::mutex g_ticket_mu;
static std::map<std::pair<int, hipStream_t>, unsigned int> g_ticket_next;
unsigned int *next_ticket(hipStream_t st)
{
    bool fresh = false;
    unsigned int *ring = reinterpret_cast<unsigned int *>(scratch_for(st, 16 * TICKET_RING * sizeof(unsigned int), 6, &fresh));
    if (!ring) return nullptr;                 // (a first use under capture ends here: the stream needs its warm-up)
    unsigned int k;
    {
        std::lock_guard<std::mutex> lock(g_ticket_mu);
        k = g_ticket_next[std::make_pair(current_device(), st)]++;
    }
    // zeroed whenever the ring's memory is new to this (device, stream) -- first use, or an entry of the scratch table that changed
    // owner -- not keyed on the launch counter (a ring re-created after an eviction was never zeroed)
    if (fresh && hipMemsetAsync(ring, 0, 16 * TICKET_RING * sizeof(unsigned int), st) != hipSuccess) return nullptr;
    return ring + 16 * (k % TICKET_RING);
}
}  // namespace prcnn

using namespace prcnn;

// P (b,n,128) = features @ W1f^T + b1, wxyz (3,128); w2t (128,128), w3t (128,c3) stored k-major
// (row = input channel); out[(b*m rows)][out_col .. out_col + c3) with row stride out_stride.
// Supported shape: c1 = c2 = 128, c3 in {128, 256}, nsample = 64 (the RCNN SA1 / SA2 levels).
extern "C" int prcnn_sa_mlp_fused(int b, int n, int m, int nsample, int c1, int c2, int c3,
                                  const float *new_xyz, const float *xyz, const float *P, const float *wxyz,
                                  const int *idx, const float *w2t, const float *b2, const float *w3t,
                                  const float *b3, float *out, int out_stride, int out_col, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "sa_mlp_fused: bad sizes");
    PRCNN_REQUIRE(c1 == SA_C && c2 == SA_C && (c3 == 128 || c3 == 256) && nsample == SA_NS,
                  "sa_mlp_fused: unsupported shape c1=%d c2=%d c3=%d nsample=%d (need 128,128,128|256,64)", c1, c2, c3, nsample);
    PRCNN_REQUIRE(out_stride >= out_col + c3 && out_col >= 0, "sa_mlp_fused: bad output slice");
    const long tiles = (long)b * m;
    if (tiles == 0) return PRCNN_OK;
    PRCNN_REQUIRE(new_xyz && xyz && P && wxyz && idx && w2t && b2 && w3t && b3 && out, "sa_mlp_fused: null pointer");
    PRCNN_REQUIRE((((uintptr_t)P | (uintptr_t)wxyz) & 15) == 0, "sa_mlp_fused: 16-byte alignment required");
    static const int env_tiles = getenv("PRCNN_SA_TILES") ? atoi(getenv("PRCNN_SA_TILES")) : 0;
    const int per_wg = env_tiles > 0 ? env_tiles : SA_TILES_PER_WG;
    const int grid = (int)((tiles + per_wg - 1) / per_wg);
    unsigned int *ticket = next_ticket((hipStream_t)stream);
    if (!ticket) { set_error("sa_mlp_fused: cannot set up the tile ticket"); return PRCNN_ELAUNCH; }
    if (c3 == 128)
        hipLaunchKernelGGL(sa_mlp_fused_kernel<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, m, tiles, new_xyz,
                           xyz, (const float4 *)P, (const float4 *)wxyz, idx, w2t, b2, w3t, b3, out, out_stride, out_col, ticket, per_wg);
    else
        hipLaunchKernelGGL(sa_mlp_fused256_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, n, m, tiles, new_xyz,
                           xyz, (const float4 *)P, (const float4 *)wxyz, idx, w2t, b2, w3t, b3, out, out_stride, out_col, ticket, per_wg);
    return check_launch("sa_mlp_fused");
}
