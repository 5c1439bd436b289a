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

// ---- one 128-wide layer over 64-row tiles:  out = act(A0 @ W0 [+ A1 @ W1] + bias)
//   PRO = 0: A0 = src0 rows (128 floats at column col0, row stride ld0);   PRO = 1: A0 = relu(in5 @ Wu1 + bu1) computed by
//   the builder from columns 0..4 of src0 (the first xyz_up layer: K = 5 is VALU work);  NPANEL = 2 adds A1 = src1 rows.
// The inputs of the NEXT tile are fetched into registers right after this tile's builder, so the HBM latency runs
// behind the MFMAs; barriers order LDS only (lds_barrier), they do not drain those loads.  The accumulators are staged
// through LDS (tile 0 is dead by then) so that every output row leaves as 32 x 16-byte lanes.
template <int NPANEL, int PRO, bool RELU>
__global__ __launch_bounds__(256, 2) void rows_layer_kernel(
    long tiles, int per_wg, const float *__restrict__ src0, int ld0, int col0, const float *__restrict__ src1, int ld1, int col1,
    const float4 *__restrict__ wu1, const float4 *__restrict__ bu1, const float *__restrict__ w0, const float *__restrict__ w1,
    const float *__restrict__ bias, float *__restrict__ out, unsigned int *__restrict__ ticket,
    const int *__restrict__ tilemap, const unsigned int *__restrict__ ntiles_dev)
{
    // tilemap (optional): ticket j serves tile tilemap[j], j < *ntiles_dev -- only the tiles that hold DISTINCT pooled rows
    // (prcnn_pooled_tiles); the grid is still sized for all tiles and surplus workgroups draw a ticket and leave.
    if (ntiles_dev) tiles = (long)*ntiles_dev;
#define TILE_OF(tk) (tilemap ? (long)tilemap[tk] : (long)(tk))
    __shared__ float lds[NPANEL * PM_ROWS * PM_LD + 4];
    float *T0 = lds, *T1 = lds + (NPANEL - 1) * PM_ROWS * PM_LD;
    unsigned int *slot = reinterpret_cast<unsigned int *>(lds + NPANEL * PM_ROWS * PM_LD);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
    const int chunk = tid & 31, r0 = tid >> 5;
    float wa[64], wb[NPANEL == 2 ? 64 : 1];
    load_panel(wa, w0, w, j, h);
    if constexpr (NPANEL == 2) load_panel(wb, w1, w, j, h);
    const float bcol = bias[32 * w + j];
    float4 k1[PRO ? 5 : 1], b1 = {0, 0, 0, 0};
    if constexpr (PRO == 1) {
#pragma unroll
        for (int k = 0; k < 5; ++k) k1[k] = wu1[k * 32 + chunk];
        b1 = bu1[chunk];
    }

    constexpr bool PF = (NPANEL == 1);   // two panels: 128 weight registers leave no room for a 64-register prefetch
    float4 p0[8], p1[NPANEL == 2 ? 8 : 1];
    float pd[PRO ? 8 : 1];
#define ROWS_FETCH(tile)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                   \
        const long g_ = (tile) * PM_ROWS + r0 + 8 * i;                                                                \
        if constexpr (PRO == 1) {                                                                                     \
            p0[i] = *reinterpret_cast<const float4 *>(src0 + g_ * ld0);                                               \
            pd[i] = src0[g_ * ld0 + 4];                                                                               \
        } else {                                                                                                      \
            p0[i] = *reinterpret_cast<const float4 *>(src0 + g_ * ld0 + col0 + 4 * chunk);                            \
        }                                                                                                             \
        if constexpr (NPANEL == 2) p1[i] = *reinterpret_cast<const float4 *>(src1 + g_ * ld1 + col1 + 4 * chunk);     \
    }

    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    lds_barrier();
    long t = slot[0];
    if (t >= tiles) { if (tid == 0) ticket_release(ticket); return; }
    long tt = TILE_OF(t);
    if (PF) { ROWS_FETCH(tt) }
    for (int served = 0; served < per_wg && t < tiles; ++served) {
        const bool more = served + 1 < per_wg;
        if (tid == 0) slot[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
        if (!PF) { ROWS_FETCH(tt) }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = r0 + 8 * i;
            float4 v = p0[i];
            if constexpr (PRO == 1) {
                const float4 q = p0[i];
                const float d = pd[i];
                v = relu4(pk_fma4(k1[4], d, pk_fma4(k1[3], q.w, pk_fma4(k1[2], q.z, pk_fma4(k1[1], q.y, pk_fma4(k1[0], q.x, b1))))));
            }
            *reinterpret_cast<float4 *>(T0 + row * PM_LD + 4 * chunk) = v;
            if constexpr (NPANEL == 2) *reinterpret_cast<float4 *>(T1 + row * PM_LD + 4 * chunk) = p1[i];
        }
        lds_barrier();
        const long t_next = slot[(served + 1) & 1];
        const long tt_next = t_next < tiles ? TILE_OF(t_next) : 0;
        if (PF && t_next < tiles) { ROWS_FETCH(tt_next) }           // in flight during the MFMAs below
        f32x16 acc0 = {0}, acc1 = {0};
        mfma_panel(T0, wa, acc0, acc1, j, h);
        if constexpr (NPANEL == 2) mfma_panel(T1, wb, acc0, acc1, j, h);
        lds_barrier();                                              // every wave has finished reading the tiles
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, h);
            const float v0 = acc0[r] + bcol, v1 = acc1[r] + bcol;
            T0[row * PM_LD + 32 * w + j] = RELU ? fmaxf(v0, 0.f) : v0;
            T0[(32 + row) * PM_LD + 32 * w + j] = RELU ? fmaxf(v1, 0.f) : v1;
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = r0 + 8 * i;
            *reinterpret_cast<float4 *>(out + (tt * PM_ROWS + row) * PM_C + 4 * chunk) =
                *reinterpret_cast<const float4 *>(T0 + row * PM_LD + 4 * chunk);
        }
        lds_barrier();                                              // the next builder overwrites T0
        t = t_next;
        tt = tt_next;
    }
#undef ROWS_FETCH
#undef TILE_OF
    if (tid == 0) ticket_release(ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}

// ---- one 128 -> 128 layer over a LIST of rows (round 5): out[r] = act(src[r] @ W + bias) for r = rowmap[0 .. hdr[1]), 64 list entries
// per tile whichever rows they are; rows that are not listed are neither read nor written.  The per-point part of the RCNN's second SA
// level runs over the level-1 centres that are their own representatives only (47 of 128 per RoI on the uniform scene: the others
// copy an earlier centre and no row list names them).  Tiles strided over the workgroups; the NEXT tile's rows (their numbers were
// fetched one tile earlier still) arrive behind the MFMAs.  Per row the arithmetic of rows_layer_kernel / packed_layer_stream_kernel.
template <bool RELU>
__global__ __launch_bounds__(256, 2) void rows_layer_list_kernel(const float *__restrict__ src, int ld, int col, const float *__restrict__ w0,
                                                                 const float *__restrict__ bias, float *__restrict__ out,
                                                                 const int *__restrict__ rowmap, const unsigned int *__restrict__ hdr)
{
    __shared__ float T0[PM_ROWS * PM_LD];
    const long nrows = (long)hdr[1];
    const long tiles = (nrows + PM_ROWS - 1) / PM_ROWS;
    long t = blockIdx.x;
    if (t >= tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
    const int chunk = tid & 31, r0 = tid >> 5;
    float wa[64];
    load_panel(wa, w0, w, j, h);
    const float bcol = bias[32 * w + j];
    const long step = gridDim.x;
    int cur[8], nxt[8];
    float4 p0[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        cur[i] = rowmap[min(t * PM_ROWS + r0 + 8 * i, nrows - 1)];
        nxt[i] = t + step < tiles ? rowmap[min((t + step) * PM_ROWS + r0 + 8 * i, nrows - 1)] : 0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) p0[i] = *reinterpret_cast<const float4 *>(src + (long)cur[i] * ld + col + 4 * chunk);
    for (; t < tiles; t += step) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(T0 + (r0 + 8 * i) * PM_LD + 4 * chunk) = p0[i];
        lds_barrier();
        int nn[8];
        if (t + step < tiles) {                                     // the next tile's rows, the numbers of the one after it: behind the MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) p0[i] = *reinterpret_cast<const float4 *>(src + (long)nxt[i] * ld + col + 4 * chunk);
#pragma unroll
            for (int i = 0; i < 8; ++i) nn[i] = t + 2 * step < tiles ? rowmap[min((t + 2 * step) * PM_ROWS + r0 + 8 * i, nrows - 1)] : 0;
        }
        f32x16 acc0 = {0}, acc1 = {0};
        mfma_panel(T0, wa, acc0, acc1, j, h);
        lds_barrier();                                              // every wave has finished reading the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, h);
            const float v0 = acc0[r] + bcol, v1 = acc1[r] + bcol;
            T0[row * PM_LD + 32 * w + j] = RELU ? fmaxf(v0, 0.f) : v0;
            T0[(32 + row) * PM_LD + 32 * w + j] = RELU ? fmaxf(v1, 0.f) : v1;
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<float4 *>(out + (long)cur[i] * PM_C + 4 * chunk) = *reinterpret_cast<const float4 *>(T0 + (r0 + 8 * i) * PM_LD + 4 * chunk);
        lds_barrier();                                              // the next builder overwrites T0
        if (t + step < tiles) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { cur[i] = nxt[i]; nxt[i] = nn[i]; }
        }
    }
}

unsigned int *next_ticket(hipStream_t st);    // sa_mlp_fused.hip

// ---- the whole entrance chain over a tile that stays in LDS (round 2, after the stage stamps of csrc/rpn_tail.hip) -----------
// Per 64-row tile: builder (first xyz_up layer, K = 5, VALU; the tile's 128 RPN features go to the second LDS tile) ->
// xyz_up layer 2 -> merge_down over [xfeat | feats] (two panels into one accumulator) -> SA1's per-point part; only P leaves
// the CU.  Four panel stages per tile, each one's 128 x 32 weight slice per wave streaming in behind the MFMAs of the stage
// before it (mfma_stream.hpp); persistent workgroups draw the live tiles from a ticket counter.  Arithmetic per row = the three
// rows_layer_kernel launches above, bit for bit (same builder chain, same panel order, same epilogues).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RT_LD = PM_LD;
}  // namespace prcnn
#include "mfma_stream.hpp"
namespace prcnn {

__global__ __launch_bounds__(256, 2) void rcnn_entrance_kernel(
    long tiles, const float *__restrict__ rows, int ld, int fcol, const float4 *__restrict__ wu1, const float4 *__restrict__ bu1,
    const float *__restrict__ wu2, const float *__restrict__ bu2, const float *__restrict__ wm, const float *__restrict__ bm,
    const float *__restrict__ wp, const float *__restrict__ bp, float *__restrict__ p_out, unsigned int *__restrict__ ticket,
    const int *__restrict__ tilemap, const unsigned int *__restrict__ ntiles_dev, const int *__restrict__ rowmap)
{
    // rowmap != NULL (round 5, prcnn_rcnn_point_mlp_rows): tile t = the rows rowmap[64 t .. 64 t + 63] of `rows` / `p_out` -- the
    // DISTINCT pooled rows of all RoIs back to back (prcnn_pooled_rows), ntiles_dev[1] of them; the last tile's missing entries repeat
    // the list's last row (computed and stored again: the same values to the same place).  With whole 64-row tiles per RoI
    // (tilemap) a RoI's 47 distinct rows of 512 filled a tile of 64; per row the arithmetic does not depend on the tile: same bits.
    long nrows = 0;
    if (rowmap) { nrows = (long)ntiles_dev[1]; tiles = (nrows + PM_ROWS - 1) / PM_ROWS; }
    else if (ntiles_dev) tiles = (long)*ntiles_dev;
    __shared__ float T0[PM_ROWS * PM_LD];
    __shared__ float T1[PM_ROWS * PM_LD];
    __shared__ unsigned int slot[2];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    const unsigned int lane_off = ((unsigned int)(64 * h) * 128u + (unsigned int)(32 * w + j)) * 4u;
    const __amdgpu_buffer_rsrc_t rs_u2 = __builtin_amdgcn_make_buffer_rsrc((void *)wu2, 0, 128 * 128 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void *)wm, 0, 256 * 128 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, 128 * 128 * 4, 0x00020000);
    const float bias_u2 = bu2[32 * w + j], bias_m = bm[32 * w + j], bias_p = bp[32 * w + j];
    float4 k1[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) k1[k] = wu1[k * 32 + chunk];
    const float4 b1 = bu1[chunk];

    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = __builtin_amdgcn_readfirstlane((int)slot[0]);
    float wa[64], wb[64];
    f32x16 acc0, acc1;
    RT_LOAD_W(wa, rs_u2, 0)
    int ridx[8], nidx[8];                                      // this thread's 8 rows of the running tile / of the next one
#pragma unroll
    for (int i = 0; i < 8; ++i) ridx[i] = (rowmap && t < tiles) ? rowmap[min(t * PM_ROWS + r0 + 8 * i, nrows - 1)] : 0;
    for (unsigned int served = 0; t < tiles; ++served) {
        if (!rowmap) {
            const long tt = tilemap ? (long)tilemap[t] : t;
#pragma unroll
            for (int i = 0; i < 8; ++i) ridx[i] = (int)(tt * PM_ROWS + r0 + 8 * i);
        }
        // ---- builder: layer 1 of xyz_up from the 5 input columns -> T0, the row's RPN features -> T1
        {
            float4 q[8], f[8];
            float d[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float *row = rows + (long)ridx[i] * (long)ld;
                q[i] = *reinterpret_cast<const float4 *>(row);
                d[i] = row[4];
                f[i] = *reinterpret_cast<const float4 *>(row + fcol + 4 * chunk);
            }
            if (tid == 0) slot[(served + 1) & 1] = atomicAdd(ticket, 1u);     // the next tile's ticket rides behind the loads
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float4 v;
                v = relu4(pk_fma4(k1[4], d[i], pk_fma4(k1[3], q[i].w, pk_fma4(k1[2], q[i].z, pk_fma4(k1[1], q[i].y, pk_fma4(k1[0], q[i].x, b1))))));
                *reinterpret_cast<float4 *>(T0 + (r0 + 8 * i) * PM_LD + 4 * chunk) = v;
                *reinterpret_cast<float4 *>(T1 + (r0 + 8 * i) * PM_LD + 4 * chunk) = f[i];
            }
        }
        RT_VM_DRAIN                                            // (also: this tile's first panel, fetched during the last stage)
        lds_barrier();
        if (rowmap) {                                          // the next tile's row numbers: in flight behind the first stage
            const long tq = (long)__builtin_amdgcn_readfirstlane((int)slot[(served + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 8; ++i) nidx[i] = tq < tiles ? rowmap[min(tq * PM_ROWS + r0 + 8 * i, nrows - 1)] : 0;
        }
        // ---- xyz_up layer 2 (wa) while merge panel a (wb) comes in
        RT_STAGE(T0, wa, wb, rs_m, 0, true)
        lds_barrier();                                         // every wave has read the layer-1 rows
        RT_EPILOGUE(T0, bias_u2, true)
        lds_barrier();
        // ---- merge_down: [xfeat | feats] as two panels into one accumulator
        RT_VM_DRAIN
        RT_STAGE(T0, wb, wa, rs_m, 128, true)
        RT_VM_DRAIN
        RT_STAGE(T1, wa, wb, rs_p, 0, false)
        lds_barrier();                                         // every wave has read xfeat
        RT_EPILOGUE(T0, bias_m, true)
        lds_barrier();
        // ---- SA1's per-point part (no activation) while the next tile's first panel (wa) comes in
        RT_VM_DRAIN
        RT_STAGE(T0, wb, wa, rs_u2, 0, true)
        RT_EPILOGUE(T1, bias_p, false)
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = r0 + 8 * i;
            *reinterpret_cast<f32x4 *>(p_out + (long)ridx[i] * PM_C + 4 * chunk) = *reinterpret_cast<const f32x4 *>(T1 + row * PM_LD + 4 * chunk);
        }
        const long tn = __builtin_amdgcn_readfirstlane((int)slot[(served + 1) & 1]);
        lds_barrier();                                         // T1 is free for the next builder
        t = tn;
        if (rowmap) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ridx[i] = nidx[i];
        }
    }
    if (tid == 0) ticket_release(ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}

// cnt[c] distinct rows of cloud c (rows_per_cloud rows each, a multiple of 64) -> the list of 64-row tiles that hold
// them, in cloud order: tilemap[j], j < hdr[0].  One workgroup, clouds in chunks of 1024 with a running offset.
__global__ __launch_bounds__(1024) void pooled_tiles_kernel(int clouds, int rows_per_cloud, const int *__restrict__ cnt,
                                                            int *__restrict__ tilemap, unsigned int *__restrict__ hdr)
{
    __shared__ int part[1024];
    __shared__ int s_run;
    const int tid = threadIdx.x, per = rows_per_cloud / PM_ROWS;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int c0 = 0; c0 < clouds; c0 += 1024) {
        const int c = c0 + tid;
        const int live = c < clouds ? min(per, (max(cnt[c], 1) + PM_ROWS - 1) / PM_ROWS) : 0;
        part[tid] = live;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const int base = s_run + part[tid] - live;
        for (int q = 0; q < live; ++q) tilemap[base + q] = c * per + q;
        __syncthreads();
        if (tid == 1023) s_run += part[1023];
        __syncthreads();
    }
    if (tid == 0) hdr[0] = (unsigned int)s_run;
}

// cnt[c] distinct rows of cloud c -> the list of those rows (row c * rows_per_cloud + j, j < max(cnt[c], 1)) of ALL clouds back to back:
// rowmap[0 .. hdr[1]).  A wave per cloud draws its block from the list's row counter hdr[1] (zero on entry): the order of the clouds in
// the list is the counter's, every row of the input is computed on its own.
__global__ __launch_bounds__(256) void pooled_rows_kernel(int clouds, int rows_per_cloud, const int *__restrict__ cnt, int *__restrict__ rowmap,
                                                          unsigned int *__restrict__ hdr)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= clouds) return;
    const int n = min(max(cnt[c], 1), rows_per_cloud);
    int base = 0;
    if (lane == 0) base = (int)atomicAdd(&hdr[1], (unsigned int)n);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int j = lane; j < n; j += 64) rowmap[base + j] = c * rows_per_cloud + j;
}

}  // namespace prcnn

using namespace prcnn;

// rows (r, ld) f32 = the pooled RCNN input rows [x',y',z',mask,depth,0,0,0 | 128 features at column fcol] (r % 64 == 0);
// wu1 (8,128), wu2 (128,128), wm (256,128) = [Wm_a ; Wm_b], wp (128,128): k-major, BN folded.  Three launches of
// rows_layer_kernel:   xfeat  = relu(relu(in5 wu1 + bu1) wu2 + bu2)            (xyz_up_layer)
//                      merged = relu(xfeat wm_a + feats wm_b + bm)               (concat + merge_down_layer)
//                      p      = merged wp + bp                                   (per-point part of SA1's layer 1)
// xfeat, merged, p: (r,128) each, caller-allocated.
extern "C" int prcnn_rcnn_point_mlp(long r, int ld, int fcol, const float *rows, const float *wu1, const float *bu1,
                                    const float *wu2, const float *bu2, const float *wm, const float *bm, const float *wp,
                                    const float *bp, float *xfeat, float *merged, float *p, const int *tilemap,
                                    const unsigned int *ntiles, void *stream)
{
    PRCNN_REQUIRE((tilemap == nullptr) == (ntiles == nullptr), "rcnn_point_mlp: tilemap and ntiles go together");
    PRCNN_REQUIRE(r >= 0 && r % PM_ROWS == 0, "rcnn_point_mlp: %ld rows is not a multiple of %d", r, PM_ROWS);
    PRCNN_REQUIRE(ld >= 8 && ld % 4 == 0 && fcol >= 8 && fcol % 4 == 0 && fcol + PM_C <= ld,
                  "rcnn_point_mlp: bad row layout ld=%d fcol=%d", ld, fcol);
    if (r == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rows && wu1 && bu1 && wu2 && bu2 && wm && bm && wp && bp && p, "rcnn_point_mlp: null pointer");
    PRCNN_REQUIRE((xfeat == nullptr) == (merged == nullptr), "rcnn_point_mlp: xfeat and merged are both given or both NULL");
    PRCNN_REQUIRE((((uintptr_t)rows | (uintptr_t)wu1 | (uintptr_t)bu1 | (uintptr_t)xfeat | (uintptr_t)merged | (uintptr_t)p) & 15) == 0,
                  "rcnn_point_mlp: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    const long tiles = r / PM_ROWS;
    if (!xfeat) {
        // only P is wanted: the whole chain in one kernel, the tile never leaves LDS
        PRCNN_REQUIRE((((uintptr_t)wu2 | (uintptr_t)wm | (uintptr_t)wp) & 15) == 0, "rcnn_point_mlp: 16-byte alignment required");
        unsigned int *tk = next_ticket(st);
        if (!tk) { set_error("rcnn_point_mlp: cannot set up the tile ticket"); return PRCNN_ELAUNCH; }
        const long grid = tiles < mfma_grid_cap() ? tiles : mfma_grid_cap();
        hipLaunchKernelGGL(rcnn_entrance_kernel, dim3((unsigned)grid), dim3(256), 0, st, tiles, rows, ld, fcol, (const float4 *)wu1,
                           (const float4 *)bu1, wu2, bu2, wm, bm, wp, bp, p, tk, tilemap, ntiles, nullptr);
        return check_launch("rcnn_point_mlp(fused)");
    }
    // one generation of workgroups when the launch is short (tiles / slots per workgroup, tickets balance the rest);
    // at least PM_TILES_PER_WG so that long launches still turn workgroups over
    const long slots = 512;
    int per_wg = (int)((tiles + slots - 1) / slots);
    if (per_wg < PM_TILES_PER_WG) per_wg = PM_TILES_PER_WG;
    const int grid = (int)((tiles + per_wg - 1) / per_wg);
    unsigned int *t1 = next_ticket(st), *t2 = next_ticket(st), *t3 = next_ticket(st);
    if (!t1 || !t2 || !t3) { set_error("rcnn_point_mlp: cannot set up the tile tickets"); return PRCNN_ELAUNCH; }
    hipLaunchKernelGGL((rows_layer_kernel<1, 1, true>), dim3(grid), dim3(256), 0, st, tiles, per_wg, rows, ld, 0, nullptr, 0, 0,
                       (const float4 *)wu1, (const float4 *)bu1, wu2, nullptr, bu2, xfeat, t1, tilemap, ntiles);
    int rc = check_launch("rcnn_point_mlp(xyz_up)");
    if (rc != PRCNN_OK) return rc;
    hipLaunchKernelGGL((rows_layer_kernel<2, 0, true>), dim3(grid), dim3(256), 0, st, tiles, per_wg, xfeat, PM_C, 0, rows, ld, fcol,
                       nullptr, nullptr, wm, wm + (size_t)PM_C * PM_C, bm, merged, t2, tilemap, ntiles);
    rc = check_launch("rcnn_point_mlp(merge)");
    if (rc != PRCNN_OK) return rc;
    hipLaunchKernelGGL((rows_layer_kernel<1, 0, false>), dim3(grid), dim3(256), 0, st, tiles, per_wg, merged, PM_C, 0, nullptr, 0, 0,
                       nullptr, nullptr, wp, nullptr, bp, p, t3, tilemap, ntiles);
    return check_launch("rcnn_point_mlp(sa1 per-point)");
}


// out (r,128) = act(A0 @ w[0:128] [+ A1 @ w[128:256]] + bias) for r % 64 == 0 rows: the layer kernel above as a general
// entry for the 128-wide shared-MLP / Conv1d layers of the network (FP level 0, the first layers of the RPN heads, the
// per-point parts of SA levels).  A0 = src0 rows (128 floats at column col0, row stride ld0); npanel = 2 adds
// A1 = src1 rows (col1, ld1), i.e. a K = 256 layer whose input is two 128-wide halves (of one tensor or of two).
extern "C" int prcnn_rows_gemm128(long r, int npanel, const float *src0, int ld0, int col0, const float *src1, int ld1,
                                  int col1, const float *w, const float *bias, int relu, float *out, void *stream)
{
    PRCNN_REQUIRE(r >= 0 && r % PM_ROWS == 0, "rows_gemm128: %ld rows is not a multiple of %d", r, PM_ROWS);
    PRCNN_REQUIRE(npanel == 1 || npanel == 2, "rows_gemm128: npanel must be 1 or 2");
    PRCNN_REQUIRE(ld0 % 4 == 0 && col0 % 4 == 0 && col0 >= 0 && col0 + PM_C <= ld0, "rows_gemm128: bad layout of source 0");
    PRCNN_REQUIRE(npanel == 1 || (ld1 % 4 == 0 && col1 % 4 == 0 && col1 >= 0 && col1 + PM_C <= ld1), "rows_gemm128: bad layout of source 1");
    if (r == 0) return PRCNN_OK;
    PRCNN_REQUIRE(src0 && w && bias && out && (npanel == 1 || src1), "rows_gemm128: null pointer");
    PRCNN_REQUIRE((((uintptr_t)src0 | (uintptr_t)out | (npanel == 2 ? (uintptr_t)src1 : 0)) & 15) == 0, "rows_gemm128: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    const long tiles = r / PM_ROWS;
    int per_wg = (int)((tiles + 511) / 512);
    if (per_wg < PM_TILES_PER_WG) per_wg = PM_TILES_PER_WG;
    const int grid = (int)((tiles + per_wg - 1) / per_wg);
    unsigned int *t = next_ticket(st);
    if (!t) { set_error("rows_gemm128: cannot set up the tile ticket"); return PRCNN_ELAUNCH; }
    const float *w1 = w + (size_t)PM_C * PM_C;
#define LAUNCH(NP, RL) hipLaunchKernelGGL((rows_layer_kernel<NP, 0, RL>), dim3(grid), dim3(256), 0, st, tiles, per_wg, src0, ld0, col0, \
                                          src1, ld1, col1, nullptr, nullptr, w, w1, bias, out, t, nullptr, nullptr)
    if (npanel == 1) { if (relu) LAUNCH(1, true); else LAUNCH(1, false); }
    else             { if (relu) LAUNCH(2, true); else LAUNCH(2, false); }
#undef LAUNCH
    return check_launch("rows_gemm128");
}


// The fused entrance chain (only p) over a LIST of rows: rowmap[0 .. hdr[1]) from prcnn_pooled_rows -- tiles of 64 list entries, whichever
// RoIs they belong to.  p rows that are not listed are left as they are.
extern "C" int prcnn_rcnn_point_mlp_rows(long r, int ld, int fcol, const float *rows, const float *wu1, const float *bu1,
                                         const float *wu2, const float *bu2, const float *wm, const float *bm, const float *wp,
                                         const float *bp, float *p, const int *rowmap, const unsigned int *hdr, void *stream)
{
    PRCNN_REQUIRE(r >= 0 && r % PM_ROWS == 0 && r <= 0x7fffffffL, "rcnn_point_mlp_rows: %ld rows is not a multiple of %d", r, PM_ROWS);
    PRCNN_REQUIRE(ld >= 8 && ld % 4 == 0 && fcol >= 8 && fcol % 4 == 0 && fcol + PM_C <= ld,
                  "rcnn_point_mlp_rows: bad row layout ld=%d fcol=%d", ld, fcol);
    if (r == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rows && wu1 && bu1 && wu2 && bu2 && wm && bm && wp && bp && p && rowmap && hdr, "rcnn_point_mlp_rows: null pointer");
    PRCNN_REQUIRE((((uintptr_t)rows | (uintptr_t)wu1 | (uintptr_t)bu1 | (uintptr_t)p | (uintptr_t)wu2 | (uintptr_t)wm | (uintptr_t)wp) & 15) == 0,
                  "rcnn_point_mlp_rows: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    const long tiles = r / PM_ROWS;                            // at most: the list is on the device
    unsigned int *tk = next_ticket(st);
    if (!tk) { set_error("rcnn_point_mlp_rows: cannot set up the tile ticket"); return PRCNN_ELAUNCH; }
    const long grid = tiles < mfma_grid_cap() ? tiles : mfma_grid_cap();
    hipLaunchKernelGGL(rcnn_entrance_kernel, dim3((unsigned)grid), dim3(256), 0, st, tiles, rows, ld, fcol, (const float4 *)wu1,
                       (const float4 *)bu1, wu2, bu2, wm, bm, wp, bp, p, tk, nullptr, hdr, rowmap);
    return check_launch("rcnn_point_mlp_rows");
}

// out (r, 128) rows rowmap[0 .. hdr[1]) = act(src rows (128 floats at column col, row stride ld) @ w (128, 128) k-major + bias): the
// 128-wide layer of prcnn_rows_gemm128 (npanel = 1) over a LIST of rows; rows that are not listed are left as they are.
extern "C" int prcnn_rows_gemm128_rows(long r, const float *src, int ld, int col, const float *w, const float *bias, int relu, float *out,
                                       const int *rowmap, const unsigned int *hdr, void *stream)
{
    PRCNN_REQUIRE(r >= 0 && r <= 0x7fffffffL && ld % 4 == 0 && col % 4 == 0 && col >= 0 && col + PM_C <= ld, "rows_gemm128_rows: bad layout");
    if (r == 0) return PRCNN_OK;
    PRCNN_REQUIRE(src && w && bias && out && rowmap && hdr, "rows_gemm128_rows: null pointer");
    PRCNN_REQUIRE((((uintptr_t)src | (uintptr_t)out) & 15) == 0, "rows_gemm128_rows: 16-byte alignment required");
    const long tiles = (r + PM_ROWS - 1) / PM_ROWS;              // at most: the list is on the device
    const long grid = tiles < mfma_grid_cap() ? tiles : mfma_grid_cap();
    if (relu) hipLaunchKernelGGL(rows_layer_list_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, ld, col, w, bias, out, rowmap, hdr);
    else hipLaunchKernelGGL(rows_layer_list_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, ld, col, w, bias, out, rowmap, hdr);
    return check_launch("rows_gemm128_rows");
}

// cnt (clouds) i32 -> rowmap (clouds * rows_per_cloud entries at most), hdr[1] = number of listed rows (hdr (4 u32) zeroed here unless
// hdr_is_zero): the distinct pooled rows of all clouds back to back, for prcnn_rcnn_point_mlp_rows.
extern "C" int prcnn_pooled_rows(int clouds, int rows_per_cloud, const int *cnt, int *rowmap, unsigned int *hdr, int hdr_is_zero, void *stream)
{
    PRCNN_REQUIRE(clouds >= 0 && rows_per_cloud > 0 && (long)clouds * rows_per_cloud <= 0x7fffffffL, "pooled_rows: bad sizes");
    PRCNN_REQUIRE(hdr && (clouds == 0 || (cnt && rowmap)), "pooled_rows: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (!hdr_is_zero && hipMemsetAsync(hdr, 0, 4 * sizeof(unsigned int), st) != hipSuccess) { set_error("pooled_rows: memset failed"); return PRCNN_ELAUNCH; }
    if (clouds == 0) return PRCNN_OK;
    hipLaunchKernelGGL(pooled_rows_kernel, dim3((clouds + 3) / 4), dim3(256), 0, st, clouds, rows_per_cloud, cnt, rowmap, hdr);
    return check_launch("pooled_rows");
}

// cnt (clouds) i32 distinct rows per cloud of rows_per_cloud rows (prcnn_roipool3d_canonical's pooled_cnt) -> tilemap
// (clouds * rows_per_cloud / 64 entries at most) and hdr[0] = number of live 64-row tiles, for prcnn_rcnn_point_mlp.
extern "C" int prcnn_pooled_tiles(int clouds, int rows_per_cloud, const int *cnt, int *tilemap, unsigned int *hdr, void *stream)
{
    PRCNN_REQUIRE(clouds >= 0 && rows_per_cloud > 0 && rows_per_cloud % PM_ROWS == 0, "pooled_tiles: bad sizes");
    PRCNN_REQUIRE(hdr && (clouds == 0 || (cnt && tilemap)), "pooled_tiles: null pointer");
    hipLaunchKernelGGL(pooled_tiles_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, clouds, rows_per_cloud, cnt, tilemap, hdr);
    return check_launch("pooled_tiles");
}
