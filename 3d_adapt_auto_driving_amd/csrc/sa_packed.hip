// sa_packed.hip -- set-abstraction MLP over the DISTINCT grouped rows only.
//
// The reference's ball query back-fills every slot beyond the hit count with the FIRST hit
// (ball_query_gpu.cu:35-39), and QueryAndGroup / SharedMLP / max_pool2d (pointnet2_utils.py:241-264,
// pointnet2_modules.py:37-53) then push all nsample rows of a group through the three layers -- although the
// back-filled rows are exact copies of row 0 and the max over a group does not change when copies are dropped.
// On the RCNN levels of a KITTI-shaped scene a ball of nsample = 64 holds ~14 (SA1) / ~7 (SA2) distinct
// points; on real clouds it is rarely full either.  So:
//
//   ball_pack_kernel      per cloud: cnt[c] = 1 + (last slot that differs from slot 0), exclusive scan, the
//                         cloud's rows written as a dense list of (centre, point) pairs cut into 64-row tiles
//                         (tiles are allocated from one atomic counter, so the tile count stays on the device)
//   sa_packed_mlp_kernel  the fused gather -> layer 1 -> layer 2 (MFMA) -> layer 3 (MFMA) -> max kernel of
//                         sa_mlp_fused.hip over those tiles: a tile now holds rows of SEVERAL centres, so the
//                         max over nsample becomes a segmented max -- done in the accumulator registers with a
//                         wave-uniform boundary mask -- and partial maxima reach the output with atomicMax
//                         (outputs are >= 0 after ReLU, so the integer order of the bit patterns is the float order).
//
// Every row's result is the same k-ordered fma chain as in sa_mlp_fused.hip (a row's MFMA result does not depend on the
// other rows of its tile), so the output is BIT-IDENTICAL to the unpacked kernel's; only duplicates are skipped.
// `cnt` is defined so that this holds for ANY index tensor: rows beyond the last slot that differs from slot 0 are
// copies of row 0 whatever produced them.
#include "common.hpp"
#include "segmax.hpp"
#include <stdlib.h>

namespace prcnn {

constexpr int PK_C = 128;            // C1 = C2 (narrower levels are zero-padded by the caller)
constexpr int PK_ROWS = 64;
#ifndef PK_PF
#define PK_PF 4                    // rows of P per thread the 128-wide kernel gathers one tile ahead
#endif
constexpr int PK_LD = PK_C + 4;
constexpr int PK_TILES_PER_WG = 8;


// ------------------------------------------------------------------------------------------------ packing
// one workgroup per cloud.  LDS: cnt / offset per centre (m ints) + per-thread partial sums.
// `limit` (optional, per cloud): the points k >= limit[cloud] of the cloud are COPIES of point k % limit[cloud] (the
// wrap-around fill of RoI pooling, roipool3d_kernel.cu:152-159).  A copy lies in a ball iff its original does, and the
// original has the lower index, so in a ball query's answer (first nsample hits in index order) every copy is preceded by
// its original: the rows of the slots with index >= limit are duplicates of rows already listed and are dropped too.
// Every listed row also gets its RELATIVE coordinates xyz[point] - new_xyz[centre] (rowdxyz): the subtraction the tile
// builders used to do (three loads of the point + three of the centre per row) is done once here.
// `rep` (optional, (b, n) i32, round 3): rep[cloud][k] = the lowest-indexed point of the cloud that is an exact copy of point k
// (coordinates AND features; k itself when it is the first of its kind) -- the SA1 centres of a RoI that were sampled from
// wrap-around copies of the same pooled point.  Same argument as for `limit`: a copy lies in a ball iff its representative
// does, and the representative has the lower index, so it is listed earlier in the same row; the slots whose point is not
// its own representative are dropped.  They may sit anywhere in a row, so with `rep` the kept slots are a 64-bit MASK per
// centre (nsample <= 64) instead of a prefix, and output row p of a centre is its p-th kept slot.
__global__ void ball_pack_kernel(int n, int m, int ns, int tiles_cap_cloud, const int *__restrict__ idx, const int *__restrict__ limit,
                                 const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                 unsigned int *__restrict__ rowinfo, float4 *__restrict__ rowdxyz, int *__restrict__ tilecloud,
                                 unsigned int *__restrict__ hdr, int group, const int *__restrict__ rep,
                                 const int *__restrict__ crep)
{
    // Both passes are parallel over ELEMENTS, not over centres (a thread per centre left 32 of 256 threads busy on the RoI
    // clouds' second level and walked each centre's rows as a chain of dependent loads: 47 us for 800 clouds x 32 centres):
    //   1. every thread takes 16-byte pieces of index rows; the centre's distinct count is an LDS atomicMax over its pieces;
    //   2. block scan of the counts -> exclusive offsets;
    //   3. every thread takes OUTPUT rows: centre by binary search in the offsets, slot = row - offset.
    extern __shared__ int pk_lds[];
    int *cnts = pk_lds;                   // [m]   distinct count per centre
    int *offs = pk_lds + m;               // [m]   exclusive offsets
    int *part = pk_lds + 2 * m;           // [blockDim.x] partial sums, then their inclusive scan
    unsigned int *keep = reinterpret_cast<unsigned int *>(pk_lds + 2 * m + blockDim.x);   // [2 m] kept-slot masks (only with rep)
    __shared__ int s_base;
    // `group` consecutive clouds share one row list (one batch of a geometry group: csrc entry prcnn_ball_pack_groups): list l =
    // blockIdx.x / group owns its own slice of the outputs and its own header; tiles record the cloud's index INSIDE its list
    const int gb = blockIdx.x, list = gb / group, b = gb - list * group, tid = threadIdx.x, T = blockDim.x;
    rowinfo += (long)list * group * tiles_cap_cloud * PK_ROWS;
    rowdxyz += (long)list * group * tiles_cap_cloud * PK_ROWS;
    tilecloud += (long)list * group * tiles_cap_cloud;
    hdr += 4 * list;
    const int *rows = idx + (long)gb * m * ns;
    const int lim = limit ? max(limit[gb], 1) : 0x7fffffff;
    const int *__restrict__ rp = rep ? rep + (long)gb * n : nullptr;
    for (int c = tid; c < m; c += T) cnts[c] = 1;
    if (rp)
        for (int c = tid; c < 2 * m; c += T) keep[c] = (c & 1) ? 0u : 1u;       // slot 0 is always kept
    __syncthreads();
    if (rp) {
        for (int e = tid; e < m * ns; e += T) {
            const int c = e / ns, p = e - c * ns;
            const int v = rows[e];
            if (p > 0 && v != rows[(long)c * ns] && v < lim && rp[v] == v) atomicOr(&keep[2 * c + (p >> 5)], 1u << (p & 31));
        }
        __syncthreads();
        for (int c = tid; c < m; c += T) cnts[c] = __popc(keep[2 * c]) + __popc(keep[2 * c + 1]);
    } else if ((ns & 3) == 0) {
        const int q4 = ns >> 2;
        for (int e = tid; e < m * q4; e += T) {
            const int c = e / q4, p = (e - c * q4) * 4;
            const int first = rows[(long)c * ns];
            const int4 v = *reinterpret_cast<const int4 *>(rows + (long)c * ns + p);
            int last = -1;
            if (v.x != first && v.x < lim) last = p;
            if (v.y != first && v.y < lim) last = p + 1;
            if (v.z != first && v.z < lim) last = p + 2;
            if (v.w != first && v.w < lim) last = p + 3;
            if (last > 0) atomicMax(&cnts[c], last + 1);
        }
    } else {
        for (int e = tid; e < m * ns; e += T) {
            const int c = e / ns, p = e - c * ns;
            const int v = rows[e];
            if (p > 0 && v != rows[(long)c * ns] && v < lim) atomicMax(&cnts[c], p + 1);
        }
    }
    __syncthreads();
    if (crep) {
        // `crep` (optional, (b, m) i32): centre c is an exact copy of centre crep[c] <= c (same coordinates, hence the same ball and
        // the same pooled output): it gets NO rows -- its output row stays as the caller left it (zero) and nobody may read it;
        // the level above lists only representatives (its `rep` is this map)
        const int *__restrict__ cr = crep + (long)gb * m;
        for (int c = tid; c < m; c += T)
            if (cr[c] != c) cnts[c] = 0;
        __syncthreads();
    }
    const int chunk = (m + T - 1) / T;
    const int c0 = min(m, tid * chunk), c1 = min(m, c0 + chunk);
    int sum = 0;
    for (int c = c0; c < c1; ++c) sum += cnts[c];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < T; d <<= 1) {     // inclusive scan of the per-thread sums (Hillis-Steele over T <= 1024 entries)
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const int total = part[T - 1];
    int run = part[tid] - sum;
    for (int c = c0; c < c1; ++c) { offs[c] = run; run += cnts[c]; }
    if (tid == 0) {
        const int ntiles = (total + PK_ROWS - 1) / PK_ROWS;
        s_base = (int)atomicAdd(&hdr[0], (unsigned int)ntiles);
        atomicAdd(&hdr[1], (unsigned int)total);
    }
    __syncthreads();
    const int base = s_base;
    const int ntiles = (total + PK_ROWS - 1) / PK_ROWS;
    for (int t = tid; t < ntiles; t += T) tilecloud[base + t] = b;
    unsigned int *dst = rowinfo + (long)base * PK_ROWS;
    float4 *dxyz = rowdxyz + (long)base * PK_ROWS;
    const float *cloud = xyz + (long)gb * n * 3;
    // rows beyond `total` fill the cloud's last tile with copies of its last row (copies do not change a max)
    for (int r = tid; r < ntiles * PK_ROWS; r += T) {
        int c, p;
        if (r < total) {
            int lo = 0, hi = m - 1;         // the last centre whose offset is <= r
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (offs[mid] <= r) lo = mid; else hi = mid - 1;
            }
            c = lo; p = r - offs[lo];
        } else {
            c = m - 1; p = 0;
        }
        const int *row = rows + (long)c * ns;
        if (rp) {                           // p-th kept slot of the centre: select in its 64-bit mask
            unsigned long long mm = ((unsigned long long)keep[2 * c + 1] << 32) | keep[2 * c];
            int slot = 0;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                const int below = __popcll(mm & ((1ull << sft) - 1ull));
                if (p >= below) { p -= below; mm >>= sft; slot += sft; }
            }
            p = slot;
        }
        // slots beyond the limit inside the kept prefix (possible only for index rows that are not a ball query's answer) fall
        // back to the row's first entry: still a copy of a listed row
        const int v = row[p];
        const int k = v < lim ? v : row[0];
        const float *ct = new_xyz + ((long)gb * m + c) * 3;
        const float *pt = cloud + 3 * (long)k;
        dst[r] = ((unsigned int)c << 16) | (unsigned int)k;
        dxyz[r] = make_float4(pt[0] - ct[0], pt[1] - ct[1], pt[2] - ct[2], 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ packing, many workgroups per cloud
// Round 6.  ball_pack_kernel is ONE workgroup per cloud: 32 workgroups for a geometry group, 142 us for the 1.2 M rows of SA1's wider
// scale on LiDAR-shaped scenes, 2.5 launches and 178 us per step there on geometry streams that are 90 % busy (profiles/
// r06_bench_step_kernel_stats_lidar.md).  The plain form (no representative map over the points) as TWO launches whose workgroups own
// PK_CH centres each: (1) counts per centre + their sum per chunk; (2) every workgroup finds its place from the chunk sums -- the
// clouds of a list in INDEX order (no atomic: the tile list is deterministic now), its chunk behind the chunks before it -- and writes
// its rows.  Same rows per cloud in the same order as ball_pack_kernel, same padding of a cloud's last tile, same header.
constexpr int PK_CH = 64;           // centres per workgroup

__global__ __launch_bounds__(256) void ball_pack_count_kernel(int m, int ns, int nchunk, const int *__restrict__ idx, const int *__restrict__ limit,
                                                              const int *__restrict__ crep, int *__restrict__ cnts, int *__restrict__ csum)
{
    __shared__ int s_cnt[PK_CH];
    __shared__ int s_part[4];
    const int chunk = blockIdx.x, gb = blockIdx.y, tid = threadIdx.x;
    const int c0 = chunk * PK_CH, nc = min(PK_CH, m - c0);
    const int *rows = idx + ((long)gb * m + c0) * ns;
    const int lim = limit ? max(limit[gb], 1) : 0x7fffffff;
    if (tid < PK_CH) s_cnt[tid] = 1;
    __syncthreads();
    if ((ns & 3) == 0) {
        const int q4 = ns >> 2;
        for (int e = tid; e < nc * q4; e += 256) {
            const int c = e / q4, p = (e - c * q4) * 4;
            const int first = rows[(long)c * ns];
            const int4 v = *reinterpret_cast<const int4 *>(rows + (long)c * ns + p);
            int last = -1;
            if (v.x != first && v.x < lim) last = p;
            if (v.y != first && v.y < lim) last = p + 1;
            if (v.z != first && v.z < lim) last = p + 2;
            if (v.w != first && v.w < lim) last = p + 3;
            if (last > 0) atomicMax(&s_cnt[c], last + 1);
        }
    } else {
        for (int e = tid; e < nc * ns; e += 256) {
            const int c = e / ns, p = e - c * ns;
            const int v = rows[e];
            if (p > 0 && v != rows[(long)c * ns] && v < lim) atomicMax(&s_cnt[c], p + 1);
        }
    }
    __syncthreads();
    int v = 0;
    if (tid < nc) {
        v = s_cnt[tid];
        if (crep && crep[(long)gb * m + c0 + tid] != c0 + tid) v = 0;          // a centre that copies an earlier one gets no rows
        cnts[(long)gb * m + c0 + tid] = v;
    }
    if (tid < 64) {                                                            // PK_CH = 64: one wave holds the chunk
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (tid == 0) csum[(long)gb * nchunk + chunk] = v;
    }
    (void)s_part;
}

__global__ __launch_bounds__(256) void ball_pack_write_kernel(int n, int m, int ns, int nchunk, int tiles_cap_cloud, const int *__restrict__ idx,
                                                              const int *__restrict__ limit, const float *__restrict__ xyz,
                                                              const float *__restrict__ new_xyz, const int *__restrict__ cnts,
                                                              const int *__restrict__ csum, unsigned int *__restrict__ rowinfo,
                                                              float4 *__restrict__ rowdxyz, int *__restrict__ tilecloud,
                                                              unsigned int *__restrict__ hdr, int group)
{
    __shared__ int s_off[PK_CH + 1];
    __shared__ int s_red[256];
    __shared__ int s_red2[256];
    const int chunk = blockIdx.x, gb = blockIdx.y, tid = threadIdx.x;
    const int list = gb / group, b = gb - list * group;
    rowinfo += (long)list * group * tiles_cap_cloud * PK_ROWS;
    rowdxyz += (long)list * group * tiles_cap_cloud * PK_ROWS;
    tilecloud += (long)list * group * tiles_cap_cloud;
    hdr += 4 * list;
    const int c0 = chunk * PK_CH, nc = min(PK_CH, m - c0);
    // tiles of the list's clouds before this one (index order), rows of this cloud's chunks before this one, rows of the cloud
    const int *lsum = csum + (long)list * group * nchunk;
    int tiles_before = 0, tiles_all = 0, rows_all = 0;
    for (int j = tid; j < group; j += 256) {
        int tot = 0;
        for (int k = 0; k < nchunk; ++k) tot += lsum[(long)j * nchunk + k];
        const int tl = (tot + PK_ROWS - 1) / PK_ROWS;
        if (j < b) tiles_before += tl;
        tiles_all += tl; rows_all += tot;
    }
    int before = 0, total = 0;
    for (int k = tid; k < nchunk; k += 256) {
        const int v = lsum[(long)b * nchunk + k];
        if (k < chunk) before += v;
        total += v;
    }
    auto block_sum = [&](int v, int *buf) {
        buf[tid] = v;
        __syncthreads();
        for (int d = 128; d >= 1; d >>= 1) {
            if (tid < d) buf[tid] += buf[tid + d];
            __syncthreads();
        }
        const int r = buf[0];
        __syncthreads();
        return r;
    };
    tiles_before = block_sum(tiles_before, s_red);
    before = block_sum(before, s_red2);
    total = block_sum(total, s_red);
    if (chunk == 0 && b == 0) {                                                 // the list's header (block-uniform condition)
        tiles_all = block_sum(tiles_all, s_red2);
        rows_all = block_sum(rows_all, s_red);
        if (tid == 0) { hdr[0] = (unsigned int)tiles_all; hdr[1] = (unsigned int)rows_all; }
    }
    // exclusive offsets of this chunk's centres (one wave)
    if (tid < 64) {
        const int v = tid < nc ? cnts[(long)gb * m + c0 + tid] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (tid >= d) incl += o;
        }
        s_off[tid] = incl - v;
        if (tid == 63) s_off[PK_CH] = incl;
    }
    __syncthreads();
    const int mine = s_off[PK_CH];                                              // rows of this chunk
    const int ntiles = (total + PK_ROWS - 1) / PK_ROWS;
    const int lim = limit ? max(limit[gb], 1) : 0x7fffffff;
    const int *rows = idx + (long)gb * m * ns;
    const float *cloud = xyz + (long)gb * n * 3;
    unsigned int *dst = rowinfo + (long)tiles_before * PK_ROWS;
    float4 *dxyz = rowdxyz + (long)tiles_before * PK_ROWS;
    // the cloud's tiles that START inside this chunk's rows are recorded by this workgroup (tile 0 by the first chunk that has rows)
    for (int t = (before + PK_ROWS - 1) / PK_ROWS + tid; (long)t * PK_ROWS < (long)before + mine; t += 256) tilecloud[tiles_before + t] = b;
    // rows beyond `total` fill the cloud's last tile with copies of its LAST CENTRE's first row (as ball_pack_kernel): by the last chunk
    const int extra = (chunk == nchunk - 1) ? ntiles * PK_ROWS - total : 0;
    for (int r = tid; r < mine + extra; r += 256) {
        int c, p;
        if (r < mine) {
            int lo = 0, hi = nc - 1;                                            // the last centre of the chunk whose offset is <= r
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_off[mid] <= r) lo = mid; else hi = mid - 1;
            }
            c = c0 + lo; p = r - s_off[lo];
        } else {
            c = m - 1; p = 0;
        }
        const int *row = rows + (long)c * ns;
        const int v = row[p];
        const int k = v < lim ? v : row[0];
        const float *ct = new_xyz + ((long)gb * m + c) * 3;
        const float *pt = cloud + 3 * (long)k;
        const long o = r < mine ? (long)before + r : (long)total + (r - mine);
        dst[o] = ((unsigned int)c << 16) | (unsigned int)k;
        dxyz[o] = make_float4(pt[0] - ct[0], pt[1] - ct[1], pt[2] - ct[2], 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ C3 = 128
// Two workgroups per CU (see sa_mlp_fused.hip for the MFMA mapping; identical here).
// Round 5: ONE launch serves up to two PROBLEMS (the two scales of an MSG level: their own row lists, weights and output slices):
// workgroup x works on problem x % nprob and draws its tiles from that problem's counter (words 0 and 2 of the launch's ticket record).
// On the sparse launches of the real step (1-2 tiles per CU) a launch is mostly ramp: two of them side by side cost one ramp.
struct SaPkProblem {
    int n, m;
    const unsigned int *hdr;
    const float4 *rowdxyz, *P, *wxyz;
    const unsigned int *rowinfo;
    const int *tilecloud;
    const float *w2t, *b2, *w3t, *b3;
    float *out;
    int out_stride, out_col, lds_pool;
    int c1, c2;                       // the real widths of layers 1 and 2 (the arrays are zero-padded to 128 whatever they are)
};
struct SaPkBatch {
    int nprob;
    SaPkProblem p[2];
};
// thread 0 of every workgroup, once, after its last draw: the launch's last workgroup zeroes the record's draw counters and arrival count
__device__ __forceinline__ void ticket_release3(unsigned int *rec)
{
    if (atomicAdd(rec + 1, 1u) == gridDim.x * gridDim.y * gridDim.z - 1u) {
        atomicExch(rec, 0u);
        atomicExch(rec + 2, 0u);
        atomicExch(rec + 1, 0u);
    }
}

// Round 5, third session: the two scales of RPN SA2 (64-64-128 and 64-96-128 after the per-point layer; tools/cfgs/default.yaml's
// SA_CONFIG.MLPS[1]) arrive zero-padded to 128-128-128, and the padded kernel spends 256 MFMAs per wave and tile where 96 / 160 carry
// non-zero operands.  A padded chain is  acc = fma(a[s], w[s], acc); acc = fma(a[s + 64], w[s + 64], acc)  for s = 0..63 (the
// instruction's two k values in that order); with zeros at k >= K a second step is  fma(0, 0, acc) = acc,  so the chain IS the
// chain over the real k in this order:  K = 64: k = 0, 1, 2, ... 63;  K = 96: k = 0, 64, 1, 65, ... 31, 95, then 32, 33, ... 63.
// Fed to the instruction as pairs -- K = 64: step s takes k = 2s (lanes 0-31) and 2s + 1 (lanes 32-63), 32 steps;  K = 96:
// steps 0-31 as before (s, s + 64), then 16 steps (32 + 2u, 33 + 2u) -- the result is the padded chain's, bit for bit (an
// accumulator of -0 would come out of fma(0, 0, -0) as +0: no chain that starts from +0 gets there short of products below
// 1e-45).  Layer 2 of the 64-wide scale has 2 x 2 blocks of 32 x 32: one per wave; the 96-wide one leaves wave 3 idle.  The tile
// builder works on 64 channels: 16 float4 chunks x 16 row slots, four rows per thread, ALL of them gathered one tile ahead.
template <int C2>
__device__ __forceinline__ void sa_pk128_narrow(const SaPkProblem &q, unsigned int *__restrict__ ticket, int tiles_per_wg, float *lds,
                                                unsigned int *slot, int (*ctr)[PK_ROWS], float4 (*dxyz_s)[PK_ROWS])
{
    static_assert(C2 == 64 || C2 == 96, "layer 2 of RPN SA2");
    float *A1 = lds, *Y1 = lds + PK_ROWS * PK_LD;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n = q.n, m = q.m;
    const unsigned int *__restrict__ hdr = q.hdr;
    const float4 *__restrict__ rowdxyz = q.rowdxyz;
    const float4 *__restrict__ P = q.P, *__restrict__ wxyz = q.wxyz;
    const unsigned int *__restrict__ rowinfo = q.rowinfo;
    const int *__restrict__ tilecloud = q.tilecloud;
    const float *__restrict__ w2t = q.w2t, *__restrict__ b2 = q.b2, *__restrict__ w3t = q.w3t, *__restrict__ b3 = q.b3;
    float *__restrict__ out = q.out;
    const int out_stride = q.out_stride, out_col = q.out_col, lds_pool = q.lds_pool;
    const long tiles = hdr[0];

    __syncthreads();                                                  // (a second attempt reuses the slots and the tile buffers)
    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = slot[0];
    if (t >= tiles) return;

    constexpr int S3 = C2 / 2;                                        // MFMA steps of layer 3
    const int cb2 = C2 == 64 ? (w & 1) : w, rb2 = w >> 1;             // layer 2: column block; C2 == 64: row block as well
    float wf2[32], wf3[S3];
#pragma unroll
    for (int s = 0; s < 32; ++s) wf2[s] = w2t[(long)(2 * s + h) * PK_C + 32 * cb2 + j];
#pragma unroll
    for (int s = 0; s < S3; ++s) {
        const int k = C2 == 64 ? 2 * s + h : (s < 32 ? s + 64 * h : 32 + 2 * (s - 32) + h);
        wf3[s] = w3t[(long)k * 128 + 32 * w + j];
    }
    const float bias2 = b2[32 * cb2 + j], bias3 = b3[32 * w + j];

    const int chunk = tid & 15, r0 = tid >> 4;
    const float4 wx = wxyz[chunk], wy = wxyz[32 + chunk], wz = wxyz[64 + chunk];
    unsigned int info[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) info[i] = rowinfo[t * PK_ROWS + r0 + 16 * i];
    int cloud = tilecloud[t];
    if (tid < PK_ROWS) dxyz_s[0][tid] = rowdxyz[t * PK_ROWS + tid];
    float4 pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = P[((long)cloud * n + (long)(info[i] & 0xffffu)) * (PK_C / 4) + chunk];
    __syncthreads();

    for (int served = 0; served < tiles_per_wg && t < tiles; ++served) {
        const bool more = served + 1 < tiles_per_wg;
        if (tid == 0) slot[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
        int *cc = ctr[served & 1];
        const float4 *dcur = dxyz_s[served & 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 16 * i;
            const float4 d = dcur[row];
            const float4 v = affine_relu4(pb[i], wx, wy, wz, d.x, d.y, d.z);
            *reinterpret_cast<float4 *>(A1 + row * PK_LD + 4 * chunk) = v;
            if (chunk == 0) cc[row] = (int)((long)cloud * m + (long)(info[i] >> 16));
        }
        __syncthreads();
        const long t_next = slot[(served + 1) & 1];
        float4 dnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t_next < tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) info[i] = rowinfo[t_next * PK_ROWS + r0 + 16 * i];
            cloud = tilecloud[t_next];
            if (tid < PK_ROWS) dnext = rowdxyz[t_next * PK_ROWS + tid];                    // in flight during layer 2
        }

        // ---- layer 2: K = 64 as 32 steps of (2s, 2s + 1)
        if (C2 == 64) {
            f32x16 acc = {0};
            const float *ap = A1 + (32 * rb2 + j) * PK_LD + h;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[4 * g], wf2[2 * g], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[4 * g + 2], wf2[2 * g + 1], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                Y1[(32 * rb2 + row) * PK_LD + 32 * cb2 + j] = fmaxf(acc[r] + bias2, 0.f);
            }
        } else if (w < 3) {
            f32x16 acc0 = {0}, acc1 = {0};
            const float *a0p = A1 + j * PK_LD + h;
            const float *a1p = A1 + (32 + j) * PK_LD + h;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[4 * g], wf2[2 * g], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[4 * g], wf2[2 * g], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[4 * g + 2], wf2[2 * g + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[4 * g + 2], wf2[2 * g + 1], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                Y1[row * PK_LD + 32 * w + j] = fmaxf(acc0[r] + bias2, 0.f);
                Y1[(32 + row) * PK_LD + 32 * w + j] = fmaxf(acc1[r] + bias2, 0.f);
            }
        }
        if (tid < PK_ROWS && t_next < tiles) dxyz_s[(served + 1) & 1][tid] = dnext;   // ordered before the next builder by the barrier below
        if (t_next < tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i)          // in flight during layer 3
                pb[i] = P[((long)cloud * n + (long)(info[i] & 0xffffu)) * (PK_C / 4) + chunk];
        }
        __syncthreads();

        // ---- layer 3 (K = C2) + segmented max over the tile's rows
        {
            f32x16 acc0 = {0}, acc1 = {0};
            if (C2 == 96) {
                const float *a0p = Y1 + j * PK_LD + 64 * h;
                const float *a1p = Y1 + (32 + j) * PK_LD + 64 * h;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
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
            }
            {
                constexpr int K0 = C2 == 96 ? 32 : 0, S0 = C2 == 96 ? 32 : 0;       // the (2u, 2u + 1) steps: k from K0, weights from S0
                const float *a0p = Y1 + j * PK_LD + K0 + h;
                const float *a1p = Y1 + (32 + j) * PK_LD + K0 + h;
#pragma unroll
                for (int g = 0; g < (S3 - S0) / 2; ++g) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[4 * g], wf3[S0 + 2 * g], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[4 * g], wf3[S0 + 2 * g], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[4 * g + 2], wf3[S0 + 2 * g + 1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[4 * g + 2], wf3[S0 + 2 * g + 1], acc1, 0, 0, 0);
                }
            }
            const int myc = cc[lane], prevc = cc[lane ? lane - 1 : 0];
            const unsigned long long start = __ballot(lane == 0 || myc != prevc);       // the same word in every wave
            if (!lds_pool || __popcll(start) <= PK_LDS_MIN) {
                pk_segmented_max(acc0, acc1, cc, start, h, out, out_stride, out_col + 32 * w + j, bias3);
            } else {
                const float4 bias4 = *reinterpret_cast<const float4 *>(b3 + 4 * (tid & 31));
                pk_park(acc0, acc1, A1, PK_LD, 32 * w + j, h);
                lds_barrier();
                pk_segmented_max_lds(A1, PK_LD, cc, tid, out, out_stride, out_col, bias4);
                lds_barrier();                                 // the next builder rewrites A1
            }
        }
        t = t_next;
    }
}

template <int NPROB, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void sa_packed_mlp128_kernel(const SaPkBatch batch, unsigned int *__restrict__ rec, int tiles_per_wg)
{
    __shared__ float lds[2 * PK_ROWS * PK_LD];
    __shared__ unsigned int slot[2];
    __shared__ int ctr[2][PK_ROWS];
    // the centre-relative coordinates of a tile's rows stream from HBM (16 B per row, read once): they are fetched ONE TILE
    // AHEAD by wave 0 and parked here, so that the builder finds them in LDS instead of waiting for memory
    __shared__ float4 dxyz_s[2][PK_ROWS];
    float *A1 = lds, *Y1 = lds + PK_ROWS * PK_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    // a workgroup starts on problem x % nprob and, when that problem's tiles have run out, moves on to the other one (it reloads the
    // weights once): the tile counts live on the device, so the split of the launch's workgroups between the problems cannot be
    // chosen on the host -- LiDAR-shaped scenes give the two scales of RPN SA2 27 % and 73 % of the rows
    // (NPROB = 1, every single-problem launch: no loop, the kernel of rounds 2-4 register for register)
    const int pi0 = NPROB > 1 ? (int)(blockIdx.x % (unsigned)NPROB) : 0;
#pragma unroll 1
    for (int attempt = 0; attempt < NPROB; ++attempt) {
    const int pi = NPROB > 1 ? __builtin_amdgcn_readfirstlane(pi0 + attempt < NPROB ? pi0 + attempt : pi0 + attempt - NPROB) : 0;   // provably uniform: the problem's fields stay in SGPRs
    const SaPkProblem &q = batch.p[pi];
    if (NARROW) {                                                     // the scales of RPN SA2: every problem of the launch has c1 = 64, c2 in {64, 96}
        if (q.c2 == 64) sa_pk128_narrow<64>(q, rec + 2 * pi, tiles_per_wg, lds, slot, ctr, dxyz_s);
        else sa_pk128_narrow<96>(q, rec + 2 * pi, tiles_per_wg, lds, slot, ctr, dxyz_s);
        continue;
    }
    const int n = q.n, m = q.m;
    const unsigned int *__restrict__ hdr = q.hdr;
    const float4 *__restrict__ rowdxyz = q.rowdxyz;
    const float4 *__restrict__ P = q.P /* (b,n,128) */, *__restrict__ wxyz = q.wxyz /* (3,128) */;
    const unsigned int *__restrict__ rowinfo = q.rowinfo;
    const int *__restrict__ tilecloud = q.tilecloud;
    const float *__restrict__ w2t = q.w2t, *__restrict__ b2 = q.b2, *__restrict__ w3t = q.w3t, *__restrict__ b3 = q.b3;
    float *__restrict__ out = q.out;
    const int out_stride = q.out_stride, out_col = q.out_col, lds_pool = q.lds_pool;
    unsigned int *__restrict__ ticket = rec + 2 * pi;                 // this problem's draw counter
    // tilecloud == NULL (round 5): a list whose ROWS carry their cloud -- descriptor (cloud << 16) | (centre << 9) | point, written by
    // prcnn_rcnn_roi_geometry_packs for the RoI clouds (512 points, 128 centres at most) -- and whose hdr[1] rows are cut into tiles
    // wherever they fall: a tile holds the rows of several clouds, only the list's last tile is partly filled (its missing rows
    // read as copies of the list's last row: copies do not change a max).  With a tile per cloud the 1600 RoI clouds of a launch
    // padded 39 rows to 64 on the second level of the uniform scene, 102 to 128 on LiDAR-shaped ones.
    const bool rowcloud = tilecloud == nullptr;
    const long nrows = hdr[1];
    const long tiles = rowcloud ? (nrows + PK_ROWS - 1) / PK_ROWS : (long)hdr[0];
    const long last_row = rowcloud ? nrows - 1 : 0x7fffffffffffL;
    const unsigned int kmask = rowcloud ? 0x1ffu : 0xffffu, cmask = rowcloud ? 0x7fu : 0xffffu;
    const int cshift = rowcloud ? 9 : 16;

    // first ticket BEFORE the weights are fetched: the grid is sized for the worst case (every ball full) and most
    // workgroups of a sparse launch leave right here
    __syncthreads();                                                  // (a second attempt reuses the slots and the tile buffers)
    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = slot[0];
    if (t >= tiles) continue;

    float wf2[64], wf3[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) wf2[s] = w2t[(long)(s + 64 * h) * PK_C + 32 * w + j];
#pragma unroll
    for (int s = 0; s < 64; ++s) wf3[s] = w3t[(long)(s + 64 * h) * 128 + 32 * w + j];
    const float bias2 = b2[32 * w + j], bias3 = b3[32 * w + j];

    const int chunk = tid & 31, r0 = tid >> 5;
    const float4 wx = wxyz[chunk], wy = wxyz[32 + chunk], wz = wxyz[64 + chunk];

    unsigned int info[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) info[i] = rowinfo[min(t * PK_ROWS + r0 + 8 * i, last_row)];
    int cloud = rowcloud ? 0 : tilecloud[t];
    if (tid < PK_ROWS) dxyz_s[0][tid] = rowdxyz[min(t * PK_ROWS + tid, last_row)];
    // half of the tile's 8 rows of P per thread are gathered ONE TILE AHEAD (behind layer 3's MFMAs; all 8 do not fit the
    // register budget next to the 128 resident weights: 164 bytes of spills, slower on sparse tiles): the builder's wait for its
    // gathers was 12-15k of a tile's 50k cycles (s_memtime stamps, profiles/r02_stage_stamps.md)
    float4 pb[PK_PF];
#pragma unroll
    for (int i = 0; i < PK_PF; ++i)
        pb[i] = P[((long)(rowcloud ? (int)(info[i] >> 16) : cloud) * n + (long)(info[i] & kmask)) * (PK_C / 4) + chunk];
    __syncthreads();

    for (int served = 0; served < tiles_per_wg && t < tiles; ++served) {
        const bool more = served + 1 < tiles_per_wg;
        if (tid == 0) slot[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
        int *cc = ctr[served & 1];
        const float4 *dcur = dxyz_s[served & 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = r0 + 8 * i;
            const int k = (int)(info[i] & kmask), cl = (int)((info[i] >> cshift) & cmask);
            const long cl_ = rowcloud ? (long)(info[i] >> 16) : (long)cloud;
            const float4 d = dcur[row];
            const float dx = d.x, dy = d.y, dz = d.z;
            const float4 base = i < PK_PF ? pb[i] : P[(cl_ * n + k) * (PK_C / 4) + chunk];
            const float4 v = affine_relu4(base, wx, wy, wz, dx, dy, dz);
            *reinterpret_cast<float4 *>(A1 + row * PK_LD + 4 * chunk) = v;
            if (chunk == 0) cc[row] = (int)(cl_ * m + cl);
        }
        __syncthreads();
        const long t_next = slot[(served + 1) & 1];
        float4 dnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t_next < tiles) {
#pragma unroll
            for (int i = 0; i < 8; ++i) info[i] = rowinfo[min(t_next * PK_ROWS + r0 + 8 * i, last_row)];
            if (!rowcloud) cloud = tilecloud[t_next];
            if (tid < PK_ROWS) dnext = rowdxyz[min(t_next * PK_ROWS + tid, last_row)];     // in flight during layer 2
        }

        // ---- layer 2
        {
            f32x16 acc0 = {0}, acc1 = {0};
            const float *a0p = A1 + j * PK_LD + 64 * h;
            const float *a1p = A1 + (32 + j) * PK_LD + 64 * h;
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                Y1[row * PK_LD + 32 * w + j] = fmaxf(acc0[r] + bias2, 0.f);
                Y1[(32 + row) * PK_LD + 32 * w + j] = fmaxf(acc1[r] + bias2, 0.f);
            }
        }
        if (tid < PK_ROWS && t_next < tiles) dxyz_s[(served + 1) & 1][tid] = dnext;   // ordered before the next builder by the barrier below
        if (t_next < tiles) {
#pragma unroll
            for (int i = 0; i < PK_PF; ++i)      // in flight during layer 3
                pb[i] = P[((long)(rowcloud ? (int)(info[i] >> 16) : cloud) * n + (long)(info[i] & kmask)) * (PK_C / 4) + chunk];
        }
        __syncthreads();

        // ---- layer 3 + segmented max over the tile's rows
        {
            const float *a0p = Y1 + j * PK_LD + 64 * h;
            const float *a1p = Y1 + (32 + j) * PK_LD + 64 * h;
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
            const int myc = cc[lane], prevc = cc[lane ? lane - 1 : 0];
            const unsigned long long start = __ballot(lane == 0 || myc != prevc);       // the same word in every wave
            if (!lds_pool || __popcll(start) <= PK_LDS_MIN) {
                pk_segmented_max(acc0, acc1, cc, start, h, out, out_stride, out_col + 32 * w + j, bias3);
            } else {
                // many centres in the tile: through LDS, row-major (segmax.hpp).  A1 is free (last read in layer 2, two barriers ago).
                const float4 bias4 = *reinterpret_cast<const float4 *>(b3 + 4 * (tid & 31));
                pk_park(acc0, acc1, A1, PK_LD, 32 * w + j, h);
                lds_barrier();
                pk_segmented_max_lds(A1, PK_LD, cc, tid, out, out_stride, out_col, bias4);
                lds_barrier();                                 // the next builder rewrites A1
            }
        }
        // A1 is rewritten by the next builder only (every wave has left layer 2); Y1 after the next tile's first barrier;
        // the centre list alternates between two buffers, so a slow wave still reads this tile's list while the others
        // already write the next one.
        t = t_next;
    }
    }   // attempt
    if (tid == 0) ticket_release3(rec);          // the launch's last workgroup zeroes the counters for the record's next user
}

// ------------------------------------------------------------------------------------------------ C3 = 256
// Eight waves (sa_mlp_fused256_kernel's mapping): wave (wp, wg) owns column panel wp of layer 2 for row half wg, and column
// panel wp of column tile wg of layer 3 for both row halves.
__global__ __launch_bounds__(512, 1) void sa_packed_mlp256_kernel(
    int n, int m, const unsigned int *__restrict__ hdr, const float4 *__restrict__ rowdxyz,
    const float4 *__restrict__ P, const float4 *__restrict__ wxyz, const unsigned int *__restrict__ rowinfo,
    const int *__restrict__ tilecloud, const float *__restrict__ w2t, const float *__restrict__ b2,
    const float *__restrict__ w3t /* (128,256) */, const float *__restrict__ b3, float *__restrict__ out, int out_stride,
    int out_col, unsigned int *__restrict__ ticket, int tiles_per_wg, int lds_pool)
{
    __shared__ float lds[2 * PK_ROWS * PK_LD];
    __shared__ unsigned int slot[2];
    __shared__ int ctr[2][PK_ROWS];
    __shared__ float4 dxyz_s[2][PK_ROWS];             // next tile's coordinates, fetched one tile ahead (see the 128 kernel)
    float *A1 = lds, *Y1 = lds + PK_ROWS * PK_LD;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, wp = w & 3, wg = w >> 2;
    const int j = lane & 31, h = lane >> 5;
    const bool rowcloud = tilecloud == nullptr;                       // the rows carry their cloud: see sa_packed_mlp128_kernel
    const long nrows = hdr[1];
    const long tiles = rowcloud ? (nrows + PK_ROWS - 1) / PK_ROWS : (long)hdr[0];
    const long last_row = rowcloud ? nrows - 1 : 0x7fffffffffffL;
    const unsigned int kmask = rowcloud ? 0x1ffu : 0xffffu, cmask = rowcloud ? 0x7fu : 0xffffu;
    const int cshift = rowcloud ? 9 : 16;

    if (tid == 0) slot[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    long t = slot[0];
    if (t >= tiles) { if (tid == 0) ticket_release(ticket); return; }

    float wf2[64], wf3[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) wf2[s] = w2t[(long)(s + 64 * h) * PK_C + 32 * wp + j];
#pragma unroll
    for (int s = 0; s < 64; ++s) wf3[s] = w3t[(long)(s + 64 * h) * 256 + 128 * wg + 32 * wp + j];
    const float bias2 = b2[32 * wp + j], bias3 = b3[128 * wg + 32 * wp + j];

    const int chunk = tid & 31, r0 = tid >> 5;       // rows r0 + 16 i, i < 4
    const float4 wx = wxyz[chunk], wy = wxyz[32 + chunk], wz = wxyz[64 + chunk];

    unsigned int info[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) info[i] = rowinfo[min(t * PK_ROWS + r0 + 16 * i, last_row)];
    int cloud = rowcloud ? 0 : tilecloud[t];
    if (tid < PK_ROWS) dxyz_s[0][tid] = rowdxyz[min(t * PK_ROWS + tid, last_row)];
    // the tile's 4 rows of P per thread are gathered ONE TILE AHEAD (behind layer 3's MFMAs): the builder used to wait for them,
    // a quarter of a tile's cycles (s_memtime stamps of the 128-wide kernel, profiles/r02_stage_stamps.md)
    float4 pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = P[((long)(rowcloud ? (int)(info[i] >> 16) : cloud) * n + (long)(info[i] & kmask)) * (PK_C / 4) + chunk];
    __syncthreads();

    for (int served = 0; served < tiles_per_wg && t < tiles; ++served) {
        const bool more = served + 1 < tiles_per_wg;
        if (tid == 0) slot[(served + 1) & 1] = more ? atomicAdd(ticket, 1u) : 0xffffffffu;
        int *cc = ctr[served & 1];
        const float4 *dcur = dxyz_s[served & 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 16 * i;
            const int cl = (int)((info[i] >> cshift) & cmask);
            const long cbase = (rowcloud ? (long)(info[i] >> 16) : (long)cloud) * m;
            const float4 d = dcur[row];
            const float dx = d.x, dy = d.y, dz = d.z;
            const float4 base = pb[i];
            const float4 v = affine_relu4(base, wx, wy, wz, dx, dy, dz);
            *reinterpret_cast<float4 *>(A1 + row * PK_LD + 4 * chunk) = v;
            if (chunk == 0) cc[row] = (int)(cbase + cl);
        }
        __syncthreads();
        const long t_next = slot[(served + 1) & 1];
        float4 dnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t_next < tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) info[i] = rowinfo[min(t_next * PK_ROWS + r0 + 16 * i, last_row)];
            if (!rowcloud) cloud = tilecloud[t_next];
            if (tid < PK_ROWS) dnext = rowdxyz[min(t_next * PK_ROWS + tid, last_row)];
        }

        // ---- layer 2: rows [32 wg, 32 wg + 32) x columns [32 wp, 32 wp + 32)
        {
            f32x16 acc = {0};
            const float *ap = A1 + (32 * wg + j) * PK_LD + 64 * h;
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
                Y1[row * PK_LD + 32 * wp + j] = fmaxf(acc[r] + bias2, 0.f);
            }
        }
        if (tid < PK_ROWS && t_next < tiles) dxyz_s[(served + 1) & 1][tid] = dnext;
        if (t_next < tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i)          // in flight during layer 3
                pb[i] = P[((long)(rowcloud ? (int)(info[i] >> 16) : cloud) * n + (long)(info[i] & kmask)) * (PK_C / 4) + chunk];
        }
        __syncthreads();

        // ---- layer 3: all 64 rows x columns [128 wg + 32 wp, +32), then the segmented max
        {
            const float *a0p = Y1 + j * PK_LD + 64 * h;
            const float *a1p = Y1 + (32 + j) * PK_LD + 64 * h;
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
            const int myc = cc[lane], prevc = cc[lane ? lane - 1 : 0];
            const unsigned long long start = __ballot(lane == 0 || myc != prevc);
            if (!lds_pool || __popcll(start) <= PK_LDS_MIN) {
                pk_segmented_max(acc0, acc1, cc, start, h, out, out_stride, out_col + 128 * wg + 32 * wp + j, bias3);
            } else {
                // many centres in the tile: through LDS, row-major (segmax.hpp), one 64 x 128 tile per column tile: A1 is free
                // (last read in layer 2), Y1 once every wave has left layer 3
                float *Z = wg ? Y1 : A1;
                const float4 bias4 = *reinterpret_cast<const float4 *>(b3 + 128 * wg + 4 * (tid & 31));
                lds_barrier();
                pk_park(acc0, acc1, Z, PK_LD, 32 * wp + j, h);
                lds_barrier();
                pk_segmented_max_lds(Z, PK_LD, cc, tid & 255, out, out_stride, out_col + 128 * wg, bias4);
                lds_barrier();                                 // the next builder rewrites A1
            }
        }
        t = t_next;
    }
    if (tid == 0) ticket_release(ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}

unsigned int *next_ticket(hipStream_t st);    // sa_mlp_fused.hip

}  // namespace prcnn

using namespace prcnn;

// idx (b,m,nsample) i32 -> the distinct grouped rows of every cloud as 64-row tiles:
//   rowinfo  [b * tiles_cap * 64] u32, (centre within cloud) << 16 | (point within cloud); tile t = entries [64t, 64t+64)
//   rowdxyz  [b * tiles_cap * 64] float4, xyz[point] - new_xyz[centre] of the same rows (w = 0)
//   tilecloud[b * tiles_cap] i32, cloud of tile t
//   hdr      [4] u32: [0] = number of tiles, [1] = number of distinct rows (both written by this call)
// with tiles_cap = ceil(m * nsample / 64) tiles per cloud at most.  Needs m, n <= 65536.
// limit (b) i32, optional: points k >= limit[cloud] are copies of point k % limit[cloud] (see ball_pack_kernel).
static int ball_pack_launch(int b, int group, int n, int m, int nsample, const int *idx, const int *limit, const float *xyz,
                            const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud, unsigned int *hdr, void *stream,
                            const int *rep = nullptr, const int *crep = nullptr, int hdr_is_zero = 0)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 1, "ball_pack: bad sizes");
    PRCNN_REQUIRE(group >= 1 && b % group == 0, "ball_pack: %d clouds do not split into lists of %d", b, group);
    PRCNN_REQUIRE(m <= 65536 && m <= 15360, "ball_pack: m=%d centres per cloud unsupported (<= 15360)", m);
    PRCNN_REQUIRE(n <= 65536, "ball_pack: n=%d points per cloud unsupported (rowinfo keeps the point in 16 bits)", n);
    PRCNN_REQUIRE(hdr, "ball_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int lists = b > 0 ? b / group : 1;
    // (round 4 tried counting in a self-resetting ticket record instead of this memset: 6323 / 6337 scenes/s against 6362 / 6402 at K = 96,
    //  6103 / 6120 with agent-scope fences around the arrival count -- the five fills per step are on nobody's critical path.  Not kept.)
    //  Round 5: the caller may hand over headers that ARE zero -- slices of an arena it zeroes once per chain of launches
    //  (prcnn_ball_pack_ex: 3.65 header fills per step -> 0.75 arena fills that also cover the levels' pooled outputs).)
    if (!hdr_is_zero && hipMemsetAsync(hdr, 0, (size_t)lists * 4 * sizeof(unsigned int), st) != hipSuccess) { set_error("ball_pack: memset failed"); return PRCNN_ELAUNCH; }
    if (b == 0 || m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(idx && rowinfo && tilecloud && xyz && new_xyz && rowdxyz, "ball_pack: null pointer");
    PRCNN_REQUIRE(((uintptr_t)rowdxyz & 15) == 0, "ball_pack: rowdxyz must be 16-byte aligned");
    PRCNN_REQUIRE(((uintptr_t)idx & 15) == 0 || (nsample & 3) != 0, "ball_pack: 16-byte alignment required");
    if (!rep && m >= 512) {
        // many workgroups per cloud (round 6): counts, then rows -- see ball_pack_count_kernel
        const int nchunk = (m + PK_CH - 1) / PK_CH;
        int *scr = (int *)scratch_for(st, ((size_t)b * m + (size_t)b * nchunk) * sizeof(int), 14);
        if (!scr) { set_error("ball_pack: cannot allocate the count scratch"); return PRCNN_ELAUNCH; }
        int *cnts = scr, *csum = scr + (size_t)b * m;
        const int cap = (int)(((long)m * nsample + PK_ROWS - 1) / PK_ROWS);
        hipLaunchKernelGGL(ball_pack_count_kernel, dim3(nchunk, b), dim3(256), 0, st, m, nsample, nchunk, idx, limit, crep, cnts, csum);
        hipLaunchKernelGGL(ball_pack_write_kernel, dim3(nchunk, b), dim3(256), 0, st, n, m, nsample, nchunk, cap, idx, limit, xyz, new_xyz, cnts, csum,
                           rowinfo, (float4 *)rowdxyz, tilecloud, hdr, group);
        return check_launch("ball_pack");
    }
    int threads = 64;                              // enough threads for the cloud's index elements, 16 bytes each
    while (threads < (int)(((long)m * nsample + 3) / 4) && threads < 1024) threads *= 2;
    PRCNN_REQUIRE(!rep || nsample <= 64, "ball_pack: a representative map needs nsample <= 64 (got %d)", nsample);
    const size_t lds = ((size_t)2 * m + threads + (rep ? 2 * m : 0)) * sizeof(int);
    if (lds > 48 * 1024) {
        const int rc = ensure_dynamic_lds((const void *)ball_pack_kernel, lds, "ball_pack");
        if (rc != PRCNN_OK) return rc;
    }
    const int cap = (int)(((long)m * nsample + PK_ROWS - 1) / PK_ROWS);
    hipLaunchKernelGGL(ball_pack_kernel, dim3(b), dim3(threads), lds, st, n, m, nsample, cap, idx, limit, xyz, new_xyz, rowinfo,
                       (float4 *)rowdxyz, tilecloud, hdr, group, rep, crep);
    return check_launch("ball_pack");
}

// prcnn_ball_pack with representative maps (see ball_pack_kernel): rep (b, n) i32 over the POINTS, rep[cloud][k] <= k,
// rep[cloud][rep[cloud][k]] == rep[cloud][k]: the slots whose point is not its own representative are dropped as well;
// crep (b, m) i32 over the CENTRES, same properties: a centre that is not its own representative gets no rows at all.  Either may be null.
extern "C" int prcnn_ball_pack_rep(int b, int n, int m, int nsample, const int *idx, const int *limit, const int *rep, const int *crep,
                                   const float *xyz, const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud,
                                   unsigned int *hdr, void *stream)
{
    return ball_pack_launch(b, b > 0 ? b : 1, n, m, nsample, idx, limit, xyz, new_xyz, rowinfo, rowdxyz, tilecloud, hdr, stream, rep, crep);
}

namespace prcnn {
// rep[cloud][j] = the lowest j' with src(sel[cloud][j']) == src(sel[cloud][j]), where src(i) = prev ? prev[cloud][i] : i % limit[cloud]
// (limit: the points k >= limit[cloud] are copies of k % limit[cloud]; prev: the representative map of the level below).
__global__ void dup_rep_kernel(int n, int m, const int *__restrict__ sel, const int *__restrict__ limit, const int *__restrict__ prev,
                               int *__restrict__ rep)
{
    extern __shared__ int first[];             // [n]: lowest j sampled from each source point
    const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    for (int i = tid; i < n; i += T) first[i] = 0x7fffffff;
    __syncthreads();
    const int lim = limit ? max(limit[b], 1) : 0x7fffffff;
    const int *__restrict__ s = sel + (long)b * m;
    const int *__restrict__ pv = prev ? prev + (long)b * n : nullptr;
    for (int j = tid; j < m; j += T) {
        const int i = s[j];
        const int src = pv ? pv[i] : (i >= lim ? i % lim : i);
        atomicMin(&first[src], j);
    }
    __syncthreads();
    for (int j = tid; j < m; j += T) {
        const int i = s[j];
        const int src = pv ? pv[i] : (i >= lim ? i % lim : i);
        rep[(long)b * m + j] = first[src];
    }
}
}  // namespace prcnn

/* sel (b, m) i32: indices into clouds of n points (an FPS answer).  limit (b) i32 and / or prev (b, n) i32 say which of the n
 * points are exact copies of one another; rep (b, m) i32 <- for every sampled point the first sampled point of the same source. */
extern "C" int prcnn_dup_rep(int b, int n, int m, const int *sel, const int *limit, const int *prev, int *rep, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && n <= 12288, "dup_rep: bad sizes b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(sel && rep && (limit || prev), "dup_rep: null pointer");
    hipLaunchKernelGGL(dup_rep_kernel, dim3(b), dim3(128), (size_t)n * sizeof(int), (hipStream_t)stream, n, m, sel, limit, prev, rep);
    return check_launch("dup_rep");
}

extern "C" int prcnn_ball_pack(int b, int n, int m, int nsample, const int *idx, const int *limit, const float *xyz,
                               const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud, unsigned int *hdr,
                               void *stream)
{
    return ball_pack_launch(b, b > 0 ? b : 1, n, m, nsample, idx, limit, xyz, new_xyz, rowinfo, rowdxyz, tilecloud, hdr, stream);
}

// The same for b = lists x group clouds in ONE launch: list l = clouds [l group, (l + 1) group) gets its own row list
//   rowinfo / rowdxyz  [lists][group * tiles_cap * 64], tilecloud [lists][group * tiles_cap] (cloud index inside the list),
//   hdr [lists][4]
// -- the packed row lists of every batch of a geometry group (one list per batch: each batch's kernels walk their own tiles).
extern "C" int prcnn_ball_pack_groups(int b, int group, int n, int m, int nsample, const int *idx, const int *limit, const float *xyz,
                                      const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud, unsigned int *hdr,
                                      void *stream)
{
    return ball_pack_launch(b, group, n, m, nsample, idx, limit, xyz, new_xyz, rowinfo, rowdxyz, tilecloud, hdr, stream);
}

// Every form of the above behind one entry (round 5): group (1 list = all b clouds: pass b), limit / rep / crep optional, and
// hdr_is_zero != 0: hdr [lists][4] holds zeros already (the caller zeroes an arena of headers with ONE fill): no memset here.
extern "C" int prcnn_ball_pack_ex(int b, int group, int n, int m, int nsample, const int *idx, const int *limit, const int *rep,
                                  const int *crep, const float *xyz, const float *new_xyz, unsigned int *rowinfo, float *rowdxyz,
                                  int *tilecloud, unsigned int *hdr, int hdr_is_zero, void *stream)
{
    return ball_pack_launch(b, group > 0 ? group : (b > 0 ? b : 1), n, m, nsample, idx, limit, xyz, new_xyz, rowinfo, rowdxyz, tilecloud, hdr,
                            stream, rep, crep, hdr_is_zero);
}

// The fused set-abstraction MLP over packed rows (prcnn_ball_pack): P (b,n,128) = features @ W1f^T + b1, wxyz (3,128),
// w2t (128,128), w3t (128,c3) k-major, c3 in {128, 256}; out[(b*m rows)][out_col .. out_col + c3), row stride out_stride.
// The output slice is zeroed by this call and then receives max over each centre's distinct rows of relu(layer 3).
// max_tiles = b * ceil(m * nsample / 64) sizes the grid (the real tile count stays on the device, in hdr[0]).
extern "C" int prcnn_sa_packed_mlp(int b, int n, int m, int c3, long max_tiles, const float *P, const float *wxyz,
                                   const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud,
                                   const unsigned int *hdr, const float *w2t, const float *b2, const float *w3t,
                                   const float *b3, float *out, int out_stride, int out_col, int out_is_zero, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && max_tiles >= 0, "sa_packed_mlp: bad sizes");
    PRCNN_REQUIRE(c3 == 128 || c3 == 256, "sa_packed_mlp: unsupported output width %d (128 | 256)", c3);
    PRCNN_REQUIRE(n <= 65536 && m <= 65536, "sa_packed_mlp: cloud too large for the 16-bit row descriptors");
    PRCNN_REQUIRE(out_stride >= out_col + c3 && out_col >= 0, "sa_packed_mlp: bad output slice");
    if ((long)b * m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(P && wxyz && rowinfo && rowdxyz && hdr && w2t && b2 && w3t && b3 && out, "sa_packed_mlp: null pointer");
    // tilecloud == NULL: the rows carry their cloud (descriptor (cloud << 16) | (centre << 9) | point; prcnn_rcnn_roi_geometry_packs)
    PRCNN_REQUIRE(tilecloud || (n <= 512 && m <= 128 && b <= 65536), "sa_packed_mlp: a list without tilecloud holds clouds of <= 512 points, <= 128 centres");
    PRCNN_REQUIRE((((uintptr_t)P | (uintptr_t)wxyz) & 15) == 0, "sa_packed_mlp: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (!out_is_zero && hipMemset2DAsync(out + out_col, (size_t)out_stride * sizeof(float), 0, (size_t)c3 * sizeof(float), (size_t)b * m, st) != hipSuccess) {
        set_error("sa_packed_mlp: cannot zero the output slice");
        return PRCNN_ELAUNCH;
    }
    if (max_tiles == 0) return PRCNN_OK;
    // Persistent workgroups: at most 512 (128-wide) / 256 (256-wide) of them (round 4: 1024 / 512 until then -- the sweep 256 / 384 / 512 /
    // 640 / 768 / 1024 at K = 96 gave 6546 / 6633 / 6639 / 6625 / 6638 / 6591 scenes/s, LiDAR-shaped 4752 / 4788 / 4775 / 4753 / 4729 / 4733:
    // two resident workgroups per CU hold the LDS of 512, a second round only queues), each drawing tiles from the ticket counter until
    // none is left, so the 128 weight registers per lane are loaded once per workgroup.  (Until round 2 a workgroup served at most
    // 8 tiles and the grid was sized for the worst case -- every ball full: in the sparse launches of the real step, ~2300 live
    // tiles of 102400, some 2300 DIFFERENT workgroups each fetched 128 KB of weights to serve one tile.  RCNN SA1 at the B = 8
    // shape: 244 -> 160 us.)  PRCNN_SA_GRID=<n> overrides the cap, PRCNN_SA_GRID=0 restores the 8-tiles-per-workgroup launch.
    static const int env_grid = getenv("PRCNN_SA_GRID") ? atoi(getenv("PRCNN_SA_GRID")) : 512;
    int per_wg = PK_TILES_PER_WG;
    int grid = (int)((max_tiles + per_wg - 1) / per_wg);
    if (env_grid > 0) {
        const int cap = c3 == 128 ? env_grid : (env_grid + 1) / 2;
        if (grid > cap) grid = cap;
        per_wg = 1 << 30;
    }
    unsigned int *ticket = next_ticket(st);
    if (!ticket) { set_error("sa_packed_mlp: cannot set up the tile ticket"); return PRCNN_ELAUNCH; }
    // row-major pooling through LDS needs 16-byte aligned output rows (else every tile stays on the register form)
    const bool env_lds = true;
    const int lds_pool = env_lds && (((uintptr_t)out | (uintptr_t)b3) & 15) == 0 && out_stride % 4 == 0 && out_col % 4 == 0;
    if (c3 == 128) {
        SaPkBatch batch;
        batch.nprob = 1;
        batch.p[0] = SaPkProblem{n, m, hdr, (const float4 *)rowdxyz, (const float4 *)P, (const float4 *)wxyz, rowinfo, tilecloud, w2t, b2, w3t, b3,
                                 out, out_stride, out_col, lds_pool, 128, 128};
        batch.p[1] = batch.p[0];
        hipLaunchKernelGGL(sa_packed_mlp128_kernel<1>, dim3(grid), dim3(256), 0, st, batch, ticket, per_wg);
    } else
        hipLaunchKernelGGL(sa_packed_mlp256_kernel, dim3(grid), dim3(512), 0, st, n, m, hdr, (const float4 *)rowdxyz, (const float4 *)P,
                           (const float4 *)wxyz, rowinfo, tilecloud, w2t, b2, w3t, b3, out, out_stride, out_col, ticket, per_wg, lds_pool);
    return check_launch("sa_packed_mlp");
}

// Up to two 128-wide problems (the scales of one MSG level) in ONE launch of sa_packed_mlp128_kernel: see the kernel.  Same arguments
// per problem as prcnn_sa_packed_mlp (c3 = 128), every output slice zero already or zeroed here.
extern "C" int prcnn_sa_packed_mlp_batch(int nprob, const prcnn_sa_problem *pr, void *stream)
{
    PRCNN_REQUIRE(nprob >= 1 && nprob <= 2 && pr, "sa_packed_mlp_batch: 1 or 2 problems");
    hipStream_t st = (hipStream_t)stream;
    static const int env_grid = getenv("PRCNN_SA_GRID") ? atoi(getenv("PRCNN_SA_GRID")) : 512;
    const bool env_lds = true, env_narrow = true;          // (round 6: A/B switches PRCNN_SEGMAX_LDS / PRCNN_SA_NARROW removed)
    SaPkBatch batch;
    batch.nprob = nprob;
    long most = 0;
    int narrow = 0;
    for (int i = 0; i < nprob; ++i) {
        const prcnn_sa_problem &q = pr[i];
        PRCNN_REQUIRE(q.b >= 0 && q.n >= 0 && q.m >= 0 && q.max_tiles >= 0 && q.c3 == 128, "sa_packed_mlp_batch: bad sizes (c3 = 128 only)");
        PRCNN_REQUIRE(q.n <= 65536 && q.m <= 65536, "sa_packed_mlp_batch: cloud too large for the 16-bit row descriptors");
        PRCNN_REQUIRE(q.out_stride >= q.out_col + 128 && q.out_col >= 0, "sa_packed_mlp_batch: bad output slice");
        PRCNN_REQUIRE((long)q.b * q.m > 0 && q.max_tiles > 0, "sa_packed_mlp_batch: empty problem (use prcnn_sa_packed_mlp)");
        PRCNN_REQUIRE(q.P && q.wxyz && q.rowinfo && q.rowdxyz && q.tilecloud && q.hdr && q.w2t && q.b2 && q.w3t && q.b3 && q.out, "sa_packed_mlp_batch: null pointer");
        PRCNN_REQUIRE((((uintptr_t)q.P | (uintptr_t)q.wxyz) & 15) == 0, "sa_packed_mlp_batch: 16-byte alignment required");
        if (!q.out_is_zero && hipMemset2DAsync(q.out + q.out_col, (size_t)q.out_stride * sizeof(float), 0, (size_t)128 * sizeof(float), (size_t)q.b * q.m, st) != hipSuccess) {
            set_error("sa_packed_mlp_batch: cannot zero the output slice");
            return PRCNN_ELAUNCH;
        }
        const int lds_pool = env_lds && (((uintptr_t)q.out | (uintptr_t)q.b3) & 15) == 0 && q.out_stride % 4 == 0 && q.out_col % 4 == 0;
        batch.p[i] = SaPkProblem{q.n, q.m, q.hdr, (const float4 *)q.rowdxyz, (const float4 *)q.P, (const float4 *)q.wxyz, q.rowinfo, q.tilecloud,
                                 q.w2t, q.b2, q.w3t, q.b3, q.out, q.out_stride, q.out_col, lds_pool, 128, 128};
        // the real widths of a zero-padded problem (0 = 128): 64-64 and 64-96 have kernels of their own (round 5)
        PRCNN_REQUIRE((q.c1 == 0 || (q.c1 >= 1 && q.c1 <= 128)) && (q.c2 == 0 || (q.c2 >= 1 && q.c2 <= 128)), "sa_packed_mlp_batch: bad widths c1=%d c2=%d", q.c1, q.c2);
        if (env_narrow && nprob == 2 && q.c1 == 64 && (q.c2 == 64 || q.c2 == 96)) { batch.p[i].c1 = 64; batch.p[i].c2 = q.c2; ++narrow; }
        most = q.max_tiles > most ? q.max_tiles : most;
    }
    if (nprob == 1) batch.p[1] = batch.p[0];
    int per_wg = PK_TILES_PER_WG;
    long grid1 = (most + per_wg - 1) / per_wg;
    if (env_grid > 0) {
        const long cap = (env_grid + nprob - 1) / nprob;           // the launch's workgroups together: as many as one problem's launch had
        if (grid1 > cap) grid1 = cap;
        per_wg = 1 << 30;
    }
    unsigned int *ticket = next_ticket(st);
    if (!ticket) { set_error("sa_packed_mlp_batch: cannot set up the tile ticket"); return PRCNN_ELAUNCH; }
    if (nprob == 1) hipLaunchKernelGGL(sa_packed_mlp128_kernel<1>, dim3((unsigned)grid1), dim3(256), 0, st, batch, ticket, per_wg);
    else if (narrow == 2) hipLaunchKernelGGL((sa_packed_mlp128_kernel<2, true>), dim3((unsigned)(grid1 * nprob)), dim3(256), 0, st, batch, ticket, per_wg);
    else hipLaunchKernelGGL(sa_packed_mlp128_kernel<2>, dim3((unsigned)(grid1 * nprob)), dim3(256), 0, st, batch, ticket, per_wg);
    return check_launch("sa_packed_mlp_batch");
}
