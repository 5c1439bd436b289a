// segmax.hpp -- segmented max over the rows of a 64-row MFMA accumulator tile (shared by sa_packed.hip and packed_layer.hip).
#pragma once
#include "common.hpp"

namespace prcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Rows held by lane half h of the 32x32 MFMA accumulators, in increasing row order: q = 0..31 ->
// accumulator q >> 4, register q & 15, row = 32 (q >> 4) + (r & 3) + 8 (r >> 2) + 4 h.
__device__ __forceinline__ constexpr int pk_row(int q) { return 32 * (q >> 4) + ((q & 15) & 3) + 8 * ((q & 15) >> 2); }
__device__ __forceinline__ constexpr unsigned long long pk_bits_upto(int row) { return row >= 63 ? ~0ULL : ((2ULL << row) - 1ULL); }

// out[centre][col] = max(out[centre][col], v) for v >= 0 (integer order of the bit patterns == float order)
__device__ __forceinline__ void pk_flush(float *__restrict__ out, long centre, int out_stride, int col, float v, float bias)
{
    atomicMax(reinterpret_cast<int *>(out + centre * out_stride + col), __float_as_int(fmaxf(v + bias, 0.f)));
}

// acc0 = rows 0..31, acc1 = rows 32..63 of this wave's 32 output columns; ctr[64] = centre of every row (LDS);
// start = bit i set when row i begins a new centre (wave-uniform).
__device__ __forceinline__ void pk_segmented_max(const f32x16 &acc0, const f32x16 &acc1, const int *ctr, unsigned long long start,
                                                 int h, float *__restrict__ out, int out_stride, int col, float bias)
{
    if (start == 1ULL) {
        // the whole tile belongs to ONE centre (a full ball): plain max over the 64 rows, one store per column
        float mx = fmax_raw(acc0[0], acc1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmax_raw(mx, fmax_raw(acc0[r], acc1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (h == 0) pk_flush(out, ctr[0], out_stride, col, mx, bias);
        return;
    }
    float cur = acc0[0];
    // half 1 holds the rows of half 0 + 4: its boundary mask m1 is m0 << 4, so one shifted copy of `start` per lane serves both
    // halves with the compile-time constant m0 (as `h ? b1 : b0` the compiler kept 31 per-lane 64-bit masks in registers and
    // spilled them: a scratch reload + full wait per step of this chain)
    const unsigned long long sh = h ? (start >> 4) : start;
#pragma unroll
    for (int q = 1; q < 32; ++q) {
        const float v = (q < 16) ? acc0[q & 15] : acc1[q & 15];
        const int prev = pk_row(q - 1), row = pk_row(q);
        // a boundary between this lane's previous row and this one: any segment start in (prev, row] (+4 for half 1)
        const unsigned long long m0 = pk_bits_upto(row) & ~pk_bits_upto(prev);
        const unsigned long long m1 = pk_bits_upto(row + 4) & ~pk_bits_upto(prev + 4);
        const bool b0 = (start & m0) != 0, b1 = (start & m1) != 0;        // wave-uniform
        if (b0 | b1) {
            const bool mine = (sh & m0) != 0;
            if (mine) {
                pk_flush(out, ctr[prev + 4 * h], out_stride, col, cur, bias);
                cur = v;
            } else {
                cur = fmax_raw(cur, v);
            }
        } else {
            cur = fmax_raw(cur, v);
        }
    }
    pk_flush(out, ctr[pk_row(31) + 4 * h], out_stride, col, cur, bias);
}



// ---- the same segmented max THROUGH an LDS tile, for tiles that hold many centres --------------------------------------------
// The register form above walks a lane's 32 rows one after the other and pays one flush (centre lookup, address, bias, ReLU,
// atomic) per centre and lane: on the sparse levels of a KITTI-shaped scene a tile holds 10-60 centres, and the segmented max
// was 20k of a tile's 53k cycles (profiles/r02_stage_stamps.md, table 4).  Here the waves park their accumulators in a 64 x 128
// LDS tile and the workgroup re-reads it ROW-major: thread (chunk, g) owns the float4 of columns 4 chunk .. 4 chunk + 3 in rows
// 8 g .. 8 g + 7, so a flush is ONE 16-byte store of a 512-byte output row per 32 lanes, and a centre whose rows lie inside the
// thread's eight rows -- nearly all of them on sparse tiles -- needs no atomic at all: only segments that touch the border of
// the row group (and may continue in the neighbouring group or tile) go through atomicMax.  max is exact, bias + ReLU are
// applied to the maximum as before: same bits.
//
#ifndef PK_LDS_MIN
#define PK_LDS_MIN 3                // tiles with more centres than this pool through LDS, the others in registers
#endif
// Z: LDS tile, row stride ld floats (>= 128, rows 16-byte aligned); ctr[64]: output row of every tile row (LDS).
// Call with all 256 threads of the workgroup AFTER a barrier that follows the waves' pk_park() calls.
__device__ __forceinline__ void pk_park(const f32x16 &acc0, const f32x16 &acc1, float *Z, int ld, int col, int h)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        Z[row * ld + col] = acc0[r];
        Z[(32 + row) * ld + col] = acc1[r];
    }
}

__device__ __forceinline__ void pk_flush4(float *__restrict__ out, long centre, int out_stride, int col, float4 v, const float4 bias, bool shared)
{
    v.x = fmaxf(v.x + bias.x, 0.f); v.y = fmaxf(v.y + bias.y, 0.f); v.z = fmaxf(v.z + bias.z, 0.f); v.w = fmaxf(v.w + bias.w, 0.f);
    float *dst = out + centre * out_stride + col;
    if (shared) {
        atomicMax(reinterpret_cast<int *>(dst), __float_as_int(v.x));
        atomicMax(reinterpret_cast<int *>(dst + 1), __float_as_int(v.y));
        atomicMax(reinterpret_cast<int *>(dst + 2), __float_as_int(v.z));
        atomicMax(reinterpret_cast<int *>(dst + 3), __float_as_int(v.w));
    } else {
        *reinterpret_cast<float4 *>(dst) = v;               // (out, out_stride and the column are multiples of 4 floats: checked by the host)
    }
}

// tid: thread of the 256; col0: output column of the tile's column 0; bias4: bias of this thread's four columns
__device__ __forceinline__ void pk_segmented_max_lds(const float *Z, int ld, const int *ctr, int tid, float *__restrict__ out,
                                                     int out_stride, int col0, const float4 bias4)
{
    const int chunk = tid & 31, row0 = 8 * (tid >> 5);
    const int before = row0 > 0 ? ctr[row0 - 1] : -1;       // -1: the tile's border -- the segment may continue in another tile
    const int after = row0 + 8 < 64 ? ctr[row0 + 8] : -1;
    int c_cur = ctr[row0];
    float4 cur = *reinterpret_cast<const float4 *>(Z + row0 * ld + 4 * chunk);
    bool first = true;                                       // the running segment started at the group's first row
#pragma unroll 1                                             // (unrolled, the eight row loads are hoisted: 40 more live registers in kernels that have none to spare)
    for (int i = 1; i < 8; ++i) {
        const int c = ctr[row0 + i];
        const float4 v = *reinterpret_cast<const float4 *>(Z + (row0 + i) * ld + 4 * chunk);
        if (c != c_cur) {                                    // uniform over the 32 lanes of a row group
            pk_flush4(out, c_cur, out_stride, col0 + 4 * chunk, cur, bias4, first && (row0 == 0 || before == c_cur));
            c_cur = c; cur = v; first = false;
        } else {
            cur.x = fmax_raw(cur.x, v.x); cur.y = fmax_raw(cur.y, v.y); cur.z = fmax_raw(cur.z, v.z); cur.w = fmax_raw(cur.w, v.w);
        }
    }
    pk_flush4(out, c_cur, out_stride, col0 + 4 * chunk, cur, bias4,
              (first && (row0 == 0 || before == c_cur)) || row0 + 8 == 64 || after == c_cur);
}

}  // namespace prcnn
