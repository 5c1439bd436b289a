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
        float mx = fmaxf(acc0[0], acc1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(acc0[r], acc1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (h == 0) pk_flush(out, ctr[0], out_stride, col, mx, bias);
        return;
    }
    float cur = acc0[0];
#pragma unroll
    for (int q = 1; q < 32; ++q) {
        const float v = (q < 16) ? acc0[q & 15] : acc1[q & 15];
        const int prev = pk_row(q - 1), row = pk_row(q);
        // a boundary between this lane's previous row and this one: any segment start in (prev, row] (+4 for half 1)
        const unsigned long long m0 = pk_bits_upto(row) & ~pk_bits_upto(prev);
        const unsigned long long m1 = pk_bits_upto(row + 4) & ~pk_bits_upto(prev + 4);
        const bool b0 = (start & m0) != 0, b1 = (start & m1) != 0;        // wave-uniform
        if (b0 | b1) {
            const bool mine = h ? b1 : b0;
            if (mine) {
                pk_flush(out, ctr[prev + 4 * h], out_stride, col, cur, bias);
                cur = v;
            } else {
                cur = fmaxf(cur, v);
            }
        } else {
            cur = fmaxf(cur, v);
        }
    }
    pk_flush(out, ctr[pk_row(31) + 4 * h], out_stride, col, cur, bias);
}


}  // namespace prcnn
