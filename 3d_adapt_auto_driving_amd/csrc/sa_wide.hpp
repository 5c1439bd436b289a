// sa_wide.hpp -- what csrc/sa_wide.hip (gathered layer 1, layers 2-3 + pool) and csrc/sa_wide3.hip (all three layers + pool) share: the
// 64-MFMA panel stage of a 32-row unit and the segmented max of a unit's accumulator block.
// Names the stage macro expects in scope: j / h (lane & 31, lane >> 5), acc (f32x16), SW_LD.
#pragma once
#include <hip/hip_runtime.h>

namespace prcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SW_R = 32;            // rows per unit
constexpr int SW_LD = 128 + 4;      // floats per LDS row of a 128-column panel
constexpr int SW_PANEL = SW_R * SW_LD;

// max over each centre's rows of this lane's 16 accumulator rows -> atomicMax(out[centre][col]) (values >= 0 after ReLU);
// row of register r: (r & 3) + 8 (r >> 2) + 4 h.  start: bit i set when row i begins a new centre (wave-uniform).
__device__ __forceinline__ void sw_segmented_max(const f32x16 &acc, const int *ctr, unsigned int start, int h, float *__restrict__ out,
                                                 int out_stride, int col, float bias)
{
    auto rowof = [](int q) { return (q & 3) + 8 * (q >> 2); };
    auto upto = [](int row) { return row >= 31 ? 0xffffffffu : ((2u << row) - 1u); };
    auto flush = [&](int row, float v) {
        atomicMax(reinterpret_cast<int *>(out + (long)ctr[row] * out_stride + col), __float_as_int(fmaxf(v + bias, 0.f)));
    };
    float cur = acc[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
        const float v = acc[q];
        const int prev = rowof(q - 1), row = rowof(q);
        const unsigned int m0 = upto(row) & ~upto(prev), m1 = upto(row + 4) & ~upto(prev + 4);
        const bool b0 = (start & m0) != 0, b1 = (start & m1) != 0;       // wave-uniform
        if (b0 | b1) {
            if (h ? b1 : b0) {
                flush(prev + 4 * h, cur);
                cur = v;
            } else {
                cur = fmax_raw(cur, v);
            }
        } else {
            cur = fmax_raw(cur, v);
        }
    }
    flush(rowof(15) + 4 * h, cur);
}

#define SW_VM_DRAIN __builtin_amdgcn_s_waitcnt(0x0F70);
// one stage: acc (+)= T[32][128] @ wf; meanwhile wn <- the next stage's weights (resource RS, byte offsets VOFF + SOFF + s * RB)
#define SW_STAGE(T, wf, wn, RS, VOFF, SOFF, RB, FIRST)                                                    \
    {                                                                                                     \
        const float *ap = (T) + j * SW_LD + 64 * h;                                                       \
        f32x4 a = *reinterpret_cast<const f32x4 *>(ap);                                                   \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                  \
            f32x4 nx = a;                                                                                 \
            if (g < 15) nx = *reinterpret_cast<const f32x4 *>(ap + 4 * (g + 1));                          \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if ((FIRST) && g == 0) {                                                                      \
                const f32x16 zero = {0};                                                                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wf[0], zero, 0, 0, 0);                    \
            } else {                                                                                      \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wf[4 * g + 0], acc, 0, 0, 0);             \
            }                                                                                             \
            _Pragma("unroll") for (int q = 0; q < 6; ++q)      /* six per k-group: all 64 are out by group 10 */ \
                if (6 * g + q < 64)                                                                       \
                    wn[6 * g + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(                 \
                        RS, VOFF, (SOFF) + (unsigned int)(6 * g + q) * (RB), 0));                         \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wf[4 * g + 1], acc, 0, 0, 0);                 \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wf[4 * g + 2], acc, 0, 0, 0);                 \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wf[4 * g + 3], acc, 0, 0, 0);                 \
            a = nx;                                                                                       \
        }                                                                                                 \
    }

}  // namespace prcnn
