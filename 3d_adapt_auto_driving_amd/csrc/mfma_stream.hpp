// mfma_stream.hpp -- the panel-stage macros of the kernels that keep a 64-row tile in LDS across several 128-wide layers and
// stream each layer's 128 x 32 weight slice per wave from L2 (csrc/rpn_tail.hip, rcnn_entrance_kernel in csrc/rcnn_point_mlp.hip).
//
// Names the macros expect in scope: RT_LD (LDS row stride in floats), lane_off (this lane's byte offset into a 128-column
// weight matrix: ((64 h) * 128 + 32 w + j) * 4), j / h / w (lane & 31, lane >> 5, wave), acc0 / acc1 (f32x16 accumulators of rows
// 0-31 / 32-63), and the weight matrix's buffer resource passed as RS / rs.  Summation order: oracle/mlp_oracle.c (orc_rows_layer_mfma).
#pragma once

// s_waitcnt vmcnt(0): said explicitly before every prefetch so that the compiler's wait-count bookkeeping knows nothing older
// is outstanding and puts no wait between the prefetch and the MFMAs that hide it (a vmcnt above 63 cannot be encoded)
#define RT_VM_DRAIN __builtin_amdgcn_s_waitcnt(0x0F70);
#define RT_LOAD_W(dst, rs, krow)                                                                          \
    _Pragma("unroll") for (int s = 0; s < 64; ++s)                                                        \
        dst[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, (unsigned int)((krow) + s) * 512u, 0)); \
    __builtin_amdgcn_sched_barrier(0);
// One 128-deep panel: 16 k-groups of 8 MFMAs.  A wave issues in order, so everything that is not an MFMA is placed where
// the matrix pipe is busy anyway:
//   * the A operands of group g+1 are read from LDS before the MFMAs of group g (a ds_read in front of its first use idles
//     the pipe for an LDS round trip per group);
//   * the 64 weight loads of the NEXT panel stage (wn <- rows krow.. of the weight buffer) go out four per group, behind
//     this group's first MFMAs, instead of 64 in a row in front of the stage (their issue alone was ~10 % of a stage);
//   * FIRST: the accumulators start from the inline constant 0 in the first MFMA (no 32 v_mov per stage).
#define RT_STAGE_HOOK_LO(T, wf, wn, RS, krow, FIRST, HOOK, LOFF)                                                     \
    {                                                                                                     \
        f32x4 a0 = *reinterpret_cast<const f32x4 *>((T) + j * RT_LD + 64 * h);                            \
        f32x4 a1 = *reinterpret_cast<const f32x4 *>((T) + (32 + j) * RT_LD + 64 * h);                     \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                  \
            f32x4 n0 = a0, n1 = a1;                                                                       \
            if (g < 15) {                                                                                 \
                n0 = *reinterpret_cast<const f32x4 *>((T) + j * RT_LD + 64 * h + 4 * (g + 1));            \
                n1 = *reinterpret_cast<const f32x4 *>((T) + (32 + j) * RT_LD + 64 * h + 4 * (g + 1));     \
            }                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if ((FIRST) && g == 0) {                                                                      \
                const f32x16 zero = {0};                                                                  \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wf[0], zero, 0, 0, 0);                  \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wf[0], zero, 0, 0, 0);                  \
            } else {                                                                                      \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wf[4 * g + 0], acc0, 0, 0, 0);          \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wf[4 * g + 0], acc1, 0, 0, 0);          \
            }                                                                                             \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                 \
                wn[4 * g + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(                     \
                    RS, (LOFF), (unsigned int)((krow) + 4 * g + q) * 512u, 0));                           \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            HOOK(g)                                                                                       \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wf[4 * g + 1], acc0, 0, 0, 0);              \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, wf[4 * g + 1], acc1, 0, 0, 0);              \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wf[4 * g + 2], acc0, 0, 0, 0);              \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, wf[4 * g + 2], acc1, 0, 0, 0);              \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wf[4 * g + 3], acc0, 0, 0, 0);              \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, wf[4 * g + 3], acc1, 0, 0, 0);              \
            a0 = n0; a1 = n1;                                                                             \
        }                                                                                                 \
    }
//   * HOOK(g): side work of the caller issued behind the first two MFMAs of k-group g (g is a compile-time constant after
//     unrolling): a wave has ~14 free issue cycles behind every MFMA (profiles/r02_stage_stamps.md), enough for a few
//     instructions per group that would otherwise sit in a phase of their own in front of a stage
//   * LOFF: the lane's byte offset for the NEXT stage's weight slice (RT_STAGE_HOOK: lane_off, the slice this wave owns in every layer;
//     rpn_tail_lin_kernel's narrow last stage gives its waves other column blocks)
#define RT_STAGE_HOOK(T, wf, wn, RS, krow, FIRST, HOOK) RT_STAGE_HOOK_LO(T, wf, wn, RS, krow, FIRST, HOOK, lane_off)
#define RT_NO_HOOK(g)
#define RT_STAGE(T, wf, wn, RS, krow, FIRST) RT_STAGE_HOOK(T, wf, wn, RS, krow, FIRST, RT_NO_HOOK)
// act(acc + bias) of this wave's 64 x 32 block -> tile T (the next layer's A operand)
#define RT_EPILOGUE(T, bias, RELU)                                                                        \
    {                                                                                                     \
        const float bcol = (bias);                                                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                  \
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;                                               \
            const float v0 = acc0[r] + bcol, v1 = acc1[r] + bcol;                                         \
            (T)[row * RT_LD + 32 * w + j] = (RELU) ? fmaxf(v0, 0.f) : v0;                                 \
            (T)[(32 + row) * RT_LD + 32 * w + j] = (RELU) ? fmaxf(v1, 0.f) : v1;                          \
        }                                                                                                 \
    }

