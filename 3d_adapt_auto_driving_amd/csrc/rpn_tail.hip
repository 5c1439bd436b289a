// The last stretch of the RPN over all input points, one kernel instead of nine: the finest feature-propagation module
// (pointnet2_modules.py:136-160 PointnetFPModule with no skip features: three_interpolate -> SharedMLP 256-128-128) and the
// two per-point heads on its output (rpn.py:28-50: cls 128-128-1, reg 128-128-reg_channel; Conv1d + BN folded, dropout is
// the identity in eval mode).
//
// Layer by layer these are GEMMs at the ridge of the machine (K = N = 128: 32 flop per HBM byte), and each one writes its
// activations to HBM for the next one to read back: 0.9 GB per B = 8 step for 26 GFLOP.  Here a 64-row tile stays in LDS from
// the interpolation to the last head layer; HBM sees the gathered coarse features (L2-resident table), the 128 backbone
// features, the score and the regression vector of every point -- nothing else.
//
// Arithmetic is that of the separate kernels, bit for bit, so the same oracle functions check it (oracle/mlp_oracle.c
// orc_rows_layer_mfma, orc_rows_dot; oracle/prcnn_oracle.c orc_three_interpolate):
//   interpolation  (w0*f0 + w1*f1) + w2*f2, one rounding per operation         (csrc/pointmajor.hip)
//   layers         v_mfma_f32_32x32x2_f32 over 128-deep panels, k = s on lanes 0-31 and s + 64 on lanes 32-63, panels in
//                  sequence, then + bias, then ReLU                           (csrc/packed_layer.hip)
//   score          32 lanes per row, lane l sums k = l, l+32, l+64, l+96 as one fma chain, xor butterfly 16..1, + bias
//                                                                              (csrc/packed_layer.hip rows_dot_kernel)
//
// One workgroup = 4 waves = a 64-row tile at a time, persistent over tiles (drawn from a ticket counter), ONE workgroup per CU.  Wave w owns
// output columns 32w .. 32w+31 of every layer; its 128 x 32 weight slice of the NEXT layer is fetched into registers while the
// MFMAs of the current one run (two register sets, six panel stages per tile, so the roles repeat tile after tile).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "mfma_stream.hpp"
#include "../../include/prcnn_hip.h"

namespace prcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RT_ROWS = 64;
constexpr int RT_LD = 128 + 4;
#ifndef RT_IROWS
#define RT_IROWS 4                  // rows a wave interpolates per round (3 x RT_IROWS 1 KB gathers in flight per wave)
#endif

struct RpnTailArgs {
    long rows;
    int n, m;                       // points per cloud (fine level), known points per cloud (coarse level)
    const float *known;             // (b, m, 256)
    const int *idx;                 // (rows, 3)
    const float *weight;            // (rows, 3)
    const float *wcat;              // (768,128): FP layer 1 (256 rows) | FP layer 2 | cls layer 1 | reg layer 1 | reg layer 2 (128 each,
                                    // the last one zero-padded beyond n_reg columns), all k-major with BN folded
    const float *bcat;              // (5,128): their biases
    const float *wc2, *bc2;         // cls layer 2: (128,1), (1)
    float *feats, *cls, *reg;       // (rows,128), (rows,1), (rows,n_reg)
    int n_reg;
    unsigned int *ticket;           // tile counter record of this launch (zero on entry)
    int xcd_split;                  // rpn_tail_lin: tiles drawn per XCD partition (1) or from one counter (0: fewer than 8 workgroups; round 6: no switch PRCNN_TAIL_XCD=0)
    // rpn_tail_lin_kernel<true> (round 5): the regression rows are DECODED where they stand in LDS and only the 7-float box leaves
    const float *xyz;               // (rows, 3)
    float *boxes;                   // (rows, 7)
    float loc_scope, loc_bin_size, anchor[3];
};

// fmodf(a, (float)(2 pi)) WITHOUT the library's loop (the decode below rides inside a hand-scheduled MFMA stage: no control flow).
// The remainder of two floats is exact in f32's own format, so it can be taken in f64: three reduction steps modulo b 2^80, b 2^40
// and b (b = the f32 value of 2 pi) -- each one q = trunc(r * (1 / B)), r = fma(-q, B, r): q < 2^46 is an exact integer that misses the
// true quotient by at most one, the fma's true result is a multiple of B's last bit below 2 B and therefore exact, and every step
// keeps r congruent to a modulo b -- then two corrective steps into [0, b), a's sign put back (a zero remainder takes it too, as
// fmod's does).  inf -> NaN and NaN -> NaN as fmodf.  Checked against fmodf in tests/test_gpu_packed.py through the fused decode:
// huge quotients up to 3e38, negatives, +-0, denormals, inf, NaN.
__device__ __forceinline__ float fmod_two_pi(float a)
{
    constexpr double B0 = (double)(float)(2.0 * M_PI), B1 = B0 * 1099511627776.0 /* 2^40 */, B2 = B1 * 1099511627776.0;
    constexpr double I0 = 1.0 / B0, I1 = 1.0 / B1, I2 = 1.0 / B2;
    double r = __builtin_fabs((double)a);
    r = __builtin_fma(-__builtin_trunc(r * I2), B2, r);
    r = __builtin_fma(-__builtin_trunc(r * I1), B1, r);
    r = __builtin_fma(-__builtin_trunc(r * I0), B0, r);
    r = r < 0.0 ? r + B0 : r;
    r = r < 0.0 ? r + B0 : r;
    r = r >= B0 ? r - B0 : r;
    r = r >= B0 ? r - B0 : r;
    return __builtin_copysignf((float)r, a);
}

__global__ __launch_bounds__(256, 1) void rpn_tail_kernel(const RpnTailArgs a)
{
    // T0 / T1: the working tiles of the layer chain; X0 / X1: the two 128-column panels of the interpolated input tile.  The input
    // of tile t+1 is built WHILE tile t is in its last four stages, a few instructions behind every k-group's first MFMAs
    // (RT_STAGE_HOOK): as a phase of its own the interpolation was 6.6k of a 69k-cycle tile, and a second workgroup on the CU
    // does not hide it (profiles/r02_stage_stamps.md) -- so ONE workgroup per CU, four tiles of LDS, up to 512 registers.
    __shared__ float T0[RT_ROWS * RT_LD];
    __shared__ float T1[RT_ROWS * RT_LD];
    __shared__ float X0[RT_ROWS * RT_LD];
    __shared__ float X1[RT_ROWS * RT_LD];
    __shared__ unsigned int slot[2];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    const long tiles = (a.rows + RT_ROWS - 1) / RT_ROWS;
    const unsigned int lane_off = ((unsigned int)(64 * h) * 128u + (unsigned int)(32 * w + j)) * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.wcat, 0, 768 * 128 * 4, 0x00020000);
    const float wd0 = a.wc2[j], wd1 = a.wc2[j + 32], wd2 = a.wc2[j + 64], wd3 = a.wc2[j + 96], bd = a.bc2[0];
    const f32x4 *known4 = reinterpret_cast<const f32x4 *>(a.known);
    const float bias1 = a.bcat[32 * w + j], bias2 = a.bcat[128 + 32 * w + j], biasc = a.bcat[256 + 32 * w + j],
                biasr1 = a.bcat[384 + 32 * w + j], biasr2 = a.bcat[512 + 32 * w + j];

    // neighbour indices / weights / cloud of the 16 rows this wave interpolates in tile `tt`: one coalesced load each
    // (lane l < 48 holds element l of the 16 x 3 block, lane 3q the cloud of row q)
    int nx_i, nx_cloud;
    float nx_w;
#define RT_FETCH_IDX(tt)                                                                                  \
    {                                                                                                     \
        long ge = (tt) * RT_ROWS + 16 * w + lane / 3;                                                     \
        if (ge >= a.rows) ge = a.rows - 1; /* ragged last tile (or no next tile): the last row, never stored */ \
        const long el = ge * 3 + lane % 3;                                                                \
        nx_i = lane < 48 ? a.idx[el] : 0;                                                                 \
        nx_w = lane < 48 ? a.weight[el] : 0.f;                                                            \
        nx_cloud = (int)(ge / a.n);                                                                       \
    }
    // one round of the interpolation = RT_IROWS rows of this wave: a lane owns one float4 of the 256-wide row, so every
    // neighbour row is one coalesced 1 KB read; ISSUE sends the 3 x RT_IROWS gathers, ROW finishes one row into X0 | X1
    f32x4 f[RT_IROWS][3];
#define RT_INTERP_ISSUE(rr)                                                                               \
    _Pragma("unroll") for (int q = 0; q < RT_IROWS; ++q) {                                                \
        const long cloud = __builtin_amdgcn_readlane(nx_cloud, 3 * ((rr) + q));                           \
        _Pragma("unroll") for (int e = 0; e < 3; ++e) {                                                   \
            const int i = __builtin_amdgcn_readlane(nx_i, 3 * ((rr) + q) + e);                            \
            f[q][e] = known4[(cloud * a.m + i) * 64 + lane];                                              \
        }                                                                                                 \
    }
    // (w0 f0 + w1 f1) + w2 f2 per component, one rounding per operation (the file is compiled with -ffp-contract=off):
    // v_pk_mul_f32 / v_pk_add_f32, two components per instruction
#define RT_INTERP_ROW(rr, q)                                                                              \
    {                                                                                                     \
        const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nx_w), 3 * ((rr) + (q)))); \
        const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nx_w), 3 * ((rr) + (q)) + 1)); \
        const float w2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nx_w), 3 * ((rr) + (q)) + 2)); \
        const f32x4 v = (w0 * f[q][0] + w1 * f[q][1]) + w2 * f[q][2];                                     \
        float *dst = (lane < 32 ? X0 : X1) + (16 * w + (rr) + (q)) * RT_LD + 4 * (lane & 31);             \
        *reinterpret_cast<f32x4 *>(dst) = v;                                                              \
    }
    // the side work of one stage: round rr of the NEXT tile's interpolation, gathers behind k-group 2 (the coalesced index /
    // weight loads of RT_FETCH_IDX are a stage old by then), one row finished behind each of the k-groups 10 .. 10 + RT_IROWS - 1
#define RT_SIDE(rr, g)                                                                                    \
    if ((g) == 2) { RT_INTERP_ISSUE(rr) }                                                                 \
    else if ((g) >= 10 && (g) < 10 + RT_IROWS) { RT_INTERP_ROW(rr, (g) - 10) }
#define RT_SIDE0(g) RT_SIDE(0, g)
#define RT_SIDE1(g) RT_SIDE(RT_IROWS, g)
#define RT_SIDE2(g) RT_SIDE(2 * RT_IROWS, g)
#define RT_SIDE3(g) RT_SIDE(3 * RT_IROWS, g)
    f32x4 co[4];                                           // a tile's rows on their way out, four at a time: LDS -> registers -> 512-byte rows
#define RT_ROWS_OUT(g, TT, T, dst, ld, live)                                                              \
    if ((g) == 1 || (g) == 4) {                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) co[i] = *reinterpret_cast<const f32x4 *>((T) + (r0 + 8 * (i + ((g) == 4 ? 4 : 0))) * RT_LD + 4 * chunk); \
    } else if ((g) == 3 || (g) == 6) {                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
            const unsigned int gr = (unsigned int)(TT) * RT_ROWS + r0 + 8 * (i + ((g) == 6 ? 4 : 0)); \
            if ((live) && gr < (unsigned int)a.rows) *reinterpret_cast<f32x4 *>((dst) + (gr * (unsigned int)(ld) + 4u * chunk)) = co[i]; \
        }                                                                                             \
    }
    static_assert(RT_IROWS == 4, "four stages carry four rounds of four rows");

    // Tiles come from a ticket counter, not a static stride: the step runs this kernel next to other streams' kernels
    // (sampling chains hold 32 CUs for milliseconds), and a workgroup slowed down by a neighbour must not stretch the launch.
    if (tid == 0) { slot[0] = atomicAdd(a.ticket, 1u); }
    __syncthreads();
    long t = __builtin_amdgcn_readfirstlane((int)slot[0]);   // wave-uniform: tile arithmetic stays in scalar registers
    float wa[64], wb[64];
    f32x16 acc0, acc1;
    RT_LOAD_W(wa, rs, 0)
    if (t < tiles) {
        // the first tile's input: the whole interpolation in front of its first stage (every later one rides on the previous tile)
        RT_FETCH_IDX(t)
#pragma unroll
        for (int rr = 0; rr < 16; rr += RT_IROWS) {
            RT_INTERP_ISSUE(rr)
#pragma unroll
            for (int q = 0; q < RT_IROWS; ++q) RT_INTERP_ROW(rr, q)
        }
    }
    long tp = 0;                                               // the previous tile of this workgroup
    bool last = false;
    for (unsigned int served = 0; t < tiles; ++served) {
        // the next tile's ticket: drawn now, published by the barrier below, read behind the first layer
        if (tid == 0) slot[(served + 1) & 1] = atomicAdd(a.ticket, 1u);
        RT_VM_DRAIN                                            // (also: this tile's first panel, fetched during the last stage)
        lds_barrier();                                         // X0 | X1 hold this tile's input
        // ---- FP layer 1, panel 0 (wa) while panel 1 (wb) comes in
#define RT_REG_OUT(g) RT_ROWS_OUT(g, tp, T1, a.reg, a.n_reg, served > 0 && 4 * chunk < a.n_reg)
        RT_STAGE_HOOK(X0, wa, wb, rs, 128, true, RT_REG_OUT)  // side: the previous tile's regression rows go out
        // ---- FP layer 1, panel 1 (wb) while layer 2 (wa) comes in
        RT_VM_DRAIN
        RT_STAGE(X1, wb, wa, rs, 256, false)
        lds_barrier();                                         // every wave has read both input panels: X0 | X1 are free
        const long tn = __builtin_amdgcn_readfirstlane((int)slot[(served + 1) & 1]);
        RT_FETCH_IDX(tn)
        RT_EPILOGUE(T0, bias1, true)
        lds_barrier();
        // ---- FP layer 2 (wa) while cls layer 1 (wb) comes in; its output = the backbone features.  Side: round 0 of the next input
        RT_VM_DRAIN
        RT_STAGE_HOOK(T0, wa, wb, rs, 384, true, RT_SIDE0)
        RT_EPILOGUE(T1, bias2, true)
        lds_barrier();
        // ---- cls layer 1 (wb) while reg layer 1 (wa) comes in; the feature rows go out meanwhile.  Side: round 1
        RT_VM_DRAIN
#define RT_SIDE1F(g) RT_SIDE1(g) RT_ROWS_OUT(g, t, T1, a.feats, 128, true)
        RT_STAGE_HOOK(T1, wb, wa, rs, 512, true, RT_SIDE1F)
        RT_EPILOGUE(T0, biasc, true)
        lds_barrier();
        // ---- reg layer 1 (wa) while reg layer 2 (wb) comes in.  Side: round 2 of the next input, and the score = cls layer 2, a
        //      128-long dot product per row over the cls hidden rows in T0 (this stage reads T1): 8 rows per half-wave pass, the 8
        //      passes side by side -- 5 rounds of 8 independent lane exchanges instead of 8 dependent chains of 5 (same additions
        //      per row, in the same order) -- one piece per k-group
        RT_VM_DRAIN
        float sd[8];
#define RT_SCORE(g)                                                                                       \
        if ((g) == 4) {                                                                                   \
            _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                               \
                const float *ar = T0 + (r0 + 8 * p) * RT_LD + j;                                          \
                float v = fmaf(ar[0], wd0, 0.f);                                                          \
                v = fmaf(ar[32], wd1, v);                                                                 \
                v = fmaf(ar[64], wd2, v);                                                                 \
                sd[p] = fmaf(ar[96], wd3, v);                                                             \
            }                                                                                             \
        } else if ((g) >= 5 && (g) <= 9) {                                                                \
            float o[8];                                                                                   \
            _Pragma("unroll") for (int p = 0; p < 8; ++p) o[p] = __shfl_xor(sd[p], 16 >> ((g) - 5), 32);  \
            _Pragma("unroll") for (int p = 0; p < 8; ++p) sd[p] = __fadd_rn(sd[p], o[p]);                 \
        } else if ((g) == 14) {                                                                           \
            _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                               \
                const unsigned int gr = (unsigned int)t * RT_ROWS + r0 + 8 * p;                           \
                if (j == 0 && gr < (unsigned int)a.rows) a.cls[gr] = __fadd_rn(sd[p], bd);                \
            }                                                                                             \
        }
#define RT_SIDE2S(g) RT_SIDE2(g) RT_SCORE(g)
        RT_STAGE_HOOK(T1, wa, wb, rs, 640, true, RT_SIDE2S)
        lds_barrier();                                         // every wave has read the cls hidden rows
        RT_EPILOGUE(T0, biasr1, true)
        lds_barrier();
        // ---- reg layer 2 (wb, no activation) while the next tile's first panel (wa) comes in.  Side: round 3
        RT_VM_DRAIN
        RT_STAGE_HOOK(T0, wb, wa, rs, 0, true, RT_SIDE3)
        RT_EPILOGUE(T1, biasr2, false)
        tp = t;                                                // its regression rows leave T1 during the next tile's first stage
        t = tn;                                                // (the barrier at the top of the loop publishes T1 and X0 | X1)
        last = true;
    }
    if (last) {                                                // the last tile's regression rows
        lds_barrier();
        if (4 * chunk < a.n_reg) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = r0 + 8 * i;
                const unsigned int g = (unsigned int)tp * RT_ROWS + row;
                if (g < (unsigned int)a.rows)
                    *reinterpret_cast<f32x4 *>(a.reg + (g * (unsigned int)a.n_reg + 4u * chunk)) =
                        *reinterpret_cast<const f32x4 *>(T1 + row * RT_LD + 4 * chunk);
            }
        }
    }
    if (tid == 0) ticket_release(a.ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}


// ---- the same stretch with the FIRST layer of the FP module applied at the coarse level (round 3) ----------------------------------
// relu(W1 interp(f) + b1) = relu(interp(W1 f) + b1): the layer is linear in front of its ReLU and the interpolation is a weighted
// sum.  G = f @ W1 is computed over the 4096 coarse points of a scene (a quarter of the rows: prcnn_packed_layer), and this kernel
// interpolates the 128-wide G instead of the 256-wide f, adds the bias, applies the ReLU and starts at layer 2: four panel stages
// per tile instead of six, half the gathered bytes (the 16 MB table of G stays in L2; the 33 MB one of f did not).  Not the
// reference's association of the sums (~1e-7 relative); the oracle stand-in (oracle/ext_cpu.py rpn_tail_lin_wrapper) restates
// THIS order -- (w0 g0 + w1 g1) + w2 g2, + b1, ReLU, then the layers in the MFMA kernels' k order -- and is matched bit for bit.
// A 128-wide row is 32 float4 lanes: a wave interpolates TWO rows per gather instruction (lanes 0-31 / 32-63), 8 such sets per tile;
// the next tile's input is built behind the MFMAs of all four stages into the input panel the running tile does not use (X0 / X1
// alternate).
// DECODE (round 5): the regression head's 76 outputs of a row are decoded where they stand in LDS (tile T1) -- arg-max over the
// 12 x bins / 12 z bins / 12 heading bins, residual look-ups, anchor sizes: decode_bbox_target of lib/utils/bbox_transform.py:24-121 with
// get_xz_fine, then proposal_layer.py:31's y shift, in rpn_decode_kernel's f32 operation order (csrc/proposal.hip) -- and the 7-float box
// leaves instead of the 304-byte row: the 80 MB `reg` tensor of a 16-scene launch and the kernel that read it back with one lane per
// row (537 MB of HBM traffic for 85 MB of algorithmic bytes) are gone.  Four lanes per row (x | z | heading | y + sizes),
// 16 rows per wave, spread over four k-groups of the NEXT tile's first stage like the row stores it replaces; no control flow.
// The layout is the shipped one only (12 + 12 + 12 + 12 + 1 + 12 + 12 + 3 = 76 channels: every cfgs/*.yaml); other layouts keep `reg`.
constexpr int RD_NB = 12;           // per_loc_bin_num = 2 * int(LOC_SCOPE / LOC_BIN_SIZE) = 2 * int(3.0 / 0.5), = NUM_HEAD_BIN
// NARROW (round 5, 64 < n_reg <= 80: the 76 channels of the shipped configurations): the last stage computes 80 columns instead of a
// zero-padded 128 -- until then 52 of its 128 columns (10 % of the kernel's MFMAs) multiplied padding.  Columns 0..63 = two 32-column
// blocks x two 32-row blocks, ONE block per wave (v_mfma_f32_32x32x2_f32, the k order of every other layer: the same bits as
// before); columns 64..79 = a 16-column block, 16 rows per wave, on v_mfma_f32_16x16x4_f32 -- bitwise a chain of fused multiply-adds
// too (profiles/mfma16_probe.hip: 0 of 51 200 outputs differ), step s = 0..31 over k = s, 32 + s, 64 + s, 96 + s in that order
// (oracle/mlp_oracle.c orc_rows_layer_mfma16).  96 MFMA issues of 64 / 32 cycles per wave instead of 128 of 64: 5120 cycles, not 8192.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool DECODE, bool NARROW>
__global__ __launch_bounds__(256, 1) void rpn_tail_lin_kernel(const RpnTailArgs a)
{
    __shared__ float T0[RT_ROWS * RT_LD];
    __shared__ float T1[RT_ROWS * RT_LD];
    __shared__ float X0[RT_ROWS * RT_LD];
    __shared__ float X1[RT_ROWS * RT_LD];
    __shared__ unsigned int slot[2];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    const long tiles = (a.rows + RT_ROWS - 1) / RT_ROWS;
    const unsigned int lane_off = ((unsigned int)(64 * h) * 128u + (unsigned int)(32 * w + j)) * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.wcat, 0, 512 * 128 * 4, 0x00020000);
    const float wd0 = a.wc2[j], wd1 = a.wc2[j + 32], wd2 = a.wc2[j + 64], wd3 = a.wc2[j + 96], bd = a.bc2[0];
    const f32x4 *known4 = reinterpret_cast<const f32x4 *>(a.known);       // G (b, m, 128)
    const f32x4 bias1v = *reinterpret_cast<const f32x4 *>(a.bcat + 4 * j);
    const float bias2 = a.bcat[128 + 32 * w + j], biasc = a.bcat[256 + 32 * w + j], biasr1 = a.bcat[384 + 32 * w + j],
                biasr2 = a.bcat[512 + 32 * w + j];
    // NARROW last stage: wave w = 32 x 32 block (row block w & 1, column block w >> 1) + rows 16 w .. 16 w + 15 of the 16-column block
    const int nrb = w & 1, ncb = w >> 1;
    const unsigned int lane_off_n = ((unsigned int)(64 * h) * 128u + (unsigned int)(32 * ncb + j)) * 4u;          // k = s + 64 h, column 32 ncb + j
    const unsigned int lane_off_x = ((unsigned int)(32 * (lane >> 4)) * 128u + (unsigned int)(64 + (lane & 15))) * 4u;   // k = s + 32 (lane / 16), column 64 + lane % 16
    const float biasn = a.bcat[512 + 32 * ncb + j], biasx = a.bcat[512 + 64 + (lane & 15)];
    int nx_i, nx_cloud;
    float nx_w;
    f32x4 fl[2][3];
    // set s of a tile = rows 16 w + 2 s (lanes 0-31) and 16 w + 2 s + 1 (lanes 32-63) of this wave
#define RL_ISSUE(sl, s)                                                                                   \
    {                                                                                                     \
        const int q_ = 2 * (s) + h;                                                                       \
        const long cloud_ = __shfl(nx_cloud, 3 * q_);                                                     \
        _Pragma("unroll") for (int e = 0; e < 3; ++e) {                                                   \
            const int i_ = __shfl(nx_i, 3 * q_ + e);                                                      \
            fl[sl][e] = known4[(cloud_ * a.m + i_) * 32 + j];                                             \
        }                                                                                                 \
    }
#define RL_ROW(sl, s, XN)                                                                                 \
    {                                                                                                     \
        const int q_ = 2 * (s) + h;                                                                       \
        const float w0_ = __shfl(nx_w, 3 * q_), w1_ = __shfl(nx_w, 3 * q_ + 1), w2_ = __shfl(nx_w, 3 * q_ + 2); \
        f32x4 v_ = ((w0_ * fl[sl][0] + w1_ * fl[sl][1]) + w2_ * fl[sl][2]) + bias1v;                      \
        v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); \
        *reinterpret_cast<f32x4 *>((XN) + (16 * w + q_) * RT_LD + 4 * j) = v_;                             \
    }
    // round r (two sets) of the next tile's input behind one stage: both sets' gathers at k-group gi, one set finished at gf, gf + 1
#define RL_SIDE(r, g, gi, gf)                                                                             \
    if ((g) == (gi)) { RL_ISSUE(0, 2 * (r)) RL_ISSUE(1, 2 * (r) + 1) }                                    \
    else if ((g) == (gf)) { RL_ROW(0, 2 * (r), Xn) }                                                      \
    else if ((g) == (gf) + 1) { RL_ROW(1, 2 * (r) + 1, Xn) }
    f32x4 co[4];
    // ---- DECODE: lane = (row 16 w + lane / 4, part lane % 4); part 0: x, 1: z, 2: heading, 3: y + sizes
    const int dpart = lane & 3;
    const float *drow = T1 + (16 * w + (lane >> 2)) * RT_LD;
    const int dstart = dpart == 0 ? 0 : dpart == 1 ? RD_NB : dpart == 2 ? 4 * RD_NB + 1 : 6 * RD_NB + 4 - RD_NB;   // 12 values from here: bins | bins | bins | .. h w l
    const int dexb = dpart == 0 ? 2 * RD_NB : dpart == 1 ? 3 * RD_NB : dpart == 2 ? 5 * RD_NB + 1 : 4 * RD_NB;     // + bin: residual (part 3: the y offset)
    const int dxo = dpart == 0 ? 0 : dpart == 1 ? 2 : dpart == 2 ? 0 : 1;                                          // which coordinate of the point
    const int dbo = dpart == 0 ? 0 : dpart == 1 ? 2 : dpart == 2 ? 6 : 1;                                          // which slot of the box
    float dv[RD_NB], dp = 0.f, dex = 0.f, dout = 0.f, dh = 0.f, dw = 0.f, dl = 0.f;
    int dbi = 0;
#define RL_DEC_READ(TT)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < RD_NB; ++i) dv[i] = drow[dstart + i];                        \
        unsigned int gr_ = (unsigned int)(TT) * RT_ROWS + 16 * w + (lane >> 2);                           \
        if (gr_ >= (unsigned int)a.rows) gr_ = (unsigned int)a.rows - 1u;                                  \
        dp = a.xyz[gr_ * 3u + dxo];                                                                        \
    }
    // first maximum; a NaN counts as the largest value (torch.argmax; proposal.hip argmax_row)
#define RL_DEC_ARGMAX                                                                                     \
    {                                                                                                     \
        float bv_ = dv[0];                                                                                \
        dbi = 0;                                                                                          \
        _Pragma("unroll") for (int i = 1; i < RD_NB; ++i) {                                                \
            const bool tk_ = dv[i] > bv_ || (dv[i] != dv[i] && bv_ == bv_);                                \
            bv_ = tk_ ? dv[i] : bv_;                                                                      \
            dbi = tk_ ? i : dbi;                                                                          \
        }                                                                                                 \
        dex = drow[dexb + (dpart < 3 ? dbi : 0)];                                                          \
    }
#define RL_DEC_MATH                                                                                       \
    {                                                                                                     \
        const float fb_ = (float)dbi, bin_ = a.loc_bin_size;                                               \
        float loc_ = __fsub_rn(__fadd_rn(__fmul_rn(fb_, bin_), bin_ / 2), a.loc_scope);                    \
        loc_ = __fadd_rn(__fadd_rn(loc_, __fmul_rn(dex, bin_)), dp);                                       \
        const float apc_ = (float)((2.0 * M_PI) / RD_NB), apch_ = (float)(((2.0 * M_PI) / RD_NB) / 2.0);   \
        const float two_pi_ = (float)(2.0 * M_PI), pi_ = (float)M_PI;                                      \
        float ry_ = fmod_two_pi(__fadd_rn(__fmul_rn(fb_, apc_), __fmul_rn(dex, apch_)));           \
        ry_ = (ry_ != 0.f && ry_ < 0.f) ? __fadd_rn(ry_, two_pi_) : ry_;                                   \
        ry_ = ry_ > pi_ ? __fsub_rn(ry_, two_pi_) : ry_;                                                   \
        dh = __fadd_rn(__fmul_rn(dv[RD_NB - 3], a.anchor[0]), a.anchor[0]);                                \
        dw = __fadd_rn(__fmul_rn(dv[RD_NB - 2], a.anchor[1]), a.anchor[1]);                                \
        dl = __fadd_rn(__fmul_rn(dv[RD_NB - 1], a.anchor[2]), a.anchor[2]);                                \
        const float y_ = __fadd_rn(__fadd_rn(dp, dex), dh / 2);                                            \
        dout = dpart < 2 ? loc_ : dpart == 2 ? ry_ : y_;                                                   \
    }
#define RL_DEC_STORE(TT, live)                                                                            \
    {                                                                                                     \
        const unsigned int gr_ = (unsigned int)(TT) * RT_ROWS + 16 * w + (lane >> 2);                     \
        if ((live) && gr_ < (unsigned int)a.rows) {                                                        \
            float *o_ = a.boxes + gr_ * 7u;                                                                \
            o_[dbo] = dout;                                                                                \
            if (dpart == 3) { o_[3] = dh; o_[4] = dw; o_[5] = dl; }                                        \
        }                                                                                                 \
    }
#define RL_DECODE(g, TT, live)                                                                            \
    if ((g) == 1) RL_DEC_READ(TT)                                                                         \
    else if ((g) == 3) RL_DEC_ARGMAX                                                                      \
    else if ((g) == 5) RL_DEC_MATH                                                                        \
    else if ((g) == 7) RL_DEC_STORE(TT, live)

    // tiles by XCD: one eighth of the rows (one scene of a batch of 8) and its 2 MB of G per L2 instead of all 16.8 MB through every L2
    XcdTickets tk;
    tk.rec = a.ticket; tk.tiles = (unsigned int)tiles; tk.dead = 0u;
    tk.chunk = a.xcd_split ? (unsigned int)((tiles + 7) / 8) : (unsigned int)tiles;
    tk.xcd = a.xcd_split ? (blockIdx.x & 7u) : 0u;
    if (tid == 0) { slot[0] = tk.issue(); }
    __syncthreads();
    long t = __builtin_amdgcn_readfirstlane((int)slot[0]);
    float wa[64], wb[64];
    f32x16 acc0, acc1;
    RT_LOAD_W(wa, rs, 0)
    if (t < tiles) {
        RT_FETCH_IDX(t)
        float *Xn = X0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            RL_ISSUE(0, 2 * r) RL_ISSUE(1, 2 * r + 1)
            RL_ROW(0, 2 * r, Xn) RL_ROW(1, 2 * r + 1, Xn)
        }
    }
    long tp = 0;
    bool last = false;
    for (unsigned int served = 0; t < tiles; ++served) {
        // the next tile's ticket from this workgroup's XCD partition (a select on the returned value, no control flow: any branch
        // in this hand-scheduled loop cost 10 % of the kernel when tried, so a workgroup does not go stealing from other partitions)
        if (tid == 0) slot[(served + 1) & 1] = tk.issue();
        RT_VM_DRAIN
        lds_barrier();                                         // this tile's input panel, the previous tile's regression rows in T1, the ticket
        const float *Xc = (served & 1) ? X1 : X0;
        float *Xn = (served & 1) ? X0 : X1;
        const long tn = __builtin_amdgcn_readfirstlane((int)slot[(served + 1) & 1]);
        RT_FETCH_IDX(tn)                                       // consumed from k-group 8 of the first stage on
        // ---- FP layer 2 (wa) while cls layer 1 (wb) comes in; side: the previous tile's regression rows leave T1, round 0
#define RL_H1R(g) RT_ROWS_OUT(g, tp, T1, a.reg, a.n_reg, served > 0 && 4 * chunk < a.n_reg) RL_SIDE(0, g, 8, 13)
#define RL_H1D(g) RL_DECODE(g, tp, served > 0) RL_SIDE(0, g, 8, 13)
        if (DECODE) { RT_STAGE_HOOK(Xc, wa, wb, rs, 128, true, RL_H1D) }
        else { RT_STAGE_HOOK(Xc, wa, wb, rs, 128, true, RL_H1R) }
        lds_barrier();                                         // every wave has taken the old rows out of T1
        RT_EPILOGUE(T1, bias2, true)                           // = the backbone features
        lds_barrier();
        // ---- cls layer 1 (wb) while reg layer 1 (wa) comes in; side: the feature rows go out, round 1
        RT_VM_DRAIN
#define RL_H2(g) RT_ROWS_OUT(g, t, T1, a.feats, 128, true) RL_SIDE(1, g, 2, 10)
        RT_STAGE_HOOK(T1, wb, wa, rs, 256, true, RL_H2)
        RT_EPILOGUE(T0, biasc, true)
        lds_barrier();
        // ---- reg layer 1 (wa) while reg layer 2 (wb) comes in; side: the score (cls layer 2 over the hidden rows in T0), round 2
        RT_VM_DRAIN
        float sd[8];
#define RL_H3(g) RT_SCORE(g) RL_SIDE(2, g, 2, 10)
        RT_STAGE_HOOK_LO(T1, wa, wb, rs, 384, true, RL_H3, (NARROW ? lane_off_n : lane_off))
        lds_barrier();                                         // every wave has read the cls hidden rows
        RT_EPILOGUE(T0, biasr1, true)
        lds_barrier();
        // ---- reg layer 2 (wb, no activation) while the next tile's layer 2 (wa) comes in; side: round 3
        RT_VM_DRAIN
#define RL_H4(g) RL_SIDE(3, g, 2, 10)
        if (NARROW) {
            float wx[32];
            f32x4 accx;
            {
                const float *ap = T0 + (32 * nrb + j) * RT_LD + 64 * h;
                f32x4 a0 = *reinterpret_cast<const f32x4 *>(ap);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    f32x4 n0 = a0;
                    if (g < 15) n0 = *reinterpret_cast<const f32x4 *>(ap + 4 * (g + 1));
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 0) {
                        const f32x16 zero = {0};
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wb[0], zero, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wb[4 * g + 0], acc0, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)          // the next tile's first stage (FP layer 2): this wave's usual slice
                        wa[4 * g + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, (unsigned int)(4 * g + q) * 512u, 0));
#pragma unroll
                    for (int q = 0; q < 2; ++q)          // the 16-column block's weights, consumed behind this loop
                        wx[2 * g + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane_off_x, (unsigned int)(384 + 2 * g + q) * 512u, 0));
                    __builtin_amdgcn_sched_barrier(0);
                    RL_H4(g)
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wb[4 * g + 1], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wb[4 * g + 2], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wb[4 * g + 3], acc0, 0, 0, 0);
                    a0 = n0;
                }
                const float *xp = T0 + (16 * w + (lane & 15)) * RT_LD + 32 * (lane >> 4);
                f32x4 x = *reinterpret_cast<const f32x4 *>(xp);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x4 nx = x;
                    if (q < 7) nx = *reinterpret_cast<const f32x4 *>(xp + 4 * (q + 1));
                    if (q == 0) {
                        const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
                        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wx[0], zero4, 0, 0, 0);
                    } else {
                        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wx[4 * q + 0], accx, 0, 0, 0);
                    }
                    accx = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, wx[4 * q + 1], accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, wx[4 * q + 2], accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, wx[4 * q + 3], accx, 0, 0, 0);
                    x = nx;
                }
            }
            // + bias (no activation) -> T1: this wave's 32 x 32 block, then its 16 rows of columns 64..79 (columns 80.. keep what the
            // FP layer left there: nobody reads them)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * nrb + (r & 3) + 8 * (r >> 2) + 4 * h;
                T1[row * RT_LD + 32 * ncb + j] = acc0[r] + biasn;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) T1[(16 * w + 4 * (lane >> 4) + r) * RT_LD + 64 + (lane & 15)] = accx[r] + biasx;
        } else {
            RT_STAGE_HOOK(T0, wb, wa, rs, 0, true, RL_H4)
            RT_EPILOGUE(T1, biasr2, false)
        }
        tp = t;
        t = tn;
        last = true;
    }
    if (last) {
        lds_barrier();
        if (DECODE) {
            RL_DEC_READ(tp) RL_DEC_ARGMAX RL_DEC_MATH RL_DEC_STORE(tp, true)
        } else if (4 * chunk < a.n_reg) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = r0 + 8 * i;
                const unsigned int g = (unsigned int)tp * RT_ROWS + row;
                if (g < (unsigned int)a.rows)
                    *reinterpret_cast<f32x4 *>(a.reg + (g * (unsigned int)a.n_reg + 4u * chunk)) =
                        *reinterpret_cast<const f32x4 *>(T1 + row * RT_LD + 4 * chunk);
            }
        }
    }
    if (tid == 0) tk.release();                      // the launch's last workgroup zeroes the record for its next user
}

}  // namespace prcnn

namespace prcnn {
unsigned int *next_ticket(hipStream_t st);    // sa_mlp_fused.hip

__global__ __launch_bounds__(256) void fmod_two_pi_selftest_kernel(long n, const float *__restrict__ a, float *__restrict__ mine,
                                                                   float *__restrict__ lib)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    mine[i] = fmod_two_pi(a[i]);
    lib[i] = fmodf(a[i], (float)(2.0 * M_PI));
}
}

using namespace prcnn;

extern "C" int prcnn_rpn_tail(int b, int n, int m, const float *known, const int *idx, const float *weight, const float *wcat,
                              const float *bcat, const float *wc2, const float *bc2, int n_reg, float *feats, float *cls,
                              float *reg, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 1 && n_reg >= 4 && n_reg <= 128 && n_reg % 4 == 0,
                  "rpn_tail: bad sizes (n_reg=%d must be a multiple of 4 in 4..128)", n_reg);
    const long rows = (long)b * n;
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rows <= (1L << 23), "rpn_tail: too many points (32-bit element offsets)");
    PRCNN_REQUIRE(known && idx && weight && wcat && bcat && wc2 && bc2 && feats && cls && reg, "rpn_tail: null pointer");
    PRCNN_REQUIRE((((uintptr_t)known | (uintptr_t)feats | (uintptr_t)wcat | (uintptr_t)reg) & 15) == 0,
                  "rpn_tail: 16-byte alignment required");
    RpnTailArgs a;
    a.rows = rows; a.n = n; a.m = m; a.known = known; a.idx = idx; a.weight = weight;
    a.wcat = wcat; a.bcat = bcat; a.wc2 = wc2; a.bc2 = bc2; a.feats = feats; a.cls = cls; a.reg = reg; a.n_reg = n_reg;
    a.ticket = next_ticket((hipStream_t)stream);
    if (!a.ticket) { set_error("rpn_tail: cannot set up the tile ticket"); return PRCNN_ELAUNCH; }
    a.xcd_split = 0;
    const long tiles = (rows + RT_ROWS - 1) / RT_ROWS;
    const long cap = mfma_grid_cap() < 256 ? mfma_grid_cap() : 256;               // gfx950: 256 CUs, one resident workgroup each (135 KB of LDS)
    const long grid = tiles < cap ? tiles : cap;
    hipLaunchKernelGGL(rpn_tail_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("rpn_tail");
}


/* prcnn_rpn_tail with the FP module's first layer already applied at the coarse level (csrc/rpn_tail.hip rpn_tail_lin_kernel):
 * G (b,m,128) = coarse features @ the layer's weights (no bias); wcat (512,128) = [FP layer 2 | cls layer 1 | reg layer 1 | reg layer 2],
 * bcat (5,128) = the biases of FP layer 1 (added after the interpolation) and of those four.
 * boxes != NULL (prcnn_rpn_tail_lin_boxes): the regression rows are decoded in the kernel and `reg` is not written (may be NULL). */
static int rpn_tail_lin_any(int b, int n, int m, const float *G, const int *idx, const float *weight, const float *wcat,
                            const float *bcat, const float *wc2, const float *bc2, int n_reg, float *feats, float *cls,
                            float *reg, const float *xyz, float *boxes, float loc_scope, float loc_bin_size, const float *anchor,
                            void *stream, const char *who)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 1 && n_reg >= 4 && n_reg <= 128 && n_reg % 4 == 0,
                  "%s: bad sizes (n_reg=%d must be a multiple of 4 in 4..128)", who, n_reg);
    const long rows = (long)b * n;
    if (rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rows <= (1L << 23), "%s: too many points (32-bit element offsets)", who);
    PRCNN_REQUIRE(G && idx && weight && wcat && bcat && wc2 && bc2 && feats && cls && (reg || boxes), "%s: null pointer", who);
    PRCNN_REQUIRE((((uintptr_t)G | (uintptr_t)feats | (uintptr_t)wcat | (uintptr_t)reg | (uintptr_t)bcat) & 15) == 0,
                  "%s: 16-byte alignment required", who);
    RpnTailArgs a;
    a.rows = rows; a.n = n; a.m = m; a.known = G; a.idx = idx; a.weight = weight;
    a.wcat = wcat; a.bcat = bcat; a.wc2 = wc2; a.bc2 = bc2; a.feats = feats; a.cls = cls; a.reg = reg; a.n_reg = n_reg;
    a.xyz = xyz; a.boxes = boxes; a.loc_scope = loc_scope; a.loc_bin_size = loc_bin_size;
    for (int i = 0; i < 3; ++i) a.anchor[i] = anchor ? anchor[i] : 0.f;
    a.ticket = next_ticket((hipStream_t)stream);
    if (!a.ticket) { set_error("%s: cannot set up the tile ticket", who); return PRCNN_ELAUNCH; }
    const int xcd_split = 1;                               // (round 6: A/B switch PRCNN_TAIL_XCD removed)
    const long tiles = (rows + RT_ROWS - 1) / RT_ROWS;
    // PRCNN_TAIL_GRID (tuning): workgroups of the fused tail.  Its waves hold a SIMD's whole register file, so while it runs on all 256 CUs no
    // other stream's kernel makes progress; measured in round 6 (bench.py, alternating): 224 / 192 workgroups 8412 / 8443 scenes/s at K = 100
    // against 8347-8361 with 256 (the tail itself a third longer at 192), 160: 8273; at K = 20 level within the windows' spread -- default kept
    static const long tail_cap = getenv("PRCNN_TAIL_GRID") && atol(getenv("PRCNN_TAIL_GRID")) > 0 ? atol(getenv("PRCNN_TAIL_GRID")) : 256;
    const long cap = mfma_grid_cap() < tail_cap ? mfma_grid_cap() : tail_cap;
    const long grid = tiles < cap ? tiles : cap;
    // a workgroup draws only from partition blockIdx.x & 7 (no stealing): every non-empty partition needs a workgroup of its own --
    // with fewer than 8 workgroups for 8 or more tiles (PRCNN_MFMA_GRID < 8) the tiles come from one counter instead
    a.xcd_split = xcd_split && (grid >= 8 || grid == tiles);
    // the narrow last stage (80 columns instead of 128) for 64 < n_reg <= 80; PRCNN_TAIL_NARROW=0: the padded stage everywhere (another
    // k order in columns 64.. -- a numerics switch: ~1e-7 relative)
    static const bool narrow_ok = !(getenv("PRCNN_TAIL_NARROW") && atoi(getenv("PRCNN_TAIL_NARROW")) == 0);
    const bool narrow = narrow_ok && n_reg > 64 && n_reg <= 80;
    if (boxes && narrow) hipLaunchKernelGGL((rpn_tail_lin_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (boxes) hipLaunchKernelGGL((rpn_tail_lin_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (narrow) hipLaunchKernelGGL((rpn_tail_lin_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((rpn_tail_lin_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch(who);
}

extern "C" int prcnn_rpn_tail_lin(int b, int n, int m, const float *G, const int *idx, const float *weight, const float *wcat,
                                  const float *bcat, const float *wc2, const float *bc2, int n_reg, float *feats, float *cls,
                                  float *reg, void *stream)
{
    PRCNN_REQUIRE(reg || (long)b * n == 0, "rpn_tail_lin: null pointer");
    return rpn_tail_lin_any(b, n, m, G, idx, weight, wcat, bcat, wc2, bc2, n_reg, feats, cls, reg, nullptr, nullptr, 0.f, 0.f, nullptr,
                            stream, "rpn_tail_lin");
}

/* prcnn_rpn_tail_lin with the proposal layer's decode inside (round 5): boxes (b*n,7) = decode_bbox_target(xyz, reg, get_xz_fine = True,
 * get_y_by_bin = False, get_ry_fine = False) with y += h / 2 (bbox_transform.py:24-121, proposal_layer.py:23-31) -- what rpn_decode_kernel
 * of csrc/proposal.hip computes from the stored rows, operation for operation -- and the regression rows themselves are not
 * written.  Served layout: LOC_SCOPE / LOC_BIN_SIZE = 6 bins per side (12 x bins, 12 z bins), NUM_HEAD_BIN = 12, LOC_XZ_FINE: the 76
 * channels of every shipped configuration (prcnn_rpn_tail_boxes_supported); anchor_size_host = (h, w, l) in host memory. */
/* test hook: out_mine[i] = the branch-free fmod of the fused decode, out_lib[i] = fmodf(a[i], (float)(2 pi)) on the same device */
extern "C" int prcnn_selftest_fmod_two_pi(long n, const float *a, float *out_mine, float *out_lib, void *stream)
{
    PRCNN_REQUIRE(n >= 0, "selftest_fmod_two_pi: bad size");
    if (n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(a && out_mine && out_lib, "selftest_fmod_two_pi: null pointer");
    hipLaunchKernelGGL(fmod_two_pi_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, a, out_mine, out_lib);
    return check_launch("selftest_fmod_two_pi");
}

extern "C" int prcnn_rpn_tail_boxes_supported(int channels, float loc_scope, float loc_bin_size, int num_head_bin, int xz_fine)
{
    return loc_bin_size > 0.f && (int)(loc_scope / loc_bin_size) * 2 == RD_NB && num_head_bin == RD_NB && xz_fine &&
           channels == 6 * RD_NB + 4;
}

extern "C" int prcnn_rpn_tail_lin_boxes(int b, int n, int m, const float *G, const int *idx, const float *weight, const float *wcat,
                                        const float *bcat, const float *wc2, const float *bc2, int n_reg, float loc_scope,
                                        float loc_bin_size, int num_head_bin, int xz_fine, const float *anchor_size_host,
                                        const float *xyz, float *feats, float *cls, float *boxes, void *stream)
{
    PRCNN_REQUIRE(prcnn_rpn_tail_boxes_supported(n_reg, loc_scope, loc_bin_size, num_head_bin, xz_fine),
                  "rpn_tail_lin_boxes: regression layout (%d channels, scope %g / bin %g, %d heading bins, xz_fine %d) not served",
                  n_reg, loc_scope, loc_bin_size, num_head_bin, xz_fine);
    PRCNN_REQUIRE(anchor_size_host && ((xyz && boxes) || (long)b * n == 0), "rpn_tail_lin_boxes: null pointer");
    return rpn_tail_lin_any(b, n, m, G, idx, weight, wcat, bcat, wc2, bc2, n_reg, feats, cls, nullptr, xyz, boxes, loc_scope, loc_bin_size,
                            anchor_size_host, stream, "rpn_tail_lin_boxes");
}
