// sa_wide.hip -- one scale of a set-abstraction level whose layers are wider than the register-resident fused kernels take
// (RPN SA3 / SA4: 128-196-256, 256-256-512, 256-384-512, pointrcnn/lib/config.py:58-61; the RCNN's GroupAll level 256-256-512,
// rcnn_net.py:64-92), over the DISTINCT grouped rows (csrc/sa_packed.hip ball_pack), in ONE kernel:
//
//   rows of a half tile (32 packed rows)  ->  layer 1 = relu(P[point] + wxyz . (xyz[point] - centre))      (builder, VALU)
//                                          ->  layer 2 (C1 -> C2)  ->  layer 3 (C2 -> C3) + max over each centre's rows
//
// It replaces three launches per scale (packed_gather_affine, packed_layer, packed_layer_segmax: csrc/packed_layer.hip) whose
// activations went through HBM and which, on sparse levels, were each a latency-bound launch for a handful of live tiles.
// The 32 rows stay in LDS from the builder to the pooling (A1: C1 columns, Y1: C2 columns, 128-column panels of 132-float rows);
// every (output column block, K panel) pair is one 64-MFMA stage of a wave (32 rows x 32 columns, one accumulator), and each
// stage's 128 x 32 weight slice per wave streams in from L2 behind the MFMAs of the stage before it (two register sets
// alternate; the stage count is even for every supported shape, so the roles repeat unit after unit).  Persistent workgroups
// draw (tile, half) units from a ticket counter.
//
// Arithmetic per row = the three separate kernels, bit for bit: builder fma chain as packed_gather_affine_kernel, k panels in
// ascending order into one accumulator (v_mfma_f32_32x32x2_f32, k = s on lanes 0-31, s + 64 on lanes 32-63), + bias, ReLU;
// the max is order-independent.  oracle: orc_gather_affine_fma + orc_rows_layer_mfma (oracle/mlp_oracle.c).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "sa_wide.hpp"
#include "../../include/prcnn_hip.h"

namespace prcnn {

struct SaWideArgs {
    int n, m, c1, c2, c3;
    const unsigned int *hdr;
    const float4 *rowdxyz;
    const float *P;                  // (b, n, c1)
    const float *wxyz;               // (3, c1)
    const unsigned int *rowinfo;
    const int *tilecloud;
    const float *w2, *b2, *w3, *b3;  // (c1, c2), (c2), (c2, c3), (c3), k-major
    float *out;
    int out_stride, out_col;
    unsigned int *ticket;
};

__global__ __launch_bounds__(256, 2) void sa_wide_fused_kernel(const SaWideArgs a)
{
    extern __shared__ float sw_lds[];
    const int kp1 = a.c1 >> 7, kp2 = a.c2 >> 7, nb3 = a.c3 >> 7;          // panels of A1, of Y1 (= column blocks of layer 2), blocks of layer 3
    float *A1 = sw_lds, *Y1 = sw_lds + kp1 * SW_PANEL;
    int *ctr = reinterpret_cast<int *>(Y1 + kp2 * SW_PANEL);               // centre of every row
    unsigned int *slot = reinterpret_cast<unsigned int *>(ctr + SW_R);
    // the two bias vectors in LDS (round 4): read at every epilogue -- six per unit -- they were global loads issued at their point of
    // use, each an exposed round trip (~1 us of a unit's ~25) with the matrix pipe idle
    float *sb2 = reinterpret_cast<float *>(slot + 2), *sb3 = sb2 + a.c2;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    const long units = 2L * (long)a.hdr[0];
    const int ns2 = kp1 * kp2, ns = ns2 + kp2 * nb3;                       // stages of layer 2, of both layers (even)
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, a.c1 * a.c2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w3, 0, a.c2 * a.c3 * 4, 0x00020000);
    const unsigned int rb2 = (unsigned int)a.c2 * 4u, rb3 = (unsigned int)a.c3 * 4u;
    const unsigned int voff2 = ((unsigned int)(64 * h) * (unsigned int)a.c2 + (unsigned int)(32 * w + j)) * 4u;
    const unsigned int voff3 = ((unsigned int)(64 * h) * (unsigned int)a.c3 + (unsigned int)(32 * w + j)) * 4u;
    const float4 *P4 = reinterpret_cast<const float4 *>(a.P);
    const float4 *wx4 = reinterpret_cast<const float4 *>(a.wxyz);
    const int q1 = a.c1 >> 2;                                              // float4 chunks per row of P

    if (tid == 0) slot[0] = atomicAdd(a.ticket, 1u);
    for (int i = tid; i < a.c2; i += 256) sb2[i] = a.b2[i];
    for (int i = tid; i < a.c3; i += 256) sb3[i] = a.b3[i];
    __syncthreads();
    long u = __builtin_amdgcn_readfirstlane((int)slot[0]);
    if (u >= units) { if (tid == 0) ticket_release(a.ticket); return; }
    float wa[64], wb[64];
    f32x16 acc;
    // stage 0 of every unit: layer 2, column block 0, K panel 0
#pragma unroll
    for (int s = 0; s < 64; ++s) wa[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, voff2, (unsigned int)s * rb2, 0));

    for (unsigned int served = 0; u < units; ++served) {
        const long t = u >> 1;
        const long row0 = t * 64 + 32 * (u & 1);
        // ---- builder: layer 1 of the unit's 32 rows, panel by panel
        {
            const int cloud = a.tilecloud[t];
            const long pbase = (long)cloud * a.n;
            unsigned int info[4];
            float4 d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                info[i] = a.rowinfo[row0 + r0 + 8 * i];
                d[i] = a.rowdxyz[row0 + r0 + 8 * i];
            }
            if (tid == 0) slot[(served + 1) & 1] = atomicAdd(a.ticket, 1u);
            if (tid < SW_R) ctr[tid] = cloud * a.m + (int)(a.rowinfo[row0 + tid] >> 16);
            for (int pc = 0; pc < kp1; ++pc) {
                const float4 wx = wx4[pc * 32 + chunk], wy = wx4[q1 + pc * 32 + chunk], wz = wx4[2 * q1 + pc * 32 + chunk];
                float4 base[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) base[i] = P4[(pbase + (long)(info[i] & 0xffffu)) * q1 + pc * 32 + chunk];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dx = d[i].x, dy = d[i].y, dz = d[i].z;
                    const float4 v = affine_relu4(base[i], wx, wy, wz, dx, dy, dz);
                    *reinterpret_cast<float4 *>(A1 + pc * SW_PANEL + (r0 + 8 * i) * SW_LD + 4 * chunk) = v;
                }
            }
        }
        SW_VM_DRAIN
        lds_barrier();
        // ---- the stages, two at a time (wa -> wb -> wa): s < ns2: layer 2 (block s / kp1, panel s % kp1), else layer 3
        for (int s = 0; s < ns; s += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int sc = s + half;                                   // this stage
                const bool l3 = sc >= ns2;
                const int kpn = l3 ? kp2 : kp1;
                const int si = l3 ? sc - ns2 : sc;
                const int nb = si / kpn, kp = si - nb * kpn;
                const float *T = (l3 ? Y1 : A1) + kp * SW_PANEL;
                // the stage after it (wraps to stage 0 of the next unit)
                const int sn = sc + 1 == ns ? 0 : sc + 1;
                const bool n3 = sn >= ns2;
                const int kpnn = n3 ? kp2 : kp1;
                const int sni = n3 ? sn - ns2 : sn;
                const int nbn = sni / kpnn, kpn2 = sni - nbn * kpnn;
                const unsigned int rbn = n3 ? rb3 : rb2;
                const unsigned int voffn = n3 ? voff3 : voff2;
                const unsigned int soffn = (unsigned int)(kpn2 * 128) * rbn + (unsigned int)nbn * 512u;
                if (l3 && kp == 0 && nb == 0) lds_barrier();               // every wave has written its columns of Y1
                if (half == 0) {
                    if (n3) { SW_STAGE(T, wa, wb, rs3, voffn, soffn, rbn, kp == 0) } else { SW_STAGE(T, wa, wb, rs2, voffn, soffn, rbn, kp == 0) }
                } else {
                    if (n3) { SW_STAGE(T, wb, wa, rs3, voffn, soffn, rbn, kp == 0) } else { SW_STAGE(T, wb, wa, rs2, voffn, soffn, rbn, kp == 0) }
                }
                // the next stage's weights have arrived (the last of them went out five k-groups ago); waited for HERE, in front of the
                // epilogue, and not in front of the next stage as until round 4: the pooling's atomics then stay in flight behind the
                // next stage's MFMAs instead of being waited for with the matrix pipe idle (vmcnt counts them like loads)
                SW_VM_DRAIN
                if (kp == kpn - 1) {
                    if (!l3) {
                        const float bcol = sb2[nb * 128 + 32 * w + j];
                        float *Y = Y1 + nb * SW_PANEL;
#pragma unroll
                        for (int r = 0; r < 16; ++r) Y[((r & 3) + 8 * (r >> 2) + 4 * h) * SW_LD + 32 * w + j] = fmaxf(acc[r] + bcol, 0.f);
                    } else {
                        const int myc = ctr[j], prevc = ctr[j ? j - 1 : 0];
                        const unsigned int start = (unsigned int)__ballot(lane < 32 && (lane == 0 || myc != prevc));
                        sw_segmented_max(acc, ctr, start, h, a.out, a.out_stride, a.out_col + nb * 128 + 32 * w + j, sb3[nb * 128 + 32 * w + j]);
                    }
                }
            }
        }
        const long un = __builtin_amdgcn_readfirstlane((int)slot[(served + 1) & 1]);
        lds_barrier();                                                     // A1 / Y1 / ctr are free for the next unit
        u = un;
    }
    if (tid == 0) ticket_release(a.ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}

}  // namespace prcnn

using namespace prcnn;

namespace prcnn {
unsigned int *next_ticket(hipStream_t st);    // sa_mlp_fused.hip
}

extern "C" int prcnn_sa_wide_fused_supported(int c1, int c2, int c3)
{
    if (c1 <= 0 || c2 <= 0 || c3 <= 0 || c1 % 128 || c2 % 128 || c3 % 128) return 0;
    const int kp1 = c1 / 128, kp2 = c2 / 128, nb3 = c3 / 128;
    if ((kp1 * kp2 + kp2 * nb3) % 2) return 0;                             // the two weight register sets alternate
    return (size_t)(kp1 + kp2) * SW_PANEL * sizeof(float) <= 96 * 1024 ? 1 : 0;
}

// P (b,n,c1) per-point part of layer 1, wxyz (3,c1); row list of prcnn_ball_pack; w2 (c1,c2), w3 (c2,c3) k-major, widths
// multiples of 128 (prcnn_sa_wide_fused_supported); out[(b*m)][out_col .. out_col + c3) with row stride out_stride, zeroed here
// unless out_is_zero.
extern "C" int prcnn_sa_wide_fused(int b, int n, int m, int c1, int c2, int c3, long max_tiles, const float *P, const float *wxyz,
                                   const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud, const unsigned int *hdr,
                                   const float *w2, const float *b2, const float *w3, const float *b3, float *out, int out_stride,
                                   int out_col, int out_is_zero, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && max_tiles >= 0, "sa_wide_fused: bad sizes");
    PRCNN_REQUIRE(prcnn_sa_wide_fused_supported(c1, c2, c3), "sa_wide_fused: unsupported widths %d-%d-%d", c1, c2, c3);
    PRCNN_REQUIRE(n <= 65536 && m <= 65536, "sa_wide_fused: cloud too large for the 16-bit row descriptors");
    PRCNN_REQUIRE(out_stride >= out_col + c3 && out_col >= 0, "sa_wide_fused: bad output slice");
    if ((long)b * m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(P && wxyz && rowinfo && rowdxyz && tilecloud && hdr && w2 && b2 && w3 && b3 && out, "sa_wide_fused: null pointer");
    PRCNN_REQUIRE((((uintptr_t)P | (uintptr_t)wxyz | (uintptr_t)w2 | (uintptr_t)w3) & 15) == 0, "sa_wide_fused: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (!out_is_zero && hipMemset2DAsync(out + out_col, (size_t)out_stride * sizeof(float), 0, (size_t)c3 * sizeof(float), (size_t)b * m, st) != hipSuccess) {
        set_error("sa_wide_fused: cannot zero the output slice");
        return PRCNN_ELAUNCH;
    }
    if (max_tiles == 0) return PRCNN_OK;
    const size_t lds = (size_t)(c1 / 128 + c2 / 128) * SW_PANEL * sizeof(float) + SW_R * sizeof(int) + 2 * sizeof(unsigned int) +
                       (size_t)(c2 + c3) * sizeof(float);
    const int rc = ensure_dynamic_lds((const void *)sa_wide_fused_kernel, lds, "sa_wide_fused");
    if (rc != PRCNN_OK) return rc;
    SaWideArgs a;
    a.n = n; a.m = m; a.c1 = c1; a.c2 = c2; a.c3 = c3; a.hdr = hdr; a.rowdxyz = (const float4 *)rowdxyz; a.P = P; a.wxyz = wxyz;
    a.rowinfo = rowinfo; a.tilecloud = tilecloud; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.out = out; a.out_stride = out_stride;
    a.out_col = out_col;
    a.ticket = next_ticket(st);
    if (!a.ticket) { set_error("sa_wide_fused: cannot set up the unit ticket"); return PRCNN_ELAUNCH; }
    const long units = 2 * max_tiles;
    const long grid = units < mfma_grid_cap() ? units : mfma_grid_cap();
    hipLaunchKernelGGL(sa_wide_fused_kernel, dim3((unsigned)grid), dim3(256), lds, st, a);
    return check_launch("sa_wide_fused");
}
