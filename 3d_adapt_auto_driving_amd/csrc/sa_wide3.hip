// sa_wide3.hip -- a dense wide set-abstraction scale (the RCNN's GroupAll level 256-256-512 over the 32 sampled points of every RoI,
// rcnn_net.py:64-92, pointrcnn/lib/config.py:118-120) with ALL THREE layers in one kernel:
//
//   feature rows of a unit's 32 packed rows (gathered, a plain copy)  ->  layer 1 (C0 -> C1) on the matrix cores, and in its epilogue
//   + bias, + wxyz . (xyz[point] - centre), ReLU                      ->  layer 2 (C1 -> C2)  ->  layer 3 (C2 -> C3) + max over each centre's rows
//
// csrc/sa_wide.hip takes layer 1's per-point part P = f @ W1 + b1 from a launch of its own (prcnn_packed_layer over all b * n points) and
// applies the coordinate part in its builder.  Where every point is grouped exactly once -- GroupAll -- that launch computes nothing the
// scale would not compute itself, costs a launch on the feature stream (48 us alone, 120 us beside the other streams for 3.4 GFLOP) and
// sends 26 MB of P through HBM and back.  Here the unit's feature rows are layer 1's A operand.
//
// Arithmetic = the separate kernels, bit for bit: P = (k panels in ascending order into one accumulator, v_mfma_f32_32x32x2_f32 with
// k = s on lanes 0-31 and s + 64 on lanes 32-63) + b1 as csrc/packed_layer.hip computes it, then fma(wz, dz, fma(wy, dy, fma(wx, dx, P))),
// ReLU as packed_gather_affine_kernel / sa_wide_fused_kernel's builder; layers 2 and 3 as there.  oracle: orc_rows_layer_mfma +
// orc_gather_affine_fma (oracle/mlp_oracle.c); stand-in for the CPU shadow run: oracle/ext_cpu.py sa_wide_fused3_wrapper.
//
// LDS: the input panels (C0 / 128) and layer 2's output panels (C2 / 128) share one region -- the input is dead once every wave has
// left layer 1 -- beside layer 1's output panels: 4 panels = 66 KB for 256-256-256-512, two workgroups per CU.  The three weight
// matrices come as ONE buffer (w1 | w2 | w3, k-major): one buffer resource, the next stage's slice is a scalar offset.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "sa_wide.hpp"
#include "../../include/prcnn_hip.h"

namespace prcnn {

struct SaWide3Args {
    int n, m, c0, c1, c2, c3;
    const unsigned int *hdr;
    const float4 *rowdxyz;
    const float *F;                  // (b, n, c0) point features
    const float *wxyz;               // (3, c1)
    const unsigned int *rowinfo;
    const int *tilecloud;
    const float *wcat;               // (c0, c1) | (c1, c2) | (c2, c3), k-major
    const float *b1, *b2, *b3;
    float *out;
    int out_stride, out_col;
    unsigned int *ticket;
};

// stage sc of a unit -> its layer (0, 1, 2), output column block and K panel; the weight slice's byte offset apart from the lane's part
struct S3Stage {
    int layer, nb, kp, kpn;
    unsigned int rb, soff;
};

__global__ __launch_bounds__(256, 2) void sa_wide3_kernel(const SaWide3Args a)
{
    extern __shared__ __align__(16) float sw_lds[];
    const int kp0 = a.c0 >> 7, kp1 = a.c1 >> 7, kp2 = a.c2 >> 7, nb3 = a.c3 >> 7;
    const int kx = kp0 > kp2 ? kp0 : kp2;
    float *X = sw_lds;                                                     // layer 1's input panels, later layer 2's output panels
    float *A1 = sw_lds + kx * SW_PANEL;                                    // layer 1's output panels
    float4 *s_d = reinterpret_cast<float4 *>(A1 + kp1 * SW_PANEL);         // (dx, dy, dz, -) of every row
    int *ctr = reinterpret_cast<int *>(s_d + SW_R);                        // centre of every row
    unsigned int *slot = reinterpret_cast<unsigned int *>(ctr + SW_R);
    float *sb1 = reinterpret_cast<float *>(slot + 2), *sb2 = sb1 + a.c1, *sb3 = sb2 + a.c2, *swx = sb3 + a.c3;   // biases, (3, c1) coordinate weights
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = tid & 31, r0 = tid >> 5;
    // tilecloud == NULL (round 5): the rows carry their cloud -- descriptor (cloud << 16) | (centre << 9) | point, hdr[1] rows back to back
    // (prcnn_rcnn_roi_geometry_packs' third list; see sa_packed_mlp128_kernel) -- and a unit is 32 consecutive rows of the list
    const bool rowcloud = a.tilecloud == nullptr;
    const long nrows = (long)a.hdr[1];
    const long units = rowcloud ? (nrows + 31) >> 5 : 2L * (long)a.hdr[0];
    const long last_row = rowcloud ? nrows - 1 : 0x7fffffffffffL;
    const int ns1 = kp0 * kp1, ns2 = kp1 * kp2, ns = ns1 + ns2 + kp2 * nb3;                       // (even)
    const unsigned int o2 = (unsigned int)(a.c0 * a.c1) * 4u, o3 = o2 + (unsigned int)(a.c1 * a.c2) * 4u;
    const unsigned int total = o3 + (unsigned int)(a.c2 * a.c3) * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.wcat, 0, (int)total, 0x00020000);
    const unsigned int lane_k = (unsigned int)(64 * h), lane_c = (unsigned int)(32 * w + j) * 4u;
    const unsigned int voff1 = lane_k * (unsigned int)a.c1 * 4u + lane_c, voff2 = lane_k * (unsigned int)a.c2 * 4u + lane_c,
                       voff3 = lane_k * (unsigned int)a.c3 * 4u + lane_c;
    const float4 *F4 = reinterpret_cast<const float4 *>(a.F);
    const int q0 = a.c0 >> 2;

    auto stage_of = [&](int sc) __attribute__((always_inline)) {
        S3Stage st;
        int si;
        unsigned int base;
        if (sc < ns1) { st.layer = 0; st.kpn = kp0; si = sc; st.rb = (unsigned int)a.c1 * 4u; base = 0u; }
        else if (sc < ns1 + ns2) { st.layer = 1; st.kpn = kp1; si = sc - ns1; st.rb = (unsigned int)a.c2 * 4u; base = o2; }
        else { st.layer = 2; st.kpn = kp2; si = sc - ns1 - ns2; st.rb = (unsigned int)a.c3 * 4u; base = o3; }
        st.nb = si / st.kpn;
        st.kp = si - st.nb * st.kpn;
        st.soff = base + (unsigned int)(st.kp * 128) * st.rb + (unsigned int)st.nb * 512u;
        return st;
    };

    if (tid == 0) slot[0] = atomicAdd(a.ticket, 1u);
    for (int i = tid; i < a.c1; i += 256) sb1[i] = a.b1[i];
    for (int i = tid; i < a.c2; i += 256) sb2[i] = a.b2[i];
    for (int i = tid; i < a.c3; i += 256) sb3[i] = a.b3[i];
    for (int i = tid; i < 3 * a.c1; i += 256) swx[i] = a.wxyz[i];
    __syncthreads();
    long u = __builtin_amdgcn_readfirstlane((int)slot[0]);
    if (u >= units) { if (tid == 0) ticket_release(a.ticket); return; }
    float wa[64], wb[64];
    f32x16 acc;
    // stage 0 of every unit: layer 1, column block 0, K panel 0
#pragma unroll
    for (int s = 0; s < 64; ++s) wa[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff1, (unsigned int)s * (unsigned int)a.c1 * 4u, 0));

    for (unsigned int served = 0; u < units; ++served) {
        const long t = u >> 1;
        const long row0 = rowcloud ? u * 32 : t * 64 + 32 * (u & 1);
        // ---- builder: the unit's 32 feature rows, panel by panel (a copy: 4 rows x one 16-byte chunk per thread and panel)
        {
            const int cloud = rowcloud ? 0 : a.tilecloud[t];
            unsigned int info[4];
            long prow[4];                                                 // the rows' points as rows of F
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                info[i] = a.rowinfo[min(row0 + r0 + 8 * i, last_row)];
                prow[i] = rowcloud ? (long)(info[i] >> 16) * a.n + (long)(info[i] & 0x1ffu) : (long)cloud * a.n + (long)(info[i] & 0xffffu);
            }
            if (tid == 0) slot[(served + 1) & 1] = atomicAdd(a.ticket, 1u);
            if (tid < SW_R) {
                const unsigned int wd = a.rowinfo[min(row0 + tid, last_row)];
                ctr[tid] = rowcloud ? (int)(wd >> 16) * a.m + (int)((wd >> 9) & 0x7fu) : cloud * a.m + (int)(wd >> 16);
                s_d[tid] = a.rowdxyz[min(row0 + tid, last_row)];
            }
            for (int pc = 0; pc < kp0; ++pc) {
                float4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = F4[prow[i] * q0 + pc * 32 + chunk];
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(X + pc * SW_PANEL + (r0 + 8 * i) * SW_LD + 4 * chunk) = v[i];
            }
        }
        SW_VM_DRAIN
        lds_barrier();
        // ---- the stages, two at a time (wa -> wb -> wa)
        for (int s = 0; s < ns; s += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int sc = s + half;
                const S3Stage st = stage_of(sc);
                const S3Stage nst = stage_of(sc + 1 == ns ? 0 : sc + 1);
                const float *T = (st.layer == 0 ? X : (st.layer == 1 ? A1 : X)) + st.kp * SW_PANEL;
                const unsigned int voffn = nst.layer == 0 ? voff1 : (nst.layer == 1 ? voff2 : voff3);
                // every wave has written its columns of the layer's input (and, in front of layer 2, has left layer 1: its input panels
                // are layer 2's output panels)
                if (st.layer > 0 && st.kp == 0 && st.nb == 0) lds_barrier();
                if (half == 0) { SW_STAGE(T, wa, wb, rs, voffn, nst.soff, nst.rb, st.kp == 0) } else { SW_STAGE(T, wb, wa, rs, voffn, nst.soff, nst.rb, st.kp == 0) }
                // the next stage's weights have arrived; waited for in front of the epilogue (the pooling's atomics stay in flight
                // behind the next stage's MFMAs: csrc/sa_wide.hip)
                SW_VM_DRAIN
                if (st.kp == st.kpn - 1) {
                    const int col = st.nb * 128 + 32 * w + j;
                    if (st.layer == 0) {
                        const float bcol = sb1[col], wxc = swx[col], wyc = swx[a.c1 + col], wzc = swx[2 * a.c1 + col];
                        float *Y = A1 + st.nb * SW_PANEL;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                            const float4 d = s_d[row];
                            const float p = acc[r] + bcol;                       // = the per-point part as prcnn_packed_layer stores it
                            Y[row * SW_LD + 32 * w + j] = fmaxf(fmaf(wzc, d.z, fmaf(wyc, d.y, fmaf(wxc, d.x, p))), 0.f);
                        }
                    } else if (st.layer == 1) {
                        const float bcol = sb2[col];
                        float *Y = X + st.nb * SW_PANEL;
#pragma unroll
                        for (int r = 0; r < 16; ++r) Y[((r & 3) + 8 * (r >> 2) + 4 * h) * SW_LD + 32 * w + j] = fmaxf(acc[r] + bcol, 0.f);
                    } else {
                        const int myc = ctr[j], prevc = ctr[j ? j - 1 : 0];
                        const unsigned int start = (unsigned int)__ballot(lane < 32 && (lane == 0 || myc != prevc));
                        sw_segmented_max(acc, ctr, start, h, a.out, a.out_stride, a.out_col + col, sb3[col]);
                    }
                }
            }
        }
        const long un = __builtin_amdgcn_readfirstlane((int)slot[(served + 1) & 1]);
        lds_barrier();                                                     // X / A1 / ctr / s_d are free for the next unit
        u = un;
    }
    if (tid == 0) ticket_release(a.ticket);          // the launch's last workgroup zeroes the counter for the word's next user
}

}  // namespace prcnn

using namespace prcnn;

namespace prcnn {
unsigned int *next_ticket(hipStream_t st);    // sa_mlp_fused.hip
}

static size_t sa_wide3_lds(int c0, int c1, int c2, int c3)
{
    const int kp0 = c0 / 128, kp1 = c1 / 128, kp2 = c2 / 128;
    return (size_t)((kp0 > kp2 ? kp0 : kp2) + kp1) * SW_PANEL * sizeof(float) + SW_R * sizeof(float4) + SW_R * sizeof(int) +
           2 * sizeof(unsigned int) + (size_t)(c1 + c2 + c3 + 3 * c1) * sizeof(float);
}

extern "C" int prcnn_sa_wide_fused3_supported(int c0, int c1, int c2, int c3)
{
    if (c0 <= 0 || c1 <= 0 || c2 <= 0 || c3 <= 0 || c0 % 128 || c1 % 128 || c2 % 128 || c3 % 128) return 0;
    const int kp0 = c0 / 128, kp1 = c1 / 128, kp2 = c2 / 128, nb3 = c3 / 128;
    if ((kp0 * kp1 + kp1 * kp2 + kp2 * nb3) % 2) return 0;                  // the two weight register sets alternate
    return sa_wide3_lds(c0, c1, c2, c3) <= 96 * 1024 ? 1 : 0;
}

// F (b,n,c0) point features; wcat = w1 (c0,c1) | w2 (c1,c2) | w3 (c2,c3), k-major, in one allocation; wxyz (3,c1); row list of
// prcnn_ball_pack; widths multiples of 128 (prcnn_sa_wide_fused3_supported); out[(b*m)][out_col .. out_col + c3) with row stride
// out_stride, zeroed here unless out_is_zero.
extern "C" int prcnn_sa_wide_fused3(int b, int n, int m, int c0, int c1, int c2, int c3, long max_tiles, const float *F, const float *wxyz,
                                    const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud, const unsigned int *hdr,
                                    const float *wcat, const float *b1, const float *b2, const float *b3, float *out, int out_stride,
                                    int out_col, int out_is_zero, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && max_tiles >= 0, "sa_wide_fused3: bad sizes");
    PRCNN_REQUIRE(prcnn_sa_wide_fused3_supported(c0, c1, c2, c3), "sa_wide_fused3: unsupported widths %d-%d-%d-%d", c0, c1, c2, c3);
    PRCNN_REQUIRE(n <= 65536 && m <= 65536, "sa_wide_fused3: cloud too large for the 16-bit row descriptors");
    PRCNN_REQUIRE(out_stride >= out_col + c3 && out_col >= 0, "sa_wide_fused3: bad output slice");
    if ((long)b * m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(F && wxyz && rowinfo && rowdxyz && hdr && wcat && b1 && b2 && b3 && out, "sa_wide_fused3: null pointer");
    // tilecloud == NULL: the rows carry their cloud (descriptor (cloud << 16) | (centre << 9) | point; prcnn_rcnn_roi_geometry_packs)
    PRCNN_REQUIRE(tilecloud || (n <= 512 && m <= 128 && b <= 65536), "sa_wide_fused3: a list without tilecloud holds clouds of <= 512 points, <= 128 centres");
    PRCNN_REQUIRE((((uintptr_t)F | (uintptr_t)wcat | (uintptr_t)rowdxyz) & 15) == 0, "sa_wide_fused3: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (!out_is_zero && hipMemset2DAsync(out + out_col, (size_t)out_stride * sizeof(float), 0, (size_t)c3 * sizeof(float), (size_t)b * m, st) != hipSuccess) {
        set_error("sa_wide_fused3: cannot zero the output slice");
        return PRCNN_ELAUNCH;
    }
    if (max_tiles == 0) return PRCNN_OK;
    const size_t lds = sa_wide3_lds(c0, c1, c2, c3);
    const int rc = ensure_dynamic_lds((const void *)sa_wide3_kernel, lds, "sa_wide_fused3");
    if (rc != PRCNN_OK) return rc;
    SaWide3Args a;
    a.n = n; a.m = m; a.c0 = c0; a.c1 = c1; a.c2 = c2; a.c3 = c3; a.hdr = hdr; a.rowdxyz = (const float4 *)rowdxyz; a.F = F; a.wxyz = wxyz;
    a.rowinfo = rowinfo; a.tilecloud = tilecloud; a.wcat = wcat; a.b1 = b1; a.b2 = b2; a.b3 = b3; a.out = out; a.out_stride = out_stride;
    a.out_col = out_col;
    a.ticket = next_ticket(st);
    if (!a.ticket) { set_error("sa_wide_fused3: cannot set up the unit ticket"); return PRCNN_ELAUNCH; }
    const long units = 2 * max_tiles;
    const long grid = units < mfma_grid_cap() ? units : mfma_grid_cap();
    hipLaunchKernelGGL(sa_wide3_kernel, dim3((unsigned)grid), dim3(256), lds, st, a);
    return check_launch("sa_wide_fused3");
}
