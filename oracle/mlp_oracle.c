/*
 * mlp_oracle.c -- CPU ORACLE (test infrastructure, NOT a product path) for the shared-MLP kernels this build owns.
 *
 * The reference runs its shared MLPs as cuDNN 1x1 convolutions (pointnet2_lib/pointnet2/pytorch_utils.py:5-101,
 * pointnet2_modules.py:37-53) whose summation order is unspecified; parity with it is a tolerance (1e-4 on boxes).
 * The kernels of csrc/sa_mlp_fused.hip, sa_packed.hip, rcnn_point_mlp.hip and sa_xyz_mlp.hip FIX an order, and this
 * file restates exactly that order in scalar C so that they can be checked BIT FOR BIT:
 *
 *   - v_mfma_f32_32x32x2_f32 is bitwise a chain of fused multiply-adds over k (MI355X_MICROARCH.md, matrix cores);
 *     the kernels feed step s of a 128-deep panel with k = s (lanes 0-31) and k = s + 64 (lanes 32-63), and the
 *     instruction accumulates its two k values in that order.  A 128-deep dot product is therefore
 *         acc = 0;  for s in 0..63:  acc = fma(a[s], w[s], acc);  acc = fma(a[s+64], w[s+64], acc)
 *     followed by  + bias  and the ReLU, each rounded once.  (ORC_MFMA_KORDER 1 swaps the pair; tests pin which.)
 *   - the tile builders and sa_xyz_mlp_kernel are explicit fmaf chains that start from the bias.
 *
 * Built with -ffp-contract=off: every fused operation here is an explicit fmaf().
 */
#include "prcnn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int g_korder = 0;
void orc_set_mfma_korder(int k) { g_korder = k ? 1 : 0; }

static inline float relu(float v) { return v > 0.f ? v : 0.f; }

/* acc[0..n) += over one 128-deep panel: a[128] (activations of one row), w = W[k][col] with row stride ldw */
static void panel128(const float *a, const float *w, int ldw, int n, float *acc)
{
    for (int s = 0; s < 64; ++s) {
        const int k0 = g_korder ? s + 64 : s, k1 = g_korder ? s : s + 64;
        const float a0 = a[k0], a1 = a[k1];
        const float *w0 = w + (long)k0 * ldw, *w1 = w + (long)k1 * ldw;
        for (int c = 0; c < n; ++c) acc[c] = fmaf(a0, w0[c], acc[c]);
        for (int c = 0; c < n; ++c) acc[c] = fmaf(a1, w1[c], acc[c]);
    }
}

/* the same for a block of up to 8 rows at once (each (row, column) chain keeps its k order; a weight row is reused by
 * the whole block, which is what makes this fast enough for full-size checks): a[r] = activations of row r, lda apart */
#define RB 8
static void panel128_block(const float *a, long lda, int nrows, const float *w, int ldw, int n, float (*acc)[256])
{
    for (int s = 0; s < 64; ++s) {
        for (int half = 0; half < 2; ++half) {
            const int k = (half ^ g_korder) ? s + 64 : s;
            const float *wk = w + (long)k * ldw;
            for (int r = 0; r < nrows; ++r) {
                const float ar = a[r * lda + k];
                float *ac = acc[r];
                for (int c = 0; c < n; ++c) ac[c] = fmaf(ar, wk[c], ac[c]);
            }
        }
    }
}

/* out[r][0..n) = act(A[r][0..K) @ W + bias) for K a multiple of 128 (panels in order), the MFMA chain order */
void orc_rows_layer_mfma(long rows, int K, int n, const float *A, long lda, const float *W, const float *bias, int do_relu,
                         float *out, long ldo)
{
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        float acc[512];
        for (int c = 0; c < n; ++c) acc[c] = 0.f;
        for (int p = 0; p < K / 128; ++p) panel128(A + r * lda + 128 * p, W + (long)128 * p * n, n, n, acc);
        for (int c = 0; c < n; ++c) {
            const float v = acc[c] + bias[c];
            out[r * ldo + c] = do_relu ? relu(v) : v;
        }
    }
}

/* out[r][c0..c0+n) = act(A[r][0..128) @ W[:, c0..c0+n) + bias[c0..]) in the order of v_mfma_f32_16x16x4_f32 as csrc/rpn_tail.hip's
 * narrow last stage feeds it (round 5; the instruction is bitwise an fma chain over its four k values, profiles/mfma16_probe.hip):
 * step s = 0..31 accumulates k = s, 32 + s, 64 + s, 96 + s in that order.  W (128, ldw) k-major. */
void orc_rows_layer_mfma16(long rows, int c0, int n, const float *A, long lda, const float *W, int ldw, const float *bias, int do_relu,
                           float *out, long ldo)
{
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const float *a = A + r * lda;
        for (int c = c0; c < c0 + n; ++c) {
            float acc = 0.f;
            for (int s = 0; s < 32; ++s)
                for (int kk = 0; kk < 4; ++kk) acc = fmaf(a[32 * kk + s], W[(long)(32 * kk + s) * ldw + c], acc);
            const float v = acc + bias[c];
            out[r * ldo + c] = do_relu ? relu(v) : v;
        }
    }
}

/* csrc/sa_mlp_fused.hip and csrc/sa_packed.hip: one set-abstraction scale, all nsample rows of every group (duplicates
 * included -- this is the reference's semantics; the packed kernel must give the same bits without them). */
void orc_sa_mlp_fused(int b, int n, int m, int ns, int c3, const float *new_xyz, const float *xyz, const float *P,
                      const float *wxyz, const int *idx, const float *w2t, const float *b2, const float *w3t,
                      const float *b3, float *out, int out_stride, int out_col)
{
    const long groups = (long)b * m;
#pragma omp parallel for schedule(dynamic, 16)
    for (long g = 0; g < groups; ++g) {
        const long bi = g / m;
        const float *ct = new_xyz + g * 3;
        float mx[256], a1[RB][128], y1[RB][128], acc[RB][256];
        for (int c = 0; c < c3; ++c) mx[c] = -INFINITY;
        for (int s0 = 0; s0 < ns; s0 += RB) {
            const int nr = ns - s0 < RB ? ns - s0 : RB;
            for (int r = 0; r < nr; ++r) {
                const int k = idx[g * ns + s0 + r];
                const float *pt = xyz + (bi * n + k) * 3;
                const float *base = P + (bi * n + k) * 128;
                const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];
                for (int c = 0; c < 128; ++c)
                    a1[r][c] = relu(fmaf(wxyz[256 + c], dz, fmaf(wxyz[128 + c], dy, fmaf(wxyz[c], dx, base[c]))));
                for (int c = 0; c < 128; ++c) acc[r][c] = 0.f;
            }
            panel128_block(&a1[0][0], 128, nr, w2t, 128, 128, acc);
            for (int r = 0; r < nr; ++r) {
                for (int c = 0; c < 128; ++c) y1[r][c] = relu(acc[r][c] + b2[c]);
                for (int c = 0; c < c3; ++c) acc[r][c] = 0.f;
            }
            panel128_block(&y1[0][0], 128, nr, w3t, c3, c3, acc);
            for (int r = 0; r < nr; ++r)
                for (int c = 0; c < c3; ++c) mx[c] = acc[r][c] > mx[c] ? acc[r][c] : mx[c];
        }
        for (int c = 0; c < c3; ++c) out[g * out_stride + out_col + c] = relu(mx[c] + b3[c]);
    }
}

/* csrc/packed_layer.hip packed_gather_affine_kernel (and the tile builders of the fused kernels): layer 1 of a grouped
 * level as an fmaf chain from the per-point part, for ALL nsample rows of every group: out[(g*ns + s)][0..c1) */
void orc_gather_affine_fma(int b, int n, int m, int ns, int c1, const float *new_xyz, const float *xyz, const float *P,
                           const float *wxyz, const int *idx, float *out)
{
    const long groups = (long)b * m;
#pragma omp parallel for schedule(static)
    for (long g = 0; g < groups; ++g) {
        const long bi = g / m;
        const float *ct = new_xyz + g * 3;
        for (int s = 0; s < ns; ++s) {
            const int k = idx[g * ns + s];
            const float *pt = xyz + (bi * n + k) * 3;
            const float *base = P + (bi * n + k) * c1;
            const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];
            float *o = out + (g * ns + s) * c1;
            for (int c = 0; c < c1; ++c)
                o[c] = relu(fmaf(wxyz[2 * c1 + c], dz, fmaf(wxyz[c1 + c], dy, fmaf(wxyz[c], dx, base[c]))));
        }
    }
}

/* csrc/sa_xyz_mlp.hip: coordinates-only scale; every chain starts from the bias and runs k = 0..K-1 */
void orc_sa_xyz_mlp(int b, int n, int m, int ns, int c1, int c2, int c3, const float *new_xyz, const float *xyz,
                    const int *idx, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                    const float *b3, float *out, int out_stride, int out_col)
{
    const long groups = (long)b * m;
#pragma omp parallel for schedule(dynamic, 64)
    for (long g = 0; g < groups; ++g) {
        const long bi = g / m;
        const float *ct = new_xyz + g * 3;
        float mx[64], a1[64], a2[64], a3[64];
        for (int c = 0; c < c3; ++c) mx[c] = -INFINITY;
        for (int s = 0; s < ns; ++s) {
            const int k = idx[g * ns + s];
            const float *pt = xyz + (bi * n + k) * 3;
            const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];
            for (int j = 0; j < c1; ++j) a1[j] = relu(fmaf(w1[2 * c1 + j], dz, fmaf(w1[c1 + j], dy, fmaf(w1[j], dx, b1[j]))));
            for (int j = 0; j < c2; ++j) a2[j] = b2[j];
            for (int q = 0; q < c1; ++q)
                for (int j = 0; j < c2; ++j) a2[j] = fmaf(w2[q * c2 + j], a1[q], a2[j]);
            for (int j = 0; j < c2; ++j) a2[j] = relu(a2[j]);
            for (int j = 0; j < c3; ++j) a3[j] = b3[j];
            for (int q = 0; q < c2; ++q)
                for (int j = 0; j < c3; ++j) a3[j] = fmaf(w3[q * c3 + j], a2[q], a3[j]);
            for (int j = 0; j < c3; ++j) mx[j] = a3[j] > mx[j] ? a3[j] : mx[j];
        }
        for (int j = 0; j < c3; ++j) out[g * out_stride + out_col + j] = relu(mx[j]);
    }
}

/* csrc/rcnn_point_mlp.hip: rows (r, ld) = [x',y',z',mask,depth,0,0,0 | 128 features at column fcol] ->
 * xfeat = relu(relu(in5 wu1 + bu1) wu2 + bu2), merged = relu([xfeat | feats] wm + bm), p = merged wp + bp */
void orc_rcnn_point_mlp(long r, int ld, int fcol, const float *rows, const float *wu1, const float *bu1, const float *wu2,
                        const float *bu2, const float *wm, const float *bm, const float *wp, const float *bp,
                        float *xfeat, float *merged, float *p)
{
#pragma omp parallel for schedule(static)
    for (long i = 0; i < r; ++i) {
        const float *in = rows + i * ld;
        float u1[128], acc[128], cat[256];
        for (int c = 0; c < 128; ++c)      /* K = 5 layer: the builder's fmaf chain from the bias */
            u1[c] = relu(fmaf(wu1[4 * 128 + c], in[4], fmaf(wu1[3 * 128 + c], in[3], fmaf(wu1[2 * 128 + c], in[2],
                         fmaf(wu1[128 + c], in[1], fmaf(wu1[c], in[0], bu1[c]))))));
        for (int c = 0; c < 128; ++c) acc[c] = 0.f;
        panel128(u1, wu2, 128, 128, acc);
        for (int c = 0; c < 128; ++c) cat[c] = xfeat[i * 128 + c] = relu(acc[c] + bu2[c]);
        for (int c = 0; c < 128; ++c) cat[128 + c] = in[fcol + c];
        for (int c = 0; c < 128; ++c) acc[c] = 0.f;
        panel128(cat, wm, 128, 128, acc);
        panel128(cat + 128, wm + 128 * 128, 128, 128, acc);
        float mg[128];
        for (int c = 0; c < 128; ++c) mg[c] = merged[i * 128 + c] = relu(acc[c] + bm[c]);
        for (int c = 0; c < 128; ++c) acc[c] = 0.f;
        panel128(mg, wp, 128, 128, acc);
        for (int c = 0; c < 128; ++c) p[i * 128 + c] = acc[c] + bp[c];
    }
}

/* csrc/packed_layer.hip rows_dot_kernel: n <= 4 outputs per row; lane l of 32 accumulates k = l, l + 32, ... (fma chain from
 * 0), the partial sums are added in an xor butterfly 16, 8, 4, 2, 1 (lane 0's view), then the bias */
void orc_rows_dot(long rows, int K, int n, const float *A, long lda, const float *W, const float *bias, float *out, long ldo)
{
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r)
        for (int c = 0; c < n; ++c) {
            float p[32];
            for (int l = 0; l < 32; ++l) {
                float acc = 0.f;
                for (int k = l; k < K; k += 32) acc = fmaf(A[r * lda + k], W[(long)k * n + c], acc);
                p[l] = acc;
            }
            for (int d = 16; d >= 1; d >>= 1) {
                float q[32];
                for (int l = 0; l < 32; ++l) q[l] = p[l] + p[l ^ d];
                memcpy(p, q, sizeof(p));
            }
            out[r * ldo + c] = p[0] + bias[c];
        }
}

/* K6 (sampling_gpu.cu:93-209) with the distance ARITHMETIC OF THE REFERENCE'S KERNEL BINARY as hipcc 7.2 builds that file for gfx950
 * (default contraction; read off the disassembly of oracle/_ref/pointnet2_kernels_ref.so, the same in all eleven block-size
 * instantiations):  d = (fma(dy, dy, dx*dx)) + dz*dz  -- v_pk_mul (dx^2, dz^2), v_fma, v_add.  Everything else as
 * orc_furthest_point_sampling_bs (prcnn_oracle.c): per-thread strict '>' scan over k = t, t + bs, ..., tree with strict '>'.
 * The counterpart of prcnn_set_fps_arithmetic(1); lives in this file because it is built with hardware fmaf. */
void orc_furthest_point_sampling_hipcc_bs(int b, int n, int m, int bs, const float *xyz, float *temp, int *idx)
{
    if (m <= 0) return;
#pragma omp parallel for
    for (int bi = 0; bi < b; ++bi) {
        const float *cloud = xyz + (long)bi * n * 3;
        float *mind = temp + (long)bi * n;
        int *sel = idx + (long)bi * m;
        float *best = (float *)malloc(sizeof(float) * (size_t)bs);
        int *besti = (int *)malloc(sizeof(int) * (size_t)bs);
        int old = 0;
        sel[0] = old;
        for (int j = 1; j < m; ++j) {
            const float x1 = cloud[3 * old], y1 = cloud[3 * old + 1], z1 = cloud[3 * old + 2];
            for (int t = 0; t < bs; ++t) {
                float bv = -1.0f;
                int bk = 0;
                for (int k = t; k < n; k += bs) {
                    const float dx = cloud[3 * k] - x1, dy = cloud[3 * k + 1] - y1, dz = cloud[3 * k + 2] - z1;
                    const float dxx = dx * dx, dzz = dz * dz;
                    const float d = fmaf(dy, dy, dxx) + dzz;
                    const float d2 = d < mind[k] ? d : mind[k];
                    mind[k] = d2;
                    if (d2 > bv) { bv = d2; bk = k; }
                }
                best[t] = bv;
                besti[t] = bk;
            }
            for (int s = bs >> 1; s >= 1; s >>= 1)
                for (int t = 0; t < s; ++t)
                    if (best[t + s] > best[t]) { best[t] = best[t + s]; besti[t] = besti[t + s]; }
            old = besti[0];
            sel[j] = old;
        }
        free(best);
        free(besti);
    }
}
