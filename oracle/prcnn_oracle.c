/*
 * prcnn_oracle.c -- CPU ORACLE (test infrastructure, NOT a product path).
 * See prcnn_oracle.h for the contract and the reference file:line each function follows.
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off [-fopenmp]).
 */
#include "prcnn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ helpers */

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* f32 trig contract: correctly rounded from the f64 libm value */
static inline float cos_f32(float x) { return (float)cos((double)x); }
static inline float sin_f32(float x) { return (float)sin((double)x); }
static inline float atan2_f32(float y, float x) { return (float)atan2((double)y, (double)x); }

/* (ax-bx)^2 + (ay-by)^2 + (az-bz)^2, left to right, no contraction.
 * ball_query_gpu.cu:33, sampling_gpu.cu:133, interpolate_gpu.cu:37 all share this shape
 * (the operand order inside each difference differs, the square does not). */
static inline float sqdist3(const float *p, const float *q)
{
    float dx = p[0] - q[0];
    float dy = p[1] - q[1];
    float dz = p[2] - q[2];
#if defined(ORC_VARIANT) && ORC_VARIANT == 2
    /* contraction form B (the other association a compiler may pick): fma(dz,dz, fma(dy,dy, dx*dx)) */
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
#else
    /* ORC_VARIANT 0: -ffp-contract=off, three products, two sums, five roundings (THE contract).
     * ORC_VARIANT 1: this same expression under -ffp-contract=fast -mfma: gcc (like LLVM/NVVM) fuses the LEFT
     * product of an add of two products first -> fma(dz,dz, fma(dx,dx, dy*dy)): contraction form A. */
    return dx * dx + dy * dy + dz * dz;
#endif
}

/* which arithmetic variant this build of the oracle is: 0 contract-off (the parity contract), 1 / 2 FMA-contracted
 * second oracles (bound on "oracle vs the reference's nvcc -O2 binary", pointnet2/setup.py:19-20) */
int orc_variant(void)
{
#ifdef ORC_VARIANT
    return ORC_VARIANT;
#else
    return 0;
#endif
}

/* cuda_utils.h:10-13 */
int orc_opt_n_threads(int work_size)
{
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

/* ------------------------------------------------------------- K1 ball query */

void orc_ball_query(int b, int n, int m, float radius, int nsample,
                    const float *new_xyz, const float *xyz, int *idx)
{
    const float r2 = radius * radius;
    const long total = (long)b * m;
#pragma omp parallel for schedule(dynamic, 64)
    for (long q = 0; q < total; ++q) {
        const int bi = (int)(q / m);
        const float *centre = new_xyz + q * 3;
        const float *cloud = xyz + (long)bi * n * 3;
        int *row = idx + q * nsample;
        int found = 0;
        for (int k = 0; k < n && found < nsample; ++k) {
            if (sqdist3(centre, cloud + 3 * k) < r2) {
                if (found == 0)
                    for (int l = 0; l < nsample; ++l) row[l] = k; /* first hit back-fills */
                row[found++] = k;
            }
        }
        /* found == 0: row untouched (caller zero-fills, pointnet2_utils.py:218) */
    }
}

/* ------------------------------------------------------ K2/K3 group (+grad) */

void orc_group_points(int b, int c, int n, int npoints, int nsample,
                      const float *points, const int *idx, float *out)
{
    const long slots = (long)npoints * nsample;
#pragma omp parallel for collapse(2)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((long)bi * c + ci) * n;
            const int *ix = idx + (long)bi * slots;
            float *dst = out + ((long)bi * c + ci) * slots;
            for (long s = 0; s < slots; ++s) dst[s] = src[ix[s]];
        }
}

void orc_group_points_grad(int b, int c, int n, int npoints, int nsample,
                           const float *grad_out, const int *idx, float *grad_points)
{
    const long slots = (long)npoints * nsample;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *dst = grad_points + ((long)bi * c + ci) * n;
            const int *ix = idx + (long)bi * slots;
            const float *g = grad_out + ((long)bi * c + ci) * slots;
            for (long s = 0; s < slots; ++s) dst[ix[s]] += g[s];
        }
}

/* ----------------------------------------------------- K4/K5 gather (+grad) */

void orc_gather_points(int b, int c, int n, int npoints,
                       const float *points, const int *idx, float *out)
{
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((long)bi * c + ci) * n;
            const int *ix = idx + (long)bi * npoints;
            float *dst = out + ((long)bi * c + ci) * npoints;
            for (int p = 0; p < npoints; ++p) dst[p] = src[ix[p]];
        }
}

void orc_gather_points_grad(int b, int c, int n, int npoints,
                            const float *grad_out, const int *idx, float *grad_points)
{
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *dst = grad_points + ((long)bi * c + ci) * n;
            const int *ix = idx + (long)bi * npoints;
            const float *g = grad_out + ((long)bi * c + ci) * npoints;
            for (int p = 0; p < npoints; ++p) dst[ix[p]] += g[p];
        }
}

/* ------------------------------------------------------------------- K6 FPS */

/* One reference thread block of `bs` threads is emulated literally:
 *   - thread t scans k = t, t+bs, ... keeping the first strict maximum of
 *     min(d, temp[k])                                   (sampling_gpu.cu:124-138)
 *   - shared-memory tree: for s = bs/2 .. 1, slot t<s takes slot t+s only when it is
 *     strictly larger                                   (sampling_gpu.cu:86-91,139-203)
 * so ties resolve exactly as on the reference GPU for that block size. */
void orc_furthest_point_sampling_bs(int b, int n, int m, int bs,
                                    const float *xyz, float *temp, int *idx)
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
            const float *pivot = cloud + 3 * old;
            for (int t = 0; t < bs; ++t) {
                float bv = -1.0f;
                int bk = 0;
                for (int k = t; k < n; k += bs) {
                    float d = sqdist3(cloud + 3 * k, pivot);
                    float d2 = d < mind[k] ? d : mind[k];
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

void orc_furthest_point_sampling(int b, int n, int m,
                                 const float *xyz, float *temp, int *idx)
{
    orc_furthest_point_sampling_bs(b, n, m, orc_opt_n_threads(n), xyz, temp, idx);
}

/* ------------------------------------------------- K7/K8/K9 three_nn, interp */

void orc_three_nn(int b, int n, int m, const float *unknown, const float *known,
                  float *dist2, int *idx)
{
    const long total = (long)b * n;
#pragma omp parallel for schedule(static)
    for (long q = 0; q < total; ++q) {
        const int bi = (int)(q / n);
        const float *u = unknown + q * 3;
        const float *kn = known + (long)bi * m * 3;
        double b1 = 1e40, b2 = 1e40, b3 = 1e40; /* double bests, f32 candidates */
        int i1 = 0, i2 = 0, i3 = 0;
        for (int k = 0; k < m; ++k) {
            float d = sqdist3(u, kn + 3 * k);
            if (d < b1) {
                b3 = b2; i3 = i2;
                b2 = b1; i2 = i1;
                b1 = d; i1 = k;
            } else if (d < b2) {
                b3 = b2; i3 = i2;
                b2 = d; i2 = k;
            } else if (d < b3) {
                b3 = d; i3 = k;
            }
        }
        dist2[q * 3 + 0] = (float)b1; dist2[q * 3 + 1] = (float)b2; dist2[q * 3 + 2] = (float)b3;
        idx[q * 3 + 0] = i1; idx[q * 3 + 1] = i2; idx[q * 3 + 2] = i3;
    }
}

void orc_three_interpolate(int b, int c, int m, int n, const float *points,
                           const int *idx, const float *weight, float *out)
{
#pragma omp parallel for collapse(2)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((long)bi * c + ci) * m;
            const int *ix = idx + (long)bi * n * 3;
            const float *w = weight + (long)bi * n * 3;
            float *dst = out + ((long)bi * c + ci) * n;
            for (int p = 0; p < n; ++p)
                dst[p] = w[3 * p] * src[ix[3 * p]] + w[3 * p + 1] * src[ix[3 * p + 1]]
                       + w[3 * p + 2] * src[ix[3 * p + 2]];
        }
}

void orc_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                const int *idx, const float *weight, float *grad_points)
{
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *dst = grad_points + ((long)bi * c + ci) * m;
            const int *ix = idx + (long)bi * n * 3;
            const float *w = weight + (long)bi * n * 3;
            const float *g = grad_out + ((long)bi * c + ci) * n;
            for (int p = 0; p < n; ++p)
                for (int j = 0; j < 3; ++j) dst[ix[3 * p + j]] += g[p] * w[3 * p + j];
        }
}

/* ------------------------------------------------ QueryAndGroup (fused view) */

void orc_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                         const float *new_xyz, const float *xyz, const float *features,
                         int *idx, float *out)
{
    const long slots = (long)m * nsample;
    memset(idx, 0, sizeof(int) * (size_t)b * (size_t)slots);
    orc_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx);
    const int cout = 3 + c;
#pragma omp parallel for
    for (int bi = 0; bi < b; ++bi) {
        const int *ix = idx + (long)bi * slots;
        for (int ax = 0; ax < 3; ++ax) { /* grouped_xyz -= new_xyz (pointnet2_utils.py:251-252) */
            float *dst = out + ((long)bi * cout + ax) * slots;
            for (int p = 0; p < m; ++p) {
                const float ctr = new_xyz[((long)bi * m + p) * 3 + ax];
                for (int s = 0; s < nsample; ++s)
                    dst[(long)p * nsample + s] =
                        xyz[((long)bi * n + ix[(long)p * nsample + s]) * 3 + ax] - ctr;
            }
        }
        for (int ci = 0; ci < c; ++ci) {
            const float *src = features + ((long)bi * c + ci) * n;
            float *dst = out + ((long)bi * cout + 3 + ci) * slots;
            for (long s = 0; s < slots; ++s) dst[s] = src[ix[s]];
        }
    }
}

/* --------------------------------------------------- K10-K13 BEV IoU and NMS */

typedef struct { float x, y; } pt2;

static const float IOU_EPS = 1e-8f; /* iou3d_kernel.cu:13 */

/* iou3d_kernel.cu:38-40 */
static inline float cross3(pt2 p1, pt2 p2, pt2 p0)
{
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

static inline float fmin2(float a, float b) { return a < b ? a : b; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }

/* iou3d_kernel.cu:73-106; (p0,p1) edge of a, (q0,q1) edge of b */
static int seg_intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2 *ans)
{
    /* bounding-rectangle rejection :42-48 */
    if (!(fmin2(p0.x, p1.x) <= fmax2(q0.x, q1.x) && fmin2(q0.x, q1.x) <= fmax2(p0.x, p1.x) &&
          fmin2(p0.y, p1.y) <= fmax2(q0.y, q1.y) && fmin2(q0.y, q1.y) <= fmax2(p0.y, p1.y)))
        return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* iou3d_kernel.cu:50-65; cos(-t)=cos t and sin(-t)=-sin t exactly for the libm pair used */
static int corner_in_box(const float *box, float cosv, float sinv, pt2 p)
{
    const float MARGIN = 1e-5f;
    float cx = (box[0] + box[2]) / 2;
    float cy = (box[1] + box[3]) / 2;
    float ac = cosv, as = -sinv; /* angle_cos = cos(-ry), angle_sin = sin(-ry) */
    float rx = (p.x - cx) * ac + (p.y - cy) * as + cx;
    float ry = -(p.x - cx) * as + (p.y - cy) * ac + cy;
    return rx > box[0] - MARGIN && rx < box[2] + MARGIN && ry > box[1] - MARGIN && ry < box[3] + MARGIN;
}

/* iou3d_kernel.cu:108-212 */
static float rbox_overlap(const float *A, const float *B)
{
    pt2 ca = { (A[0] + A[2]) / 2, (A[1] + A[3]) / 2 };
    pt2 cb = { (B[0] + B[2]) / 2, (B[1] + B[3]) / 2 };
    pt2 pa[5] = { {A[0], A[1]}, {A[2], A[1]}, {A[2], A[3]}, {A[0], A[3]} };
    pt2 pb[5] = { {B[0], B[1]}, {B[2], B[1]}, {B[2], B[3]}, {B[0], B[3]} };
    const float acos_ = cos_f32(A[4]), asin_ = sin_f32(A[4]);
    const float bcos_ = cos_f32(B[4]), bsin_ = sin_f32(B[4]);
    for (int k = 0; k < 4; ++k) { /* rotate_around_center :92-96 */
        float nx = (pa[k].x - ca.x) * acos_ + (pa[k].y - ca.y) * asin_ + ca.x;
        float ny = -(pa[k].x - ca.x) * asin_ + (pa[k].y - ca.y) * acos_ + ca.y;
        pa[k].x = nx; pa[k].y = ny;
        nx = (pb[k].x - cb.x) * bcos_ + (pb[k].y - cb.y) * bsin_ + cb.x;
        ny = -(pb[k].x - cb.x) * bsin_ + (pb[k].y - cb.y) * bcos_ + cb.y;
        pb[k].x = nx; pb[k].y = ny;
    }
    pa[4] = pa[0];
    pb[4] = pb[0];

    pt2 poly[24]; /* reference declares 16; 24 = 16 edge hits + 8 corners, never UB here */
    pt2 centre = { 0.f, 0.f };
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersection(pa[i + 1], pa[i], pb[j + 1], pb[j], &poly[cnt])) {
                centre.x = centre.x + poly[cnt].x;
                centre.y = centre.y + poly[cnt].y;
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (corner_in_box(A, acos_, asin_, pb[k])) {
            centre.x = centre.x + pb[k].x; centre.y = centre.y + pb[k].y;
            poly[cnt++] = pb[k];
        }
        if (corner_in_box(B, bcos_, bsin_, pa[k])) {
            centre.x = centre.x + pa[k].x; centre.y = centre.y + pa[k].y;
            poly[cnt++] = pa[k];
        }
    }
    centre.x /= cnt;
    centre.y /= cnt;

    /* bubble sort by angle about the centroid (:183-193); point_cmp is a pure function of
     * the point, so the angle is evaluated once per vertex and swapped with it */
    float ang[24];
    for (int i = 0; i < cnt; ++i) ang[i] = atan2_f32(poly[i].y - centre.y, poly[i].x - centre.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                pt2 tp = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = tp;
                float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }

    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        pt2 u = { poly[k].x - poly[0].x, poly[k].y - poly[0].y };
        pt2 v = { poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y };
        area += u.x * v.y - u.y * v.x;
    }
    return (float)(fabs((double)area) / 2.0);
}

/* iou3d_kernel.cu:214-221 */
static float rbox_iou(const float *A, const float *B)
{
    float sa = (A[2] - A[0]) * (A[3] - A[1]);
    float sb = (B[2] - B[0]) * (B[3] - B[1]);
    float so = rbox_overlap(A, B);
    return so / fmax2(sa + sb - so, IOU_EPS);
}

/* iou3d_kernel.cu:295-303 */
static float aabox_iou(const float *a, const float *b)
{
    float left = fmax2(a[0], b[0]), right = fmin2(a[2], b[2]);
    float top = fmax2(a[1], b[1]), bottom = fmin2(a[3], b[3]);
    float width = fmax2(right - left, 0.f), height = fmax2(bottom - top, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return interS / fmax2(Sa + Sb - interS, IOU_EPS);
}

void orc_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                           float *ans_overlap)
{
#pragma omp parallel for
    for (int i = 0; i < num_a; ++i)
        for (int j = 0; j < num_b; ++j)
            ans_overlap[(long)i * num_b + j] = rbox_overlap(boxes_a + 5 * i, boxes_b + 5 * j);
}

void orc_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                       float *ans_iou)
{
#pragma omp parallel for
    for (int i = 0; i < num_a; ++i)
        for (int j = 0; j < num_b; ++j)
            ans_iou[(long)i * num_b + j] = rbox_iou(boxes_a + 5 * i, boxes_b + 5 * j);
}

/* mask kernel (64x64 tiles) followed by the serial OR-reduce of iou3d.cpp:100-119.
 * Only the words the reduce reads (column block >= row block) are produced. */
static int nms_generic(int n, const float *boxes, long long *keep, float thresh,
                       float (*iou)(const float *, const float *))
{
    if (n <= 0) return 0;
    const int cb = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)calloc((size_t)n * cb, sizeof(unsigned long long));
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; ++i) {
        const int rb = i / 64;
        for (int c = rb; c < cb; ++c) {
            unsigned long long t = 0;
            int j0 = (c == rb) ? (i % 64) + 1 : 0;
            int lim = n - c * 64 < 64 ? n - c * 64 : 64;
            for (int j = j0; j < lim; ++j)
                if (iou(boxes + 5 * i, boxes + 5 * (c * 64 + j)) > thresh) t |= 1ULL << j;
            mask[(long)i * cb + c] = t;
        }
    }
    unsigned long long *remv = (unsigned long long *)calloc((size_t)cb, sizeof(unsigned long long));
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep[kept++] = i;
            const unsigned long long *row = mask + (long)i * cb;
            for (int j = nb; j < cb; ++j) remv[j] |= row[j];
        }
    }
    free(mask);
    free(remv);
    return kept;
}

int orc_nms(int boxes_num, const float *boxes, long long *keep, float thresh)
{
    return nms_generic(boxes_num, boxes, keep, thresh, rbox_iou);
}

int orc_nms_normal(int boxes_num, const float *boxes, long long *keep, float thresh)
{
    return nms_generic(boxes_num, boxes, keep, thresh, aabox_iou);
}

/* ------------------------------------------------------- K14-K16 RoI pooling */

/* roipool3d_kernel.cu:14-28 == roipool3d.cpp:82-95.  The trig pair depends on the box
 * only, so it is hoisted by the callers. */
typedef struct { float cx, cy, cz, h, w, l, cosa, sina; } box3d_t;

static box3d_t box_prepare(const float *bx)
{
    box3d_t r;
    r.cx = bx[0]; r.cz = bx[2]; r.h = bx[3]; r.w = bx[4]; r.l = bx[5];
    r.cy = (float)((double)bx[1] - (double)bx[3] / 2.0); /* cy = bottom_y - h / 2.0 */
    r.cosa = cos_f32(bx[6]);
    r.sina = sin_f32(bx[6]);
    return r;
}

static inline int pt_in_box(const float *p, const box3d_t *bx)
{
    const float max_dis = 10.0f;
    if (fabsf(p[0] - bx->cx) > max_dis || (double)fabsf(p[1] - bx->cy) > (double)bx->h / 2.0 ||
        fabsf(p[2] - bx->cz) > max_dis)
        return 0;
    float xr = (p[0] - bx->cx) * bx->cosa + (p[2] - bx->cz) * (-bx->sina);
    float zr = (p[0] - bx->cx) * bx->sina + (p[2] - bx->cz) * bx->cosa;
    return ((double)xr >= -(double)bx->l / 2.0) & ((double)xr <= (double)bx->l / 2.0) &
           ((double)zr >= -(double)bx->w / 2.0) & ((double)zr <= (double)bx->w / 2.0);
}

void orc_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                   int sampled_pts_num, const float *xyz, const float *boxes3d,
                   const float *pts_feature, float *pooled_features, int *pooled_empty_flag)
{
    const int width = 3 + feature_in_len;
    const long nbox = (long)batch_size * boxes_num;
#pragma omp parallel for schedule(dynamic, 4)
    for (long q = 0; q < nbox; ++q) {
        const int bi = (int)(q / boxes_num);
        const box3d_t bx = box_prepare(boxes3d + q * 7);
        const float *pts = xyz + (long)bi * pts_num * 3;
        const float *feat = pts_feature + (long)bi * pts_num * feature_in_len;
        float *dst = pooled_features + q * sampled_pts_num * width;
        int *sel = (int *)malloc(sizeof(int) * (size_t)sampled_pts_num);
        int cnt = 0;
        /* get_pooled_idx :123-160: first sampled_pts_num hits in index order */
        for (int k = 0; k < pts_num && cnt < sampled_pts_num; ++k)
            if (pt_in_box(pts + 3 * k, &bx)) sel[cnt++] = k;
        if (cnt == 0) {
            pooled_empty_flag[q] = 1; /* rows stay as the caller left them (zeros) */
        } else {
            for (int k = cnt; k < sampled_pts_num; ++k) sel[k] = sel[k % cnt]; /* wrap-around */
            for (int s = 0; s < sampled_pts_num; ++s) { /* roipool3d_forward :163-194 */
                const int k = sel[s];
                float *row = dst + (long)s * width;
                row[0] = pts[3 * k]; row[1] = pts[3 * k + 1]; row[2] = pts[3 * k + 2];
                memcpy(row + 3, feat + (long)k * feature_in_len, sizeof(float) * (size_t)feature_in_len);
            }
        }
        free(sel);
    }
}

void orc_pts_in_boxes3d(int boxes_num, int pts_num, const float *pts, const float *boxes3d,
                        long long *pts_flag)
{
#pragma omp parallel for
    for (int i = 0; i < boxes_num; ++i) {
        const box3d_t bx = box_prepare(boxes3d + 7 * i);
        for (int j = 0; j < pts_num; ++j)
            pts_flag[(long)i * pts_num + j] = pt_in_box(pts + 3 * j, &bx);
    }
}

void orc_roipool3d_cpu(int boxes_num, int pts_num, int feature_len, int sampled_pts_num,
                       const float *pts, const float *boxes3d, const float *pts_feature,
                       float *pooled_pts, float *pooled_features, long long *pooled_empty_flag)
{
    for (int i = 0; i < boxes_num; ++i) pooled_empty_flag[i] = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < boxes_num; ++i) {
        const box3d_t bx = box_prepare(boxes3d + 7 * i);
        float *op = pooled_pts + (long)i * sampled_pts_num * 3;
        float *of = pooled_features + (long)i * sampled_pts_num * feature_len;
        int cnt = 0;
        for (int j = 0; j < pts_num && cnt < sampled_pts_num; ++j) {
            if (!pt_in_box(pts + 3 * j, &bx)) continue;
            memcpy(op + 3 * cnt, pts + 3 * j, 3 * sizeof(float));
            memcpy(of + (long)cnt * feature_len, pts_feature + (long)j * feature_len,
                   sizeof(float) * (size_t)feature_len);
            ++cnt;
        }
        if (cnt == 0) {
            pooled_empty_flag[i] = 1;
        } else {
            for (int j = cnt; j < sampled_pts_num; ++j) {
                memcpy(op + 3 * j, op + 3 * (j % cnt), 3 * sizeof(float));
                memcpy(of + (long)j * feature_len, of + (long)(j % cnt) * feature_len,
                       sizeof(float) * (size_t)feature_len);
            }
        }
    }
}

/* -------------------------------------------------- K18 rotated IoU (eval) */

/* numba type rules followed here: f32 (op) f32 -> f32; f32 / int-literal -> f64;
 * values stored into float32 local arrays are rounded to f32 at the store. */

/* rotate_iou.py:203-228 */
static void rbbox_corners(float *corners, const float *rb)
{
    float a_cos = cos_f32(rb[4]);
    float a_sin = sin_f32(rb[4]);
    float cx = rb[0], cy = rb[1], xd = rb[2], yd = rb[3];
    float px[4], py[4];
    px[0] = (float)(-(double)xd / 2); px[1] = px[0];
    px[2] = (float)((double)xd / 2);  px[3] = px[2];
    py[0] = (float)(-(double)yd / 2); py[3] = py[0];
    py[1] = (float)((double)yd / 2);  py[2] = py[1];
    for (int i = 0; i < 4; ++i) {
        corners[2 * i] = a_cos * px[i] + a_sin * py[i] + cx;
        corners[2 * i + 1] = -a_sin * px[i] + a_cos * py[i] + cy;
    }
}

/* rotate_iou.py:159-177 */
static int pt_in_quad(float x, float y, const float *c)
{
    float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    float ap0 = x - c[0], ap1 = y - c[1];
    float abab = ab0 * ab0 + ab1 * ab1;
    float abap = ab0 * ap0 + ab1 * ap1;
    float adad = ad0 * ad0 + ad1 * ad1;
    float adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

/* rotate_iou.py:72-114 */
static int seg_isect_eval(const float *p1, const float *p2, int i, int j, float *out)
{
    float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    float B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
    float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    float D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
    float BA0 = B0 - A0, BA1 = B1 - A1;
    float DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    int acd = DA1 * CA0 > CA1 * DA0;
    int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd == bcd) return 0;
    int abc = CA1 * BA0 > BA1 * CA0;
    int abd = DA1 * BA0 > BA1 * DA0;
    if (abc == abd) return 0;
    float DC0 = D0 - C0, DC1 = D1 - C1;
    float ABBA = A0 * B1 - B0 * A1;
    float CDDC = C0 * D1 - D0 * C1;
    float DH = BA1 * DC0 - BA0 * DC1;
    float Dx = ABBA * DC0 - BA0 * CDDC;
    float Dy = ABBA * DC1 - BA1 * CDDC;
    out[0] = Dx / DH;
    out[1] = Dy / DH;
    return 1;
}

/* rotate_iou.py:231-244 with :180-200, :33-69, :16-29 */
static double rinter_eval(const float *r1, const float *r2)
{
    float c1[8], c2[8], ip[16 * 2 + 16]; /* reference buffer holds 8 points; see note */
    rbbox_corners(c1, r1);
    rbbox_corners(c2, r2);
    int cnt = 0;
    for (int i = 0; i < 4; ++i) {
        if (pt_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { ip[2 * cnt] = c1[2 * i]; ip[2 * cnt + 1] = c1[2 * i + 1]; ++cnt; }
        if (pt_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { ip[2 * cnt] = c2[2 * i]; ip[2 * cnt + 1] = c2[2 * i + 1]; ++cnt; }
    }
    float tp[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_isect_eval(c1, c2, i, j, tp)) { ip[2 * cnt] = tp[0]; ip[2 * cnt + 1] = tp[1]; ++cnt; }

    if (cnt > 0) { /* sort_vertex_in_convex_polygon */
        float ctr0 = 0.f, ctr1 = 0.f;
        for (int i = 0; i < cnt; ++i) { ctr0 += ip[2 * i]; ctr1 += ip[2 * i + 1]; }
        ctr0 = (float)((double)ctr0 / cnt);
        ctr1 = (float)((double)ctr1 / cnt);
        float vs[24];
        for (int i = 0; i < cnt; ++i) {
            float v0 = ip[2 * i] - ctr0, v1 = ip[2 * i + 1] - ctr1;
            float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d; v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < cnt; ++i) {
            if (vs[i - 1] > vs[i]) {
                float t = vs[i], tx = ip[2 * i], ty = ip[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > t) {
                    vs[j] = vs[j - 1];
                    ip[2 * j] = ip[2 * j - 2];
                    ip[2 * j + 1] = ip[2 * j - 1];
                    --j;
                }
                vs[j] = t; ip[2 * j] = tx; ip[2 * j + 1] = ty;
            }
        }
    }
    double area = 0.0; /* area_val is a python float (f64); each triangle term is f32 math / 2.0 */
    for (int i = 0; i < cnt - 2; ++i) {
        const float *a = ip, *b = ip + 2 * i + 2, *c = ip + 2 * i + 4;
        float num = (a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0]);
        area += fabs((double)num / 2.0);
    }
    return area;
}

void orc_rotate_iou_eval(int n, int k, const float *boxes, const float *query_boxes,
                         float *iou, int criterion)
{
    /* kernel :287-291 calls devRotateIoUEval(query_box, box): rbox1 = query box */
#pragma omp parallel for
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) {
            const float *r1 = query_boxes + 5 * j, *r2 = boxes + 5 * i;
            float area1 = r1[2] * r1[3], area2 = r2[2] * r2[3];
            double ai = rinter_eval(r1, r2);
            double v;
            if (criterion == -1) v = ai / ((double)(area1 + area2) - ai);
            else if (criterion == 0) v = ai / (double)area1;
            else if (criterion == 1) v = ai / (double)area2;
            else v = ai;
            iou[(long)i * k + j] = (float)v;
        }
}
