// rotate_iou.hip -- rotated IoU matrix of the KITTI AP evaluator (K18) for gfx950.
//
// Reference behaviour restated: evaluate/rotate_iou.py:16-291 (numba.cuda).  Boxes are in centre
// format [cx, cy, w, h, angle].  numba's typing is followed: f32 (op) f32 stays f32, an f32
// divided by an integer literal is evaluated in f64 and rounded when stored to an f32 array, the
// polygon area accumulates in f64.  One lane per (box, query) pair; the matrix is small
// (tens..hundreds per side per part, eval2.py:352-380) so no tiling is needed.
#include "common.hpp"
#include <math.h>

namespace prcnn {

__device__ __forceinline__ void rbbox_corners(float *c, const float *rb)
{
    const float a_cos = cos_f32(rb[4]), a_sin = sin_f32(rb[4]);
    const float cx = rb[0], cy = rb[1];
    const float hx = (float)((double)rb[2] / 2), hy = (float)((double)rb[3] / 2);
    const float px[4] = { -hx, -hx, hx, hx };
    const float py[4] = { -hy, hy, hy, -hy };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = __fadd_rn(__fadd_rn(__fmul_rn(a_cos, px[i]), __fmul_rn(a_sin, py[i])), cx);
        c[2 * i + 1] = __fadd_rn(__fadd_rn(__fmul_rn(-a_sin, px[i]), __fmul_rn(a_cos, py[i])), cy);
    }
}

__device__ __forceinline__ bool pt_in_quad(float x, float y, const float *c)
{
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    const float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = x - c[0], ap1 = y - c[1];
    const float abab = __fadd_rn(__fmul_rn(ab0, ab0), __fmul_rn(ab1, ab1));
    const float abap = __fadd_rn(__fmul_rn(ab0, ap0), __fmul_rn(ab1, ap1));
    const float adad = __fadd_rn(__fmul_rn(ad0, ad0), __fmul_rn(ad1, ad1));
    const float adap = __fadd_rn(__fmul_rn(ad0, ap0), __fmul_rn(ad1, ap1));
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

__device__ __forceinline__ bool seg_isect(const float *p1, const float *p2, int i, int j, float *out)
{
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    const float B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    const float D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1;
    const float DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = __fmul_rn(DA1, CA0) > __fmul_rn(CA1, DA0);
    const bool bcd = __fmul_rn(D1 - B1, C0 - B0) > __fmul_rn(C1 - B1, D0 - B0);
    if (acd == bcd) return false;
    const bool abc = __fmul_rn(CA1, BA0) > __fmul_rn(BA1, CA0);
    const bool abd = __fmul_rn(DA1, BA0) > __fmul_rn(BA1, DA0);
    if (abc == abd) return false;
    const float DC0 = D0 - C0, DC1 = D1 - C1;
    const float ABBA = __fsub_rn(__fmul_rn(A0, B1), __fmul_rn(B0, A1));
    const float CDDC = __fsub_rn(__fmul_rn(C0, D1), __fmul_rn(D0, C1));
    const float DH = __fsub_rn(__fmul_rn(BA1, DC0), __fmul_rn(BA0, DC1));
    const float Dx = __fsub_rn(__fmul_rn(ABBA, DC0), __fmul_rn(BA0, CDDC));
    const float Dy = __fsub_rn(__fmul_rn(ABBA, DC1), __fmul_rn(BA1, CDDC));
    out[0] = __fdiv_rn(Dx, DH);
    out[1] = __fdiv_rn(Dy, DH);
    return true;
}

__device__ double rinter(const float *r1, const float *r2)
{
    float c1[8], c2[8], ip[48], vs[24];
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
            if (seg_isect(c1, c2, i, j, tp)) { ip[2 * cnt] = tp[0]; ip[2 * cnt + 1] = tp[1]; ++cnt; }
    if (cnt < 3) return 0.0;
    float ctr0 = 0.f, ctr1 = 0.f;
    for (int i = 0; i < cnt; ++i) { ctr0 += ip[2 * i]; ctr1 += ip[2 * i + 1]; }
    ctr0 = (float)((double)ctr0 / cnt);
    ctr1 = (float)((double)ctr1 / cnt);
    for (int i = 0; i < cnt; ++i) {
        float v0 = ip[2 * i] - ctr0, v1 = ip[2 * i + 1] - ctr1;
        const float d = sqrtf(__fadd_rn(__fmul_rn(v0, v0), __fmul_rn(v1, v1)));     // (sqrtf is the correctly rounded one; __fsqrt_rn is the bare v_sqrt_f32)
        v0 = __fdiv_rn(v0, d); v1 = __fdiv_rn(v1, d);
        if (v1 < 0) v0 = -2 - v0;
        vs[i] = v0;
    }
    for (int i = 1; i < cnt; ++i) {
        if (vs[i - 1] > vs[i]) {
            const float tv = vs[i], tx = ip[2 * i], ty = ip[2 * i + 1];
            int j = i;
            while (j > 0 && vs[j - 1] > tv) {
                vs[j] = vs[j - 1];
                ip[2 * j] = ip[2 * j - 2];
                ip[2 * j + 1] = ip[2 * j - 1];
                --j;
            }
            vs[j] = tv; ip[2 * j] = tx; ip[2 * j + 1] = ty;
        }
    }
    double area = 0.0;
    for (int i = 0; i < cnt - 2; ++i) {
        const float *a = ip, *b = ip + 2 * i + 2, *c = ip + 2 * i + 4;
        const float num = __fsub_rn(__fmul_rn(a[0] - c[0], b[1] - c[1]), __fmul_rn(a[1] - c[1], b[0] - c[0]));
        area += fabs((double)num / 2.0);
    }
    return area;
}

// one (box, query) pair: rbox1 = query box, rbox2 = box (kernel :287-291)
__device__ __forceinline__ float pair_value(const float *__restrict__ box, const float *__restrict__ query, int criterion)
{
    float r1[5], r2[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) { r1[q] = query[q]; r2[q] = box[q]; }
    const float area1 = __fmul_rn(r1[2], r1[3]), area2 = __fmul_rn(r2[2], r2[3]);
    const double ai = rinter(r1, r2);
    double v;
    if (criterion == -1) v = ai / ((double)__fadd_rn(area1, area2) - ai);
    else if (criterion == 0) v = ai / (double)area1;
    else if (criterion == 1) v = ai / (double)area2;
    else v = ai;
    return (float)v;
}

__global__ __launch_bounds__(256) void rotate_iou_kernel(int n, int k, const float *__restrict__ boxes,
                                                         const float *__restrict__ qboxes,
                                                         float *__restrict__ iou, int criterion)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n * k) return;
    const int i = (int)(e / k), j = (int)(e - (long)i * k);
    iou[e] = pair_value(boxes + 5 * i, qboxes + 5 * j, criterion);
}

// Block-diagonal form: segment s pairs boxes[box_off[s]..box_off[s+1]) with qboxes[q_off[s]..q_off[s+1]) only
// and writes its row-major (n_s, k_s) block at out_off[s].  The reference evaluates ~50 dense "parts" of ~75
// images each to amortise launches (eval2.py:352-424) and discards the cross-image pairs; here every image
// is a segment and the whole split is one launch that computes only the pairs the evaluator reads.
__global__ __launch_bounds__(256) void rotate_iou_segmented_kernel(int nseg, const long long *__restrict__ out_off,
                                                                   const int *__restrict__ box_off,
                                                                   const int *__restrict__ q_off,
                                                                   const float *__restrict__ boxes,
                                                                   const float *__restrict__ qboxes,
                                                                   float *__restrict__ iou, int criterion)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= out_off[nseg]) return;
    int lo = 0, hi = nseg;                       // last s with out_off[s] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (out_off[mid] <= e) lo = mid; else hi = mid;
    }
    const int k = q_off[lo + 1] - q_off[lo];
    const long long local = e - out_off[lo];
    const int i = (int)(local / k), j = (int)(local - (long long)i * k);
    iou[e] = pair_value(boxes + 5 * (long)(box_off[lo] + i), qboxes + 5 * (long)(q_off[lo] + j), criterion);
}

}  // namespace prcnn

using namespace prcnn;

extern "C" int prcnn_rotate_iou_eval(int n, int k, const float *boxes, const float *query_boxes,
                                     float *iou, int criterion, void *stream)
{
    PRCNN_REQUIRE(n >= 0 && k >= 0, "rotate_iou_eval: bad sizes");
    PRCNN_REQUIRE(criterion >= -1 && criterion <= 2, "rotate_iou_eval: criterion %d not in -1..2", criterion);
    if (n == 0 || k == 0) return PRCNN_OK;
    PRCNN_REQUIRE(boxes && query_boxes && iou, "rotate_iou_eval: null pointer");
    hipLaunchKernelGGL(rotate_iou_kernel, dim3(ceil_div((long)n * k, 256)), dim3(256), 0, (hipStream_t)stream,
                       n, k, boxes, query_boxes, iou, criterion);
    return check_launch("rotate_iou_eval");
}

/* Block-diagonal rotated IoU: nseg segments; box_off / q_off (nseg+1) i32 prefix offsets into boxes / query_boxes,
 * out_off (nseg+1) i64 prefix of n_s*k_s; all three in DEVICE memory; total = out_off[nseg] given by the host. */
extern "C" int prcnn_rotate_iou_eval_segmented(int nseg, long long total, const long long *out_off, const int *box_off,
                                               const int *q_off, const float *boxes, const float *query_boxes,
                                               float *iou, int criterion, void *stream)
{
    PRCNN_REQUIRE(nseg >= 0 && total >= 0, "rotate_iou_eval_segmented: bad sizes");
    PRCNN_REQUIRE(criterion >= -1 && criterion <= 2, "rotate_iou_eval_segmented: criterion %d not in -1..2", criterion);
    if (nseg == 0 || total == 0) return PRCNN_OK;
    PRCNN_REQUIRE(out_off && box_off && q_off && boxes && query_boxes && iou, "rotate_iou_eval_segmented: null pointer");
    PRCNN_REQUIRE(total <= 0x7fffffffLL * 256, "rotate_iou_eval_segmented: %lld pairs exceed the launch grid", total);
    hipLaunchKernelGGL(rotate_iou_segmented_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, nseg, out_off, box_off, q_off, boxes, query_boxes, iou, criterion);
    return check_launch("rotate_iou_eval_segmented");
}
