// rbox_iou.hpp -- the BEV box geometry shared by the NMS kernels (iou3d.hip) and the fused final stage (proposal.hip): rotated overlap /
// IoU of two BEV boxes as the reference computes them (lib/utils/iou3d/src/iou3d_kernel.cu:50-221), the axis-aligned IoU (:295-303),
// and the (row, column) suppression test of the NMS kernels.  Moved out of iou3d.hip unchanged (round 4).
#pragma once
#include "common.hpp"
#include <math.h>

namespace prcnn {

struct P2 {
    float x, y;
};

#define IOU_EPS 1e-8f

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0)
{
    return __fsub_rn(__fmul_rn(p1.x - p0.x, p2.y - p0.y), __fmul_rn(p2.x - p0.x, p1.y - p0.y));
}

// iou3d_kernel.cu:73-106: (p0,p1) is an edge of a, (q0,q1) an edge of b
__device__ __forceinline__ bool seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans)
{
    if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
        return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(__fmul_rn(s1, s2) > 0 && __fmul_rn(s3, s4) > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans.x = __fdiv_rn(__fsub_rn(__fmul_rn(s5, q0.x), __fmul_rn(s1, q1.x)), s5 - s1);
        ans.y = __fdiv_rn(__fsub_rn(__fmul_rn(s5, q0.y), __fmul_rn(s1, q1.y)), s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = __fsub_rn(__fmul_rn(p0.x, p1.y), __fmul_rn(p1.x, p0.y));
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = __fsub_rn(__fmul_rn(q0.x, q1.y), __fmul_rn(q1.x, q0.y));
        const float D = __fsub_rn(__fmul_rn(a0, b1), __fmul_rn(a1, b0));
        ans.x = __fdiv_rn(__fsub_rn(__fmul_rn(b0, c1), __fmul_rn(b1, c0)), D);
        ans.y = __fdiv_rn(__fsub_rn(__fmul_rn(a1, c0), __fmul_rn(a0, c1)), D);
    }
    return true;
}

// rotate p about c by (cosv, sinv): iou3d_kernel.cu:92-96
__device__ __forceinline__ P2 rot_about(P2 c, float cosv, float sinv, P2 p)
{
    P2 r;
    r.x = __fadd_rn(__fadd_rn(__fmul_rn(p.x - c.x, cosv), __fmul_rn(p.y - c.y, sinv)), c.x);
    r.y = __fadd_rn(__fadd_rn(__fmul_rn(-(p.x - c.x), sinv), __fmul_rn(p.y - c.y, cosv)), c.y);
    return r;
}

// iou3d_kernel.cu:50-65 with cos(-t) = cos t, sin(-t) = -sin t
__device__ __forceinline__ bool corner_in_box(const float *box, float cosv, float sinv, P2 p)
{
    const float MARGIN = 1e-5f;
    P2 c = { (box[0] + box[2]) / 2, (box[1] + box[3]) / 2 };
    const P2 r = rot_about(c, cosv, -sinv, p);
    return r.x > box[0] - MARGIN && r.x < box[2] + MARGIN && r.y > box[1] - MARGIN && r.y < box[3] + MARGIN;
}

struct RBox {
    float v[5];
    float cosv, sinv;  // of v[4], evaluated once per box
};

__device__ __forceinline__ RBox make_rbox(const float *p)
{
    RBox r;
#pragma unroll
    for (int i = 0; i < 5; ++i) r.v[i] = p[i];
    r.cosv = cos_f32(p[4]);
    r.sinv = sin_f32(p[4]);
    return r;
}

// where the clipped polygon's vertices live while they are collected, sorted and summed.  The reference keeps them in local arrays
// (iou3d_kernel.cu:129-131) that are indexed with run-time counts: on gfx950 such arrays go to SCRATCH memory (208 bytes per thread,
// every access a trip through the vector memory pipe).  PolyLds puts them into LDS instead, element k of thread t at [k][t]
// (conflict-free across a wave), for the kernels on the step's critical path; PolyPriv is the local-array form.
struct PolyPriv {
    float px[24], py[24], an[24];
    __device__ __forceinline__ float &x(int k) { return px[k]; }
    __device__ __forceinline__ float &y(int k) { return py[k]; }
    __device__ __forceinline__ float &a(int k) { return an[k]; }
};
struct PolyLds {
    float *base; int stride;                 // base = this thread's first element, stride = threads of the workgroup
    __device__ __forceinline__ float &x(int k) { return base[(3 * k) * stride]; }
    __device__ __forceinline__ float &y(int k) { return base[(3 * k + 1) * stride]; }
    __device__ __forceinline__ float &a(int k) { return base[(3 * k + 2) * stride]; }
};
constexpr int POLY_LDS_FLOATS = 72;          // per thread

// iou3d_kernel.cu:108-212
template <class POLY>
__device__ __forceinline__ float rbox_overlap_in(const RBox &A, const RBox &B, POLY &poly)
{
    const P2 ca = { (A.v[0] + A.v[2]) / 2, (A.v[1] + A.v[3]) / 2 };
    const P2 cb = { (B.v[0] + B.v[2]) / 2, (B.v[1] + B.v[3]) / 2 };
    P2 pa[5], pb[5];
    pa[0] = rot_about(ca, A.cosv, A.sinv, P2{ A.v[0], A.v[1] });
    pa[1] = rot_about(ca, A.cosv, A.sinv, P2{ A.v[2], A.v[1] });
    pa[2] = rot_about(ca, A.cosv, A.sinv, P2{ A.v[2], A.v[3] });
    pa[3] = rot_about(ca, A.cosv, A.sinv, P2{ A.v[0], A.v[3] });
    pb[0] = rot_about(cb, B.cosv, B.sinv, P2{ B.v[0], B.v[1] });
    pb[1] = rot_about(cb, B.cosv, B.sinv, P2{ B.v[2], B.v[1] });
    pb[2] = rot_about(cb, B.cosv, B.sinv, P2{ B.v[2], B.v[3] });
    pb[3] = rot_about(cb, B.cosv, B.sinv, P2{ B.v[0], B.v[3] });
    pa[4] = pa[0];
    pb[4] = pb[0];

    P2 centre = { 0.f, 0.f };
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            P2 hit;
            if (seg_intersection(pa[i + 1], pa[i], pb[j + 1], pb[j], hit)) {
                poly.x(cnt) = hit.x; poly.y(cnt) = hit.y;
                centre.x = centre.x + hit.x;
                centre.y = centre.y + hit.y;
                ++cnt;
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (corner_in_box(A.v, A.cosv, A.sinv, pb[k])) {
            centre.x = centre.x + pb[k].x; centre.y = centre.y + pb[k].y;
            poly.x(cnt) = pb[k].x; poly.y(cnt) = pb[k].y; ++cnt;
        }
        if (corner_in_box(B.v, B.cosv, B.sinv, pa[k])) {
            centre.x = centre.x + pa[k].x; centre.y = centre.y + pa[k].y;
            poly.x(cnt) = pa[k].x; poly.y(cnt) = pa[k].y; ++cnt;
        }
    }
    if (cnt < 3) return 0.f;  // fewer than 3 vertices: the shoelace sum below is exactly 0
    centre.x = __fdiv_rn(centre.x, (float)cnt);
    centre.y = __fdiv_rn(centre.y, (float)cnt);

    for (int i = 0; i < cnt; ++i) poly.a(i) = atan2_f32(poly.y(i) - centre.y, poly.x(i) - centre.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float a0 = poly.a(i), a1 = poly.a(i + 1);
            if (a0 > a1) {
                const float tx = poly.x(i), ty = poly.y(i);
                poly.x(i) = poly.x(i + 1); poly.y(i) = poly.y(i + 1);
                poly.x(i + 1) = tx; poly.y(i + 1) = ty;
                poly.a(i) = a1; poly.a(i + 1) = a0;
            }
        }
    float area = 0.f;
    const float x0 = poly.x(0), y0 = poly.y(0);
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = poly.x(k) - x0, uy = poly.y(k) - y0;
        const float vx = poly.x(k + 1) - x0, vy = poly.y(k + 1) - y0;
        area = __fadd_rn(area, __fsub_rn(__fmul_rn(ux, vy), __fmul_rn(uy, vx)));
    }
    return fabsf(area) * 0.5f;
}

__device__ inline float rbox_overlap(const RBox &A, const RBox &B)
{
    PolyPriv poly;
    return rbox_overlap_in(A, B, poly);
}

// iou3d_kernel.cu:214-221
__device__ __forceinline__ float rbox_iou(const RBox &A, const RBox &B)
{
    const float sa = __fmul_rn(A.v[2] - A.v[0], A.v[3] - A.v[1]);
    const float sb = __fmul_rn(B.v[2] - B.v[0], B.v[3] - B.v[1]);
    const float so = rbox_overlap(A, B);
    return __fdiv_rn(so, fmaxf(__fsub_rn(__fadd_rn(sa, sb), so), IOU_EPS));
}

__device__ __forceinline__ float rbox_iou_lds(const RBox &A, const RBox &B, float *scr, int stride)
{
    const float sa = __fmul_rn(A.v[2] - A.v[0], A.v[3] - A.v[1]);
    const float sb = __fmul_rn(B.v[2] - B.v[0], B.v[3] - B.v[1]);
    PolyLds poly = {scr, stride};
    const float so = rbox_overlap_in(A, B, poly);
    return __fdiv_rn(so, fmaxf(__fsub_rn(__fadd_rn(sa, sb), so), IOU_EPS));
}

// iou3d_kernel.cu:295-303
__device__ __forceinline__ float aabox_iou(const float *a, const float *b)
{
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float interS = __fmul_rn(width, height);
    const float Sa = __fmul_rn(a[2] - a[0], a[3] - a[1]);
    const float Sb = __fmul_rn(b[2] - b[0], b[3] - b[1]);
    return __fdiv_rn(interS, fmaxf(__fsub_rn(__fadd_rn(Sa, Sb), interS), IOU_EPS));
}

template <bool ROTATED, bool LDS_POLY = false>
__device__ __forceinline__ bool suppresses(const float *s_row, int r, const RBox &C, float thresh, float *scr = nullptr, int stride = 0)
{
    if (ROTATED) {
        RBox R;
#pragma unroll
        for (int q = 0; q < 5; ++q) R.v[q] = s_row[r * 7 + q];
        R.cosv = s_row[r * 7 + 5];
        R.sinv = s_row[r * 7 + 6];
        if constexpr (LDS_POLY) return rbox_iou_lds(R, C, scr, stride) > thresh;
        else return rbox_iou(R, C) > thresh;  // (row, column) order as nms_kernel :285
    }
    return aabox_iou(&s_row[r * 7], C.v) > thresh;
}

template <bool ROTATED>
__device__ __forceinline__ RBox load_col(const float *p)
{
    if (ROTATED) return make_rbox(p);
    RBox r;
#pragma unroll
    for (int i = 0; i < 5; ++i) r.v[i] = p[i];
    r.cosv = 1.f; r.sinv = 0.f;
    return r;
}


// Can the two boxes touch at all?  A rotated BEV box lies inside the circle around its centre through its corners (rotation about the
// centre, iou3d_kernel.cu:92-96, keeps the half diagonal); circles further apart than the sum of their radii (with a margin far above
// any f32 rounding of the corner arithmetic: 0.1 % + 1 cm) hold boxes whose clipped polygon has no vertex -- rbox_overlap_in then
// returns exactly 0 and the pair is not a suppression for any threshold >= 0.  false for anything non-finite (such pairs are evaluated).
__device__ __forceinline__ bool rbox_far_apart(const float *a, const float *b)
{
    const float ax = (a[0] + a[2]) * 0.5f, ay = (a[1] + a[3]) * 0.5f, bx = (b[0] + b[2]) * 0.5f, by = (b[1] + b[3]) * 0.5f;
    const float ra = 0.5f * sqrtf((a[2] - a[0]) * (a[2] - a[0]) + (a[3] - a[1]) * (a[3] - a[1]));
    const float rb = 0.5f * sqrtf((b[2] - b[0]) * (b[2] - b[0]) + (b[3] - b[1]) * (b[3] - b[1]));
    const float reach = (ra + rb) * 1.001f + 0.01f;
    const float dx = ax - bx, dy = ay - by;
    return dx * dx + dy * dy > reach * reach;
}

}  // namespace prcnn
