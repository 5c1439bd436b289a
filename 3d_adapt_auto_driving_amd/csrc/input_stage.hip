// input_stage.hip -- the network-input stage of the KITTI loader on the device: lidar -> rectified camera frame,
// validity filter, and the 16384-point near/far random sampler
// (pointrcnn/lib/datasets/kitti_rcnn_dataset.py:249-324 get_lidar / get_valid_flag / the sampling block;
//  lib/utils/calibration.py:51-71 lidar_to_rect / rect_to_img).
//
// The reference does this per scene in numpy inside DataLoader workers; at the rate the rest of the path runs
// (~1.3 ms per scene) the host stage is the bottleneck, and the cross-domain clouds are 10x larger (~180 k raw
// points).  Here ONE workgroup per scene runs the whole stage on raw points that were uploaded as they sit in
// the .bin file:
//   1. every raw point: rect = R0 * (V2C * [p;1]), image projection with P2, validity (inside the image, depth >= 0,
//      inside PC_AREA_SCOPE), near (z < 40) / far class, and a 32-bit random key;
//   2. a random subset WITHOUT replacement of size K = the K smallest keys.  Keys are a bijection of the point index
//      (murmur3 finaliser of index ^ seed), hence distinct, so "key <= K-th smallest key" selects exactly K points;
//      the K-th smallest key is found by a 4-pass 8-bit radix select with an LDS histogram;
//   3. the reference's final np.random.shuffle = sorting the chosen indices by a second key (bitonic sort of
//      <= 16384 64-bit entries in LDS).
// Sampling rules as the reference: more valid points than npoints -> all far points (at most npoints_faraway of them,
// chosen at random) + a random subset of the near ones; fewer -> every point + random extra copies.
// The choice is random, so parity with the host sampler is distributional, not bitwise; the tests check the
// invariants (valid points only, exact counts per class, no duplicates unless the cloud is too small, determinism
// in the seed, uniformity).  The transform and the filter ARE bitwise the reference's (round 4): flags and rectified
// coordinates equal what the reference's own numpy code produced on the fixture scenes (tests/golden g11).
#include "common.hpp"
#include <math.h>
#include <algorithm>

namespace prcnn {

constexpr int IS_THREADS = 1024;
constexpr int IS_MAX_OUT = 16384;

struct SceneCalib {          // row-major, as calibration.py holds them
    float v2c[12];           // 3x4
    float r0[9];             // 3x3
    float p2[12];            // 3x4
    float img_h, img_w;
};

__device__ __forceinline__ unsigned fmix32(unsigned h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;                // a bijection on 32-bit words
}

// lidar -> rectified frame -> image, validity, near / far class of ONE raw point, in the arithmetic the reference's numpy code
// performs (pinned by tests/golden g11, reference-executed): ``np.dot`` of float32 operands is a chain of fused multiply-adds
// over the inner index, first term a plain product -- for the tiny (4,3) = V2C^T . R0^T product of Calibration.lidar_to_rect
// (calibration.py:51-59) as well as for the (n,4) . (4,3) products; rect_to_img divides by the rect depth (0 -> 1e-9,
// calibration.py:66-68) and subtracts P2[2][3] for the depth; get_valid_flag (kitti_rcnn_dataset.py:201-222) compares in f32.
struct LidarToRect {
    float m[4][3];           // np.dot(V2C.T, R0.T)
    __device__ void set(const SceneCalib &cb)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = __fmul_rn(cb.v2c[i], cb.r0[3 * j]);                       // V2C^T[i][0] * R0^T[0][j]
                acc = __fmaf_rn(cb.v2c[4 + i], cb.r0[3 * j + 1], acc);
                m[i][j] = __fmaf_rn(cb.v2c[8 + i], cb.r0[3 * j + 2], acc);
            }
    }
    __device__ __forceinline__ float row(int j, float px, float py, float pz) const
    {
        float acc = __fmul_rn(px, m[0][j]);
        acc = __fmaf_rn(py, m[1][j], acc);
        acc = __fmaf_rn(pz, m[2][j], acc);
        return __fadd_rn(acc, m[3][j]);                                               // fma(1, m, acc)
    }
};

// -> class: 0 = invalid, 1 = near (z < far_depth), 2 = far; x, y, z = rectified coordinates
__device__ __forceinline__ int classify_point(const SceneCalib &cb, const LidarToRect &l2r, int lidar_frame, int image_filter,
                                              const float *__restrict__ scope, float far_depth, float px, float py, float pz,
                                              float &x, float &y, float &z)
{
    x = px; y = py; z = pz;
    if (lidar_frame) { x = l2r.row(0, px, py, pz); y = l2r.row(1, px, py, pz); z = l2r.row(2, px, py, pz); }
    bool ok = true;
    if (image_filter) {
        float hu = __fmul_rn(x, cb.p2[0]); hu = __fmaf_rn(y, cb.p2[1], hu); hu = __fmaf_rn(z, cb.p2[2], hu); hu = __fadd_rn(hu, cb.p2[3]);
        float hv = __fmul_rn(x, cb.p2[4]); hv = __fmaf_rn(y, cb.p2[5], hv); hv = __fmaf_rn(z, cb.p2[6], hv); hv = __fadd_rn(hv, cb.p2[7]);
        float hw = __fmul_rn(x, cb.p2[8]); hw = __fmaf_rn(y, cb.p2[9], hw); hw = __fmaf_rn(z, cb.p2[10], hw); hw = __fadd_rn(hw, cb.p2[11]);
        const float zz = (z == 0.f) ? 1e-9f : z;
        const float u = __fdiv_rn(hu, zz), v = __fdiv_rn(hv, zz);
        const float depth = __fsub_rn(hw, cb.p2[11]);
        ok = u >= 0.f && u < cb.img_w && v >= 0.f && v < cb.img_h && depth >= 0.f;
    }
    if (scope)
        ok = ok && x >= scope[0] && x <= scope[1] && y >= scope[2] && y <= scope[3] && z >= scope[4] && z <= scope[5];
    return !ok ? 0 : (z < far_depth ? 1 : 2);
}

__device__ __forceinline__ int block_sum(int v, int *red)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int i = 0; i < IS_THREADS / 64; ++i) s += red[i];
    return s;
}

// K-th smallest key (1-based K) among the points whose class bit is set in `classes` (bit c = class c); keys are distinct.
__device__ unsigned radix_select(int n, const unsigned *__restrict__ key, const unsigned char *__restrict__ cls, int classes,
                                 int K, unsigned *hist, unsigned *bcast)
{
    unsigned prefix = 0, mask = 0;
    int remaining = K;
    for (int shift = 24; shift >= 0; shift -= 8) {
        __syncthreads();
        for (int i = threadIdx.x; i < 256; i += IS_THREADS) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += IS_THREADS)
            if (((classes >> cls[i]) & 1) && (key[i] & mask) == prefix) atomicAdd(&hist[(key[i] >> shift) & 255u], 1u);
        __syncthreads();
        if (threadIdx.x == 0) {
            int cum = 0, bin = 0;
            for (; bin < 256; ++bin) {
                if (cum + (int)hist[bin] >= remaining) break;
                cum += (int)hist[bin];
            }
            bcast[0] = (unsigned)bin;
            bcast[1] = (unsigned)cum;
        }
        __syncthreads();
        prefix |= bcast[0] << shift;
        mask |= 255u << shift;
        remaining -= (int)bcast[1];
    }
    return prefix;
}

__global__ __launch_bounds__(IS_THREADS) void input_stage_kernel(
    int n_max, int stride, int lidar_frame, int image_filter, const int *__restrict__ counts, const float *__restrict__ raw,
    const SceneCalib *__restrict__ calib, const float *__restrict__ scope, int npoints, int npad, float far_depth,
    int npoints_faraway, const unsigned long long *__restrict__ seeds, float *__restrict__ rect,
    unsigned *__restrict__ key, unsigned char *__restrict__ cls, float *__restrict__ out, int *__restrict__ stats, int *__restrict__ choice)
{
    extern __shared__ unsigned long long entries[];      // npad sort entries: (shuffle key << 32) | point index
    __shared__ unsigned hist[256];
    __shared__ unsigned bcast[2];
    __shared__ int red[IS_THREADS / 64];
    __shared__ int cursor;

    const int b = blockIdx.x, t = threadIdx.x;
    const int n = min(counts[b], n_max);
    const float *__restrict__ src = raw + (long)b * n_max * stride;
    float *__restrict__ rc = rect + (long)b * n_max * 3;
    unsigned *__restrict__ ky = key + (long)b * n_max;
    unsigned char *__restrict__ cl = cls + (long)b * n_max;
    const unsigned long long seed = seeds[b];
    const unsigned sa = fmix32((unsigned)seed), sb = fmix32((unsigned)(seed >> 32) ^ 0x9e3779b9u), sc = sa ^ 0x7f4a7c15u;
    const SceneCalib cb = calib[b];
    LidarToRect l2r;
    l2r.set(cb);

    // ---- 1. transform + filter + classify
    int n_near = 0, n_far = 0;
    for (int i = t; i < n; i += IS_THREADS) {
        const float px = src[(long)i * stride], py = src[(long)i * stride + 1], pz = src[(long)i * stride + 2];
        float x, y, z;
        const int c = classify_point(cb, l2r, lidar_frame, image_filter, scope, far_depth, px, py, pz, x, y, z);
        rc[3 * (long)i] = x; rc[3 * (long)i + 1] = y; rc[3 * (long)i + 2] = z;
        ky[i] = fmix32((unsigned)i ^ sa);
        cl[i] = (unsigned char)c;
        n_near += (c == 1);
        n_far += (c == 2);
    }
    n_near = block_sum(n_near, red);
    n_far = block_sum(n_far, red);
    const int n_valid = n_near + n_far;
    if (t == 0) { stats[3 * b] = n_valid; stats[3 * b + 1] = n_near; stats[3 * b + 2] = n_far; cursor = 0; }
    float *__restrict__ o = out + (long)b * npoints * 3;
    if (n_valid == 0) {
        for (int j = t; j < npoints * 3; j += IS_THREADS) o[j] = 0.f;
        if (choice)
            for (int j = t; j < npoints; j += IS_THREADS) choice[(long)b * npoints + j] = -1;
        return;
    }
    __syncthreads();          // rc / ky / cl written by this block are read below

    // ---- 2. who is taken.  Per class: nothing / everything / keys <= threshold.  `extra` further entries are copies.
    enum { NONE = 0, ALL = 1, THR = 2 };
    int near_mode = ALL, far_mode = ALL, extra = 0;
    unsigned thr_near = 0, thr_far = 0, thr_extra = 0;
    bool extra_distinct = false;               // extras = the `extra` smallest keys over all valid points (no replacement)
    int extra_cls = 0;                         // extras with replacement are drawn from this class (0 = any taken point)
    if (n_valid > npoints) {
        const int far_keep = min(min(n_far, npoints_faraway), npoints);   // (a cap above npoints would ask for a negative near count)
        if (far_keep == 0) far_mode = NONE;
        else if (far_keep < n_far) { far_mode = THR; thr_far = radix_select(n, ky, cl, 1 << 2, far_keep, hist, bcast); }
        const int need_near = npoints - far_keep;
        if (need_near == 0) near_mode = NONE;
        else if (need_near < n_near) { near_mode = THR; thr_near = radix_select(n, ky, cl, 1 << 1, need_near, hist, bcast); }
        else if (need_near > n_near) { extra = need_near - n_near; extra_cls = n_near > 0 ? 1 : 0; }   // every near point + copies
    } else {
        extra = npoints - n_valid;
        if (extra > 0 && extra <= n_valid) {
            extra_distinct = true;
            if (extra < n_valid) thr_extra = radix_select(n, ky, cl, (1 << 1) | (1 << 2), extra, hist, bcast);
            else thr_extra = 0xffffffffu;
        }
    }
    __syncthreads();

    // ---- 3. compaction into sort entries
    for (int j = t; j < npad; j += IS_THREADS) entries[j] = ~0ull;
    __syncthreads();
    for (int i = t; i < n; i += IS_THREADS) {
        const int c = cl[i];
        if (c == 0) continue;
        const int mode = (c == 1) ? near_mode : far_mode;
        const unsigned thr = (c == 1) ? thr_near : thr_far;
        if (mode == ALL || (mode == THR && ky[i] <= thr)) {
            const int pos = atomicAdd(&cursor, 1);
            if (pos < npoints) entries[pos] = ((unsigned long long)fmix32((unsigned)i ^ sb) << 32) | (unsigned)i;
        }
        if (extra_distinct && ky[i] <= thr_extra) {
            const int pos = atomicAdd(&cursor, 1);
            if (pos < npoints) entries[pos] = ((unsigned long long)fmix32((unsigned)i ^ sc) << 32) | (unsigned)i;
        }
    }
    __syncthreads();
    if (extra > 0 && !extra_distinct) {
        // copies WITH replacement: the e-th extra is the (hash(e) mod pool)-th taken entry of the pool class.  The
        // first base_count entries are exactly the taken points; when the pool is one class, entries of the other
        // class are skipped by re-drawing (bounded), falling back to any taken entry.
        const int have = min(cursor, npoints);
        for (int e = t; e < extra; e += IS_THREADS) {
            unsigned h = fmix32((unsigned)e ^ sc);
            int pick = (int)(h % (unsigned)have);
            for (int tries = 0; tries < 16 && extra_cls && cl[(unsigned)entries[pick]] != extra_cls; ++tries) {
                h = fmix32(h + 0x632be5abu);
                pick = (int)(h % (unsigned)have);
            }
            const unsigned i = (unsigned)entries[pick];
            entries[have + e] = ((unsigned long long)fmix32(h ^ sb ^ 0x51ed270bu) << 32) | i;
        }
    }
    __syncthreads();

    // ---- 4. shuffle = sort by the random 32-bit key (ties by index)
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j >= 1; j >>= 1) {
            for (int i = t; i < npad / 2; i += IS_THREADS) {
                const int lo = ((i / j) * 2 * j) + (i % j), hi = lo + j;
                const bool up = ((lo & k) == 0);
                const unsigned long long a = entries[lo], c = entries[hi];
                if ((a > c) == up) { entries[lo] = c; entries[hi] = a; }
            }
            __syncthreads();
        }

    // ---- 5. gather
    for (int j = t; j < npoints; j += IS_THREADS) {
        const unsigned long long e = entries[j];
        if (e == ~0ull) { o[3 * j] = 0.f; o[3 * j + 1] = 0.f; o[3 * j + 2] = 0.f; continue; }   // cannot happen (count == npoints)
        const unsigned i = (unsigned)e;
        o[3 * j] = rc[3 * (long)i]; o[3 * j + 1] = rc[3 * (long)i + 1]; o[3 * j + 2] = rc[3 * (long)i + 2];
        if (choice) choice[(long)b * npoints + j] = (int)i;
    }
}

// the front half alone (get_valid_flag + lidar_to_rect as an operator): rect (b, n_max, 3), cls (b, n_max) u8
__global__ __launch_bounds__(256) void valid_flags_kernel(int n_max, int stride, int lidar_frame, int image_filter,
                                                           const int *__restrict__ counts, const float *__restrict__ raw,
                                                           const SceneCalib *__restrict__ calib, const float *__restrict__ scope,
                                                           float far_depth, float *__restrict__ rect, unsigned char *__restrict__ cls)
{
    const int b = blockIdx.y;
    const int n = min(counts[b], n_max);
    const SceneCalib cb = calib[b];
    LidarToRect l2r;
    l2r.set(cb);
    const float *__restrict__ src = raw + (long)b * n_max * stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_max; i += gridDim.x * blockDim.x) {
        float x = 0.f, y = 0.f, z = 0.f;
        int c = 0;
        if (i < n)
            c = classify_point(cb, l2r, lidar_frame, image_filter, scope, far_depth, src[(long)i * stride], src[(long)i * stride + 1],
                               src[(long)i * stride + 2], x, y, z);
        if (rect) { float *r = rect + ((long)b * n_max + i) * 3; r[0] = x; r[1] = y; r[2] = z; }
        cls[(long)b * n_max + i] = (unsigned char)c;
    }
}

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace prcnn

using namespace prcnn;

// raw (b, n_max, stride) f32 with stride 3 or 4 (x, y, z[, intensity]) as read from velodyne/*.bin (lidar_frame = 1)
// or already in the rectified camera frame (lidar_frame = 0: V2C / R0 are ignored); counts (b) i32 = points per scene;
// image_filter = 0 skips the in-image / depth test (synthetic clouds); calib (b, 35) f32 = V2C (3x4) | R0 (3x3) | P2 (3x4) |
// image height, width; scope = 6 HOST floats x0,x1,y0,y1,z0,z1
// (PC_AREA_SCOPE) or NULL; seeds (b) u64 DEVICE.  -> out (b, npoints, 3), stats (b, 3) i32 = #valid, #near, #far,
// choice (b, npoints) i32 = raw index of every output point (optional, may be NULL; -1 for an empty scene).
extern "C" int prcnn_input_stage(int b, int n_max, int stride, int lidar_frame, int image_filter, const int *counts, const float *raw,
                                 const float *calib, const float *scope_host, int npoints, float far_depth,
                                 int npoints_faraway, const unsigned long long *seeds, float *out, int *stats, int *choice,
                                 void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n_max >= 0 && (stride == 3 || stride == 4), "input_stage: bad sizes (stride 3 or 4)");
    PRCNN_REQUIRE(npoints > 0 && npoints <= IS_MAX_OUT, "input_stage: npoints=%d not in 1..%d", npoints, IS_MAX_OUT);
    PRCNN_REQUIRE(npoints_faraway >= 0, "input_stage: bad npoints_faraway");
    PRCNN_REQUIRE(sizeof(SceneCalib) == 35 * sizeof(float), "input_stage: calib layout");
    if (b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(counts && calib && seeds && out && stats && (raw || n_max == 0), "input_stage: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int npad = 1;
    while (npad < npoints) npad <<= 1;
    const size_t o_rect = 0;
    const size_t o_key = o_rect + up256((size_t)b * n_max * 3 * sizeof(float));
    const size_t o_cls = o_key + up256((size_t)b * n_max * sizeof(unsigned));
    const size_t o_scope = o_cls + up256((size_t)b * n_max);
    const size_t need = o_scope + 256;
    char *base = scratch_for(st, need, 5);
    if (!base) { set_error("input_stage: cannot allocate %zu bytes of scratch", need); return PRCNN_ELAUNCH; }
    float *scope_dev = nullptr;
    if (scope_host) {
        scope_dev = (float *)(base + o_scope);
        if (hipMemcpyAsync(scope_dev, scope_host, 6 * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) {
            set_error("input_stage: scope upload failed");
            return PRCNN_ELAUNCH;
        }
    }
    const size_t lds = (size_t)npad * sizeof(unsigned long long);
    {
        const int rc = ensure_dynamic_lds((const void *)input_stage_kernel, 140 * 1024, "input_stage");
        if (rc != PRCNN_OK) return rc;
    }
    hipLaunchKernelGGL(input_stage_kernel, dim3(b), dim3(IS_THREADS), lds, st, n_max, stride, lidar_frame, image_filter, counts, raw,
                       (const SceneCalib *)calib, scope_dev, npoints, npad, far_depth, npoints_faraway, seeds,
                       (float *)(base + o_rect), (unsigned *)(base + o_key), (unsigned char *)(base + o_cls), out, stats, choice);
    return check_launch("input_stage");
}


// get_valid_flag (kitti_rcnn_dataset.py:201-222) + Calibration.lidar_to_rect / rect_to_img (calibration.py:51-71) for whole
// batches: the front half of prcnn_input_stage as an operator of its own, same arguments.  -> cls (b, n_max) u8: 0 = not valid,
// 1 = valid with z < far_depth, 2 = valid beyond (rows >= counts[b]: 0); rect (b, n_max, 3) f32 rectified coordinates (may be
// NULL).  Bit-identical to the reference's numpy results (float32 np.dot = fma chains; tests/golden g11).
extern "C" int prcnn_valid_flags(int b, int n_max, int stride, int lidar_frame, int image_filter, const int *counts, const float *raw,
                                 const float *calib, const float *scope_host, float far_depth, float *rect, unsigned char *cls,
                                 void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n_max >= 0 && (stride == 3 || stride == 4), "valid_flags: bad sizes (stride 3 or 4)");
    if (b == 0 || n_max == 0) return PRCNN_OK;
    PRCNN_REQUIRE(counts && calib && raw && cls, "valid_flags: null pointer");
    hipStream_t st = (hipStream_t)stream;
    float *scope_dev = nullptr;
    if (scope_host) {
        char *base = scratch_for(st, 256, 5);
        if (!base) { set_error("valid_flags: cannot allocate scratch"); return PRCNN_ELAUNCH; }
        scope_dev = (float *)base;
        if (hipMemcpyAsync(scope_dev, scope_host, 6 * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) {
            set_error("valid_flags: scope upload failed");
            return PRCNN_ELAUNCH;
        }
    }
    const int gx = (int)std::min<long>(1024, ((long)n_max + 255) / 256);
    hipLaunchKernelGGL(valid_flags_kernel, dim3(gx, b), dim3(256), 0, st, n_max, stride, lidar_frame, image_filter, counts, raw,
                       (const SceneCalib *)calib, scope_dev, far_depth, rect, cls);
    return check_launch("valid_flags");
}
