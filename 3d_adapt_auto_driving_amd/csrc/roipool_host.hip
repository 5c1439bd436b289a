// roipool_host.hip -- the two HOST utilities of the reference's roipool3d extension
// (lib/utils/roipool3d/src/roipool3d.cpp:82-195: pts_in_boxes3d_cpu, roipool3d_cpu), which take CPU tensors and
// serve its dataset / GT-database code (kitti_rcnn_dataset.py:507, generate_gt_database.py:75).  They are part of the
// module boundary, not a fallback of the device path: the device entry points still refuse
// host pointers' tensors at the Python layer.  Host code only; unbatched, one cloud.
//
// Arithmetic as the reference's C++: the box centre and the half extents are formed in double (h / 2.0), cos / sin of the
// float angle are the float libm functions (host glibc, as in the reference binary), the rotation is float arithmetic
// without contraction.
#include "common.hpp"
#include <math.h>
#include <string.h>

namespace prcnn {

static inline int point_in_box(float x, float y, float z, const float *b)
{
    const float cx = b[0], bottom_y = b[1], cz = b[2], h = b[3], w = b[4], l = b[5], angle = b[6];
    const float cy = (float)((double)bottom_y - (double)h / 2.0);
    if (fabsf(x - cx) > 10.0f || (double)fabsf(y - cy) > (double)h / 2.0 || fabsf(z - cz) > 10.0f) return 0;
    const float cosa = cosf(angle), sina = sinf(angle);     // C++ overload resolution on a float: the float libm functions
    const float dx = x - cx, dz = z - cz;
    const float xr = dx * cosa + dz * (-sina);
    const float zr = dx * sina + dz * cosa;
    return ((double)xr >= -(double)l / 2.0) & ((double)xr <= (double)l / 2.0) & ((double)zr >= -(double)w / 2.0) &
           ((double)zr <= (double)w / 2.0);
}

}  // namespace prcnn

using namespace prcnn;

/* pts_in_boxes3d_cpu  roipool3d.cpp:97-125: flags (boxes_num, pts_num) i64 in {0,1}; HOST pointers. */
extern "C" int prcnn_host_pts_in_boxes3d(int boxes_num, int pts_num, const float *pts, const float *boxes3d, long long *flags)
{
    PRCNN_REQUIRE(boxes_num >= 0 && pts_num >= 0, "host_pts_in_boxes3d: bad sizes");
    if (boxes_num == 0 || pts_num == 0) return PRCNN_OK;
    PRCNN_REQUIRE(pts && boxes3d && flags, "host_pts_in_boxes3d: null pointer");
    for (int i = 0; i < boxes_num; ++i)
        for (int j = 0; j < pts_num; ++j)
            flags[(long)i * pts_num + j] = point_in_box(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], boxes3d + 7 * i);
    return PRCNN_OK;
}

/* roipool3d_cpu  roipool3d.cpp:127-195: first `sampled` in-box points per box in index order, wrap-around duplication,
 * empty boxes flagged and their rows left as the caller initialised them.  pooled_pts (boxes_num, sampled, 3),
 * pooled_features (boxes_num, sampled, feature_len), empty (boxes_num) i64; HOST pointers. */
extern "C" int prcnn_host_roipool3d(int boxes_num, int pts_num, int feature_len, int sampled, const float *pts,
                                    const float *boxes3d, const float *pts_feature, float *pooled_pts,
                                    float *pooled_features, long long *empty)
{
    PRCNN_REQUIRE(boxes_num >= 0 && pts_num >= 0 && feature_len >= 0 && sampled >= 0, "host_roipool3d: bad sizes");
    if (boxes_num == 0) return PRCNN_OK;
    PRCNN_REQUIRE(boxes3d && empty && (sampled == 0 || (pooled_pts && (pooled_features || feature_len == 0))) &&
                      (pts_num == 0 || (pts && (pts_feature || feature_len == 0))), "host_roipool3d: null pointer");
    memset(empty, 0, sizeof(long long) * (size_t)boxes_num);
    for (int i = 0; i < boxes_num; ++i) {
        float *pp = pooled_pts + (size_t)i * sampled * 3;
        float *pf = pooled_features + (size_t)i * sampled * feature_len;
        int cnt = 0;
        for (int j = 0; j < pts_num && cnt < sampled; ++j) {
            if (!point_in_box(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], boxes3d + 7 * i)) continue;
            memcpy(pp + 3 * (size_t)cnt, pts + 3 * (size_t)j, 3 * sizeof(float));
            if (feature_len) memcpy(pf + (size_t)cnt * feature_len, pts_feature + (size_t)j * feature_len, sizeof(float) * feature_len);
            ++cnt;
        }
        if (cnt == 0) { empty[i] = 1; continue; }
        for (int j = cnt; j < sampled; ++j) {
            memcpy(pp + 3 * (size_t)j, pp + 3 * (size_t)(j % cnt), 3 * sizeof(float));
            if (feature_len) memcpy(pf + (size_t)j * feature_len, pf + (size_t)(j % cnt) * feature_len, sizeof(float) * feature_len);
        }
    }
    return PRCNN_OK;
}
