// iou3d_cuda as a compiled extension module: the four entry points of the reference's iou3d.cpp:31,52,73,123 (bound at :174-179).
// Every tensor must be a contiguous device tensor (iou3d.cpp:7-9) except `keep`, a CPU int64 tensor that receives the kept
// indices; the two NMS calls block (the reference's cudaMemcpy does) and return the count.
#include "binding_common.h"

int boxes_overlap_bev_gpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_overlap)
{
    PRCNN_CHECK_DEV(boxes_a); PRCNN_CHECK_DEV(boxes_b); PRCNN_CHECK_DEV(ans_overlap);
    PRCNN_CHECK_F32(boxes_a); PRCNN_CHECK_F32(boxes_b); PRCNN_CHECK_F32(ans_overlap);
    prcnn_ok(prcnn_boxes_overlap_bev((int)boxes_a.size(0), boxes_a.data_ptr<float>(), (int)boxes_b.size(0), boxes_b.data_ptr<float>(),
                                     ans_overlap.data_ptr<float>(), cur_stream(boxes_a)));
    return 1;
}

int boxes_iou_bev_gpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_iou)
{
    PRCNN_CHECK_DEV(boxes_a); PRCNN_CHECK_DEV(boxes_b); PRCNN_CHECK_DEV(ans_iou);
    PRCNN_CHECK_F32(boxes_a); PRCNN_CHECK_F32(boxes_b); PRCNN_CHECK_F32(ans_iou);
    prcnn_ok(prcnn_boxes_iou_bev((int)boxes_a.size(0), boxes_a.data_ptr<float>(), (int)boxes_b.size(0), boxes_b.data_ptr<float>(),
                                 ans_iou.data_ptr<float>(), cur_stream(boxes_a)));
    return 1;
}

static int nms_common(at::Tensor boxes, at::Tensor keep, float thresh, bool rotated)
{
    PRCNN_CHECK_DEV(boxes); PRCNN_CHECK_F32(boxes);
    TORCH_CHECK(!keep.is_cuda() && keep.is_contiguous() && keep.scalar_type() == at::kLong, "keep must be a contiguous CPU int64 tensor");
    TORCH_CHECK(keep.numel() >= boxes.size(0), "keep is shorter than the box list");
    const int rc = (rotated ? prcnn_nms : prcnn_nms_normal)((int)boxes.size(0), boxes.data_ptr<float>(), (long long *)keep.data_ptr<int64_t>(),
                                                           thresh, cur_stream(boxes));
    TORCH_CHECK(rc >= 0, "libprcnn_hip: ", prcnn_last_error());
    return rc;
}

int nms_gpu(at::Tensor boxes, at::Tensor keep, float nms_overlap_thresh) { return nms_common(boxes, keep, nms_overlap_thresh, true); }
int nms_normal_gpu(at::Tensor boxes, at::Tensor keep, float nms_overlap_thresh) { return nms_common(boxes, keep, nms_overlap_thresh, false); }

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("boxes_overlap_bev_gpu", &boxes_overlap_bev_gpu, "oriented boxes overlap");
    m.def("boxes_iou_bev_gpu", &boxes_iou_bev_gpu, "oriented boxes iou");
    m.def("nms_gpu", &nms_gpu, "oriented nms gpu");
    m.def("nms_normal_gpu", &nms_normal_gpu, "nms gpu");
}
