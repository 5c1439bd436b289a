// roipool3d_cuda as a compiled extension module: forward / forward_slow (roipool3d.cpp:48-79 and :15-46: same result, the
// reference keeps the slow one for comparison) on device tensors and the two host utilities pts_in_boxes3d_cpu /
// roipool3d_cpu (:97-195) on CPU tensors, bound as at roipool3d.cpp:198-203.
#include "binding_common.h"

int roipool3d_gpu(at::Tensor xyz, at::Tensor boxes3d, at::Tensor pts_feature, at::Tensor pooled_features, at::Tensor pooled_empty_flag)
{
    PRCNN_CHECK_DEV(xyz); PRCNN_CHECK_DEV(boxes3d); PRCNN_CHECK_DEV(pts_feature); PRCNN_CHECK_DEV(pooled_features); PRCNN_CHECK_DEV(pooled_empty_flag);
    PRCNN_CHECK_F32(xyz); PRCNN_CHECK_F32(boxes3d); PRCNN_CHECK_F32(pts_feature); PRCNN_CHECK_F32(pooled_features); PRCNN_CHECK_I32(pooled_empty_flag);
    prcnn_ok(prcnn_roipool3d((int)xyz.size(0), (int)xyz.size(1), (int)boxes3d.size(1), (int)pts_feature.size(2), (int)pooled_features.size(2),
                             xyz.data_ptr<float>(), boxes3d.data_ptr<float>(), pts_feature.data_ptr<float>(),
                             pooled_features.data_ptr<float>(), pooled_empty_flag.data_ptr<int>(), cur_stream(xyz)));
    return 1;
}

int pts_in_boxes3d_cpu(at::Tensor pts_flag, at::Tensor pts, at::Tensor boxes3d)
{
    TORCH_CHECK(!pts_flag.is_cuda() && !pts.is_cuda() && !boxes3d.is_cuda(), "pts_in_boxes3d_cpu takes CPU tensors");
    TORCH_CHECK(pts_flag.is_contiguous() && pts.is_contiguous() && boxes3d.is_contiguous(), "contiguous tensors required");
    TORCH_CHECK(pts_flag.scalar_type() == at::kLong, "pts_flag must be int64"); PRCNN_CHECK_F32(pts); PRCNN_CHECK_F32(boxes3d);
    prcnn_ok(prcnn_host_pts_in_boxes3d((int)boxes3d.size(0), (int)pts.size(0), pts.data_ptr<float>(), boxes3d.data_ptr<float>(),
                                       (long long *)pts_flag.data_ptr<int64_t>()));
    return 1;
}

int roipool3d_cpu(at::Tensor pts, at::Tensor boxes3d, at::Tensor pts_feature, at::Tensor pooled_pts, at::Tensor pooled_features,
                  at::Tensor pooled_empty_flag)
{
    for (const at::Tensor &t : {pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag})
        TORCH_CHECK(!t.is_cuda() && t.is_contiguous(), "roipool3d_cpu takes contiguous CPU tensors");
    PRCNN_CHECK_F32(pts); PRCNN_CHECK_F32(boxes3d); PRCNN_CHECK_F32(pts_feature); PRCNN_CHECK_F32(pooled_pts); PRCNN_CHECK_F32(pooled_features);
    TORCH_CHECK(pooled_empty_flag.scalar_type() == at::kLong, "pooled_empty_flag must be int64");
    prcnn_ok(prcnn_host_roipool3d((int)boxes3d.size(0), (int)pts.size(0), (int)pts_feature.size(1), (int)pooled_pts.size(1), pts.data_ptr<float>(),
                                  boxes3d.data_ptr<float>(), pts_feature.data_ptr<float>(), pooled_pts.data_ptr<float>(),
                                  pooled_features.data_ptr<float>(), (long long *)pooled_empty_flag.data_ptr<int64_t>()));
    return 1;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("pts_in_boxes3d_cpu", &pts_in_boxes3d_cpu, "pts_in_boxes3d_cpu");
    m.def("roipool3d_cpu", &roipool3d_cpu, "roipool3d_cpu");
    m.def("forward", &roipool3d_gpu, "roipool3d forward (CUDA)");
    m.def("forward_slow", &roipool3d_gpu, "roipool3d forward (CUDA)");
}
