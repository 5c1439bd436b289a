// binding_common.h -- shared by the three compiled extension modules that carry the reference's names
// (pointnet2_cuda, iou3d_cuda, roipool3d_cuda): thin pybind11 wrappers around the C ABI of libprcnn_hip.so with the
// reference bindings' signatures (Tensor arguments, caller-allocated outputs, redundant int dimensions).  Host code only --
// the kernels live in libprcnn_hip.so (include/prcnn_hip.h).
#pragma once
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include "../../../include/prcnn_hip.h"

#define PRCNN_CHECK_DEV(x) TORCH_CHECK((x).is_cuda() && (x).is_contiguous(), #x " must be a contiguous device tensor")
#define PRCNN_CHECK_F32(x) TORCH_CHECK((x).scalar_type() == at::kFloat, #x " must be float32")
#define PRCNN_CHECK_I32(x) TORCH_CHECK((x).scalar_type() == at::kInt, #x " must be int32")

static inline void *cur_stream(const at::Tensor &t)
{
    return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

// the reference exits the process on a failed launch (pointnet2) or ignores it (roipool3d); here every failure raises
static inline void prcnn_ok(int rc)
{
    TORCH_CHECK(rc == PRCNN_OK, "libprcnn_hip: ", prcnn_last_error());
}
