// pointnet2_cuda as a compiled extension module: the nine entry points of the reference's pointnet2_api.cpp:10-24 with the
// same names, argument order and return types, forwarding to libprcnn_hip.so.  Outputs are allocated by the caller
// (pointnet2_utils.py:25,55,94-95,128,172,218); kernels go to torch's current stream like the reference's
// THCState_getCurrentStream launches (ball_query.cpp:22).
#include "binding_common.h"

// ball_query.cpp:14-25 (the only pointnet2 wrapper that checks its inputs: :10-12,16-17)
int ball_query_wrapper_fast(int b, int n, int m, float radius, int nsample, at::Tensor new_xyz, at::Tensor xyz, at::Tensor idx)
{
    PRCNN_CHECK_DEV(new_xyz); PRCNN_CHECK_DEV(xyz); PRCNN_CHECK_F32(new_xyz); PRCNN_CHECK_F32(xyz); PRCNN_CHECK_I32(idx);
    prcnn_ok(prcnn_ball_query(b, n, m, radius, nsample, new_xyz.data_ptr<float>(), xyz.data_ptr<float>(), idx.data_ptr<int>(),
                              cur_stream(xyz)));
    return 1;
}

// group_points.cpp:25-36
int group_points_wrapper_fast(int b, int c, int n, int npoints, int nsample, at::Tensor points, at::Tensor idx, at::Tensor out)
{
    PRCNN_CHECK_F32(points); PRCNN_CHECK_I32(idx); PRCNN_CHECK_F32(out);
    prcnn_ok(prcnn_group_points(b, c, n, npoints, nsample, points.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(),
                                cur_stream(points)));
    return 1;
}

// group_points.cpp:11-22
int group_points_grad_wrapper_fast(int b, int c, int n, int npoints, int nsample, at::Tensor grad_out, at::Tensor idx, at::Tensor grad_points)
{
    PRCNN_CHECK_F32(grad_out); PRCNN_CHECK_I32(idx); PRCNN_CHECK_F32(grad_points);
    prcnn_ok(prcnn_group_points_grad(b, c, n, npoints, nsample, grad_out.data_ptr<float>(), idx.data_ptr<int>(),
                                     grad_points.data_ptr<float>(), cur_stream(grad_out)));
    return 1;
}

// sampling.cpp:11-20
int gather_points_wrapper_fast(int b, int c, int n, int npoints, at::Tensor points, at::Tensor idx, at::Tensor out)
{
    PRCNN_CHECK_F32(points); PRCNN_CHECK_I32(idx); PRCNN_CHECK_F32(out);
    prcnn_ok(prcnn_gather_points(b, c, n, npoints, points.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(), cur_stream(points)));
    return 1;
}

// sampling.cpp:23-33
int gather_points_grad_wrapper_fast(int b, int c, int n, int npoints, at::Tensor grad_out, at::Tensor idx, at::Tensor grad_points)
{
    PRCNN_CHECK_F32(grad_out); PRCNN_CHECK_I32(idx); PRCNN_CHECK_F32(grad_points);
    prcnn_ok(prcnn_gather_points_grad(b, c, n, npoints, grad_out.data_ptr<float>(), idx.data_ptr<int>(), grad_points.data_ptr<float>(),
                                      cur_stream(grad_out)));
    return 1;
}

// sampling.cpp:36-46
int furthest_point_sampling_wrapper(int b, int n, int m, at::Tensor points, at::Tensor temp, at::Tensor idx)
{
    PRCNN_CHECK_F32(points); PRCNN_CHECK_F32(temp); PRCNN_CHECK_I32(idx);
    prcnn_ok(prcnn_furthest_point_sampling(b, n, m, points.data_ptr<float>(), temp.data_ptr<float>(), idx.data_ptr<int>(), cur_stream(points)));
    return 1;
}

// interpolate.cpp:14-23
void three_nn_wrapper_fast(int b, int n, int m, at::Tensor unknown, at::Tensor known, at::Tensor dist2, at::Tensor idx)
{
    PRCNN_CHECK_F32(unknown); PRCNN_CHECK_F32(known); PRCNN_CHECK_F32(dist2); PRCNN_CHECK_I32(idx);
    prcnn_ok(prcnn_three_nn(b, n, m, unknown.data_ptr<float>(), known.data_ptr<float>(), dist2.data_ptr<float>(), idx.data_ptr<int>(),
                            cur_stream(unknown)));
}

// interpolate.cpp:26-39
void three_interpolate_wrapper_fast(int b, int c, int m, int n, at::Tensor points, at::Tensor idx, at::Tensor weight, at::Tensor out)
{
    PRCNN_CHECK_F32(points); PRCNN_CHECK_I32(idx); PRCNN_CHECK_F32(weight); PRCNN_CHECK_F32(out);
    prcnn_ok(prcnn_three_interpolate(b, c, m, n, points.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                     out.data_ptr<float>(), cur_stream(points)));
}

// interpolate.cpp:42-54
void three_interpolate_grad_wrapper_fast(int b, int c, int n, int m, at::Tensor grad_out, at::Tensor idx, at::Tensor weight, at::Tensor grad_points)
{
    PRCNN_CHECK_F32(grad_out); PRCNN_CHECK_I32(idx); PRCNN_CHECK_F32(weight); PRCNN_CHECK_F32(grad_points);
    prcnn_ok(prcnn_three_interpolate_grad(b, c, n, m, grad_out.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                          grad_points.data_ptr<float>(), cur_stream(grad_out)));
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("ball_query_wrapper", &ball_query_wrapper_fast, "ball_query_wrapper_fast");
    m.def("group_points_wrapper", &group_points_wrapper_fast, "group_points_wrapper_fast");
    m.def("group_points_grad_wrapper", &group_points_grad_wrapper_fast, "group_points_grad_wrapper_fast");
    m.def("gather_points_wrapper", &gather_points_wrapper_fast, "gather_points_wrapper_fast");
    m.def("gather_points_grad_wrapper", &gather_points_grad_wrapper_fast, "gather_points_grad_wrapper_fast");
    m.def("furthest_point_sampling_wrapper", &furthest_point_sampling_wrapper, "furthest_point_sampling_wrapper");
    m.def("three_nn_wrapper", &three_nn_wrapper_fast, "three_nn_wrapper_fast");
    m.def("three_interpolate_wrapper", &three_interpolate_wrapper_fast, "three_interpolate_wrapper_fast");
    m.def("three_interpolate_grad_wrapper", &three_interpolate_grad_wrapper_fast, "three_interpolate_grad_wrapper_fast");
}
