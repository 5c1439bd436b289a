/*
 * prcnn_oracle.h -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * Scalar C restatement of the reference's device kernels on the PointRCNN
 * eval_rcnn hot path (reference = cxy1997/3D_adapt_auto_driving, paths below are
 * relative to /root/reference/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped package never does.
 *
 * Arithmetic contract:
 *   - IEEE binary32, one rounding per source-level operation, NO fma contraction
 *     (built with -ffp-contract=off), sums evaluated left to right as written in
 *     the reference source.
 *   - sin/cos/atan2 on f32 arguments := (float) libm_double(f)((double) x).
 *   - comparisons against double literals are done in double, as the reference does.
 *
 * Pinning status: the reference ships no tests/golden vectors for any of these ops
 * (SURVEY.md section 4).  roipool3d / pts_in_boxes3d are pinned against the reference's
 * own CPU code compiled from /root/reference (oracle/_ref, see oracle/Makefile);
 * every other function is pinned by an independent numpy brute force in
 * tests/ and by tests/golden fixtures => "parity unpinned by reference tests"
 * for those (stated in DESIGN.md).
 */
#ifndef PRCNN_ORACLE_H
#define PRCNN_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* 0 = the contract-off oracle (liboracle.so); 1 / 2 = the FMA-contracted second oracles (liboracle_fma.so:
 * the whole file under -ffp-contract=fast -mfma, the analogue of the reference's `nvcc -O2` build,
 * pointnet2_lib/pointnet2/setup.py:19-20; liboracle_fma2.so: the other association of the squared distance) */
int orc_variant(void);

/* cuda_utils.h:10-13 -- float-log power of two, capped at 1024 */
int orc_opt_n_threads(int work_size);

/* K1 ball_query_gpu.cu:9-45 */
void orc_ball_query(int b, int n, int m, float radius, int nsample,
                    const float *new_xyz, const float *xyz, int *idx);
/* K2 group_points_gpu.cu:47-66 */
void orc_group_points(int b, int c, int n, int npoints, int nsample,
                      const float *points, const int *idx, float *out);
/* K3 group_points_gpu.cu:8-25 (scatter-add; serial order => deterministic) */
void orc_group_points_grad(int b, int c, int n, int npoints, int nsample,
                           const float *grad_out, const int *idx, float *grad_points);
/* K4 sampling_gpu.cu:8-24 */
void orc_gather_points(int b, int c, int n, int npoints,
                       const float *points, const int *idx, float *out);
/* K5 sampling_gpu.cu:46-63 */
void orc_gather_points_grad(int b, int c, int n, int npoints,
                            const float *grad_out, const int *idx, float *grad_points);
/* K6 with the distance arithmetic of the reference's hipcc-built kernel binary: (fma(dy, dy, dx*dx)) + dz*dz  (mlp_oracle.c) */
void orc_furthest_point_sampling_hipcc_bs(int b, int n, int m, int bs, const float *xyz, float *temp, int *idx);
/* K6 sampling_gpu.cu:86-209 with block size = orc_opt_n_threads(n) (dispatch :211-253) */
void orc_furthest_point_sampling(int b, int n, int m,
                                 const float *xyz, float *temp, int *idx);
/* same, with the virtual block size given explicitly (tests of the tie rule) */
void orc_furthest_point_sampling_bs(int b, int n, int m, int block_size,
                                    const float *xyz, float *temp, int *idx);
/* K7 interpolate_gpu.cu:9-52 */
void orc_three_nn(int b, int n, int m, const float *unknown, const float *known,
                  float *dist2, int *idx);
/* K8 interpolate_gpu.cu:77-97 */
void orc_three_interpolate(int b, int c, int m, int n, const float *points,
                           const int *idx, const float *weight, float *out);
/* K9 interpolate_gpu.cu:120-142 */
void orc_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                const int *idx, const float *weight, float *grad_points);

/* QueryAndGroup.forward pointnet2_utils.py:241-264 on top of K1+K2:
 * out (B, 3+C, M, ns) = cat(group(xyz^T)-new_xyz, group(features)); idx zero-filled first. */
void orc_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                         const float *new_xyz, const float *xyz, const float *features,
                         int *idx, float *out);

/* K10 iou3d_kernel.cu:223-234 (box_overlap :108-212) */
void orc_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                           float *ans_overlap);
/* K11 iou3d_kernel.cu:236-248 (iou_bev :214-221) */
void orc_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                       float *ans_iou);
/* K12 iou3d_kernel.cu:250-292 + host reduce iou3d.cpp:100-119; returns num_to_keep */
int orc_nms(int boxes_num, const float *boxes, long long *keep, float thresh);
/* K13 iou3d_kernel.cu:306-348 (iou_normal :295-303) + host reduce iou3d.cpp:150-169 */
int orc_nms_normal(int boxes_num, const float *boxes, long long *keep, float thresh);

/* K14-K16 roipool3d_kernel.cu:14-28,97-194,209-237 (batched GPU semantics) */
void orc_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                   int sampled_pts_num, const float *xyz, const float *boxes3d,
                   const float *pts_feature, float *pooled_features, int *pooled_empty_flag);
/* roipool3d.cpp:97-125 */
void orc_pts_in_boxes3d(int boxes_num, int pts_num, const float *pts, const float *boxes3d,
                        long long *pts_flag);
/* roipool3d.cpp:127-195 */
void orc_roipool3d_cpu(int boxes_num, int pts_num, int feature_len, int sampled_pts_num,
                       const float *pts, const float *boxes3d, const float *pts_feature,
                       float *pooled_pts, float *pooled_features, long long *pooled_empty_flag);

/* K18 evaluate/rotate_iou.py:261-291 (device fn :248-259), boxes = centre format */
void orc_rotate_iou_eval(int n, int k, const float *boxes, const float *query_boxes,
                         float *iou, int criterion);

/* number of OpenMP threads the library will use (1 if built without -fopenmp) */
int orc_num_threads(void);
void orc_set_num_threads(int t);

/* ---- mlp_oracle.c: the shared-MLP kernels this build owns, restated in THEIR fixed summation order (bit-exact checks).
 * csrc/sa_mlp_fused.hip + csrc/sa_packed.hip */
void orc_set_mfma_korder(int k);
void orc_rows_layer_mfma16(long rows, int c0, int n, const float *A, long lda, const float *W, int ldw, const float *bias, int do_relu,
                           float *out, long ldo);
void orc_rows_layer_mfma(long rows, int K, int n, const float *A, long lda, const float *W, const float *bias, int do_relu,
                         float *out, long ldo);
void orc_sa_mlp_fused(int b, int n, int m, int ns, int c3, const float *new_xyz, const float *xyz, const float *P,
                      const float *wxyz, const int *idx, const float *w2t, const float *b2, const float *w3t,
                      const float *b3, float *out, int out_stride, int out_col);
/* csrc/packed_layer.hip (layer 1 over grouped rows, fmaf chain) */
void orc_gather_affine_fma(int b, int n, int m, int ns, int c1, const float *new_xyz, const float *xyz, const float *P,
                           const float *wxyz, const int *idx, float *out);
/* csrc/packed_layer.hip rows_dot_kernel */
void orc_rows_dot(long rows, int K, int n, const float *A, long lda, const float *W, const float *bias, float *out, long ldo);
/* csrc/sa_xyz_mlp.hip */
void orc_sa_xyz_mlp(int b, int n, int m, int ns, int c1, int c2, int c3, const float *new_xyz, const float *xyz,
                    const int *idx, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                    const float *b3, float *out, int out_stride, int out_col);
/* csrc/rcnn_point_mlp.hip */
void orc_rcnn_point_mlp(long r, int ld, int fcol, const float *rows, const float *wu1, const float *bu1, const float *wu2,
                        const float *bu2, const float *wm, const float *bm, const float *wp, const float *bp,
                        float *xfeat, float *merged, float *p);

#ifdef __cplusplus
}
#endif
#endif
