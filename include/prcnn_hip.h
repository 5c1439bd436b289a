/*
 * prcnn_hip.h -- C ABI of libprcnn_hip.so: the MI355X (gfx950) implementation of the
 * PointRCNN eval_rcnn hot-path operators of cxy1997/3D_adapt_auto_driving.
 *
 * This is the drop-in boundary.  Every entry point is what the reference's three pybind
 * modules bind for this path (reference paths relative to /root/reference/pointrcnn/):
 *
 *   pointnet2_cuda  pointnet2_lib/pointnet2/src/pointnet2_api.cpp:10-24
 *   iou3d_cuda      lib/utils/iou3d/src/iou3d.cpp:174-179
 *   roipool3d_cuda  lib/utils/roipool3d/src/roipool3d.cpp:198-203
 *   rotate_iou      ../evaluate/rotate_iou.py:294-329 (numba.cuda kernel :261-291)
 *
 * Conventions (same as the reference wrappers):
 *   - plain device pointers + sizes; f32 / i32, C-contiguous, row-major; the CALLER allocates
 *     every output; dims are passed as ints.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  The
 *     reference launches pointnet2 on torch's current stream and iou3d/roipool3d on the default
 *     stream; here every op takes the stream explicitly and is asynchronous unless stated.
 *   - return value: 0 on success, a negative PRCNN_E* code on failure (the reference calls
 *     exit(-1) on a failed launch, ball_query_gpu.cu:63-66; we report instead).
 *     prcnn_last_error() returns a static description of the last failure on this thread.
 *   - no allocation inside any call except the blocking prcnn_nms / prcnn_nms_normal, which
 *     keep one cached device scratch (the reference cudaMalloc/cudaFree's per call,
 *     iou3d.cpp:87,97).
 */
#ifndef PRCNN_HIP_H
#define PRCNN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PRCNN_OK 0
#define PRCNN_EINVAL (-1)  /* bad argument (null pointer, negative size, unsupported size) */
#define PRCNN_ELAUNCH (-2) /* HIP launch / runtime error */

int prcnn_version(void);
const char *prcnn_last_error(void);
/* opt_n_threads() of cuda_utils.h:10-13: the reference FPS block size for n points. */
int prcnn_opt_n_threads(int work_size);

/* ---- pointnet2_cuda ------------------------------------------------------------------ */

/* ball_query_wrapper_fast  src/ball_query.cpp:14-25 -> kernel src/ball_query_gpu.cu:9-45.
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample): first nsample in-radius indices in index
 * order, first hit back-fills; rows of empty balls are left untouched (caller zero-fills). */
int prcnn_ball_query(int b, int n, int m, float radius, int nsample,
                     const float *new_xyz, const float *xyz, int *idx, void *stream);

/* prcnn_ball_query with EVERY slot of idx written: the row of an empty ball holds the zeros the reference's caller fills the tensor
 * with first (pointnet2_utils.py:218) -- the engine's form (no fill launch in front of each query). */
int prcnn_ball_query_full(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                          void *stream);

/* Algorithm selector for ball query (results are identical): 0 = automatic (bucket-sorted hashed grid with a
 * wave per centre for n >= 2048 points -- csrc/ball_dense.hip --, brute-force scan otherwise), 1 = brute-force
 * scan only, 2 = round 1's linked-list hashed grid (n >= 4096) in place of the bucket-sorted one.  For tests /
 * profiling. */
int prcnn_set_ball_query_mode(int mode);

/* group_points_wrapper_fast  src/group_points.cpp:25-36 -> src/group_points_gpu.cu:47-66.
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample). */
int prcnn_group_points(int b, int c, int n, int npoints, int nsample,
                       const float *points, const int *idx, float *out, void *stream);
/* group_points_grad_wrapper_fast  src/group_points.cpp:11-22 -> src/group_points_gpu.cu:8-25.
 * grad_points (b,c,n) must be zero-filled by the caller; atomic scatter-add. */
int prcnn_group_points_grad(int b, int c, int n, int npoints, int nsample,
                            const float *grad_out, const int *idx, float *grad_points, void *stream);

/* Ball query over clouds whose points k >= limit[cloud] are copies of point k % limit[cloud] (the pooled rows of a RoI that
 * holds fewer than 512 points, roipool3d_kernel.cu:152-159): only the first limit[cloud] points are scanned.  Same distinct
 * points per ball as prcnn_ball_query, slots past them repeat the first hit; every slot of idx is written (an empty ball gets
 * zeros).  Engine-side shortcut, not reference ABI. */
int prcnn_ball_query_limit(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                           const int *limit, int *idx, void *stream);

/* gather_points_wrapper_fast  src/sampling.cpp:11-20 -> src/sampling_gpu.cu:8-24. */
int prcnn_gather_points(int b, int c, int n, int npoints,
                        const float *points, const int *idx, float *out, void *stream);
/* gather_points_grad_wrapper_fast  src/sampling.cpp:23-33 -> src/sampling_gpu.cu:46-63. */
int prcnn_gather_points_grad(int b, int c, int n, int npoints,
                             const float *grad_out, const int *idx, float *grad_points, void *stream);

/* furthest_point_sampling_wrapper  src/sampling.cpp:36-46 -> src/sampling_gpu.cu:93-253.
 * xyz (b,n,3), temp (b,n) scratch pre-filled by the caller (1e10) -> idx (b,m).  Ties resolve
 * as in the reference kernel launched with opt_n_threads(n) threads.  On return temp holds the
 * running minimum distances, as the reference leaves them. */
int prcnn_furthest_point_sampling(int b, int n, int m,
                                  const float *xyz, float *temp, int *idx, void *stream);

/* Which ARITHMETIC the squared distance of sampling_gpu.cu:133 is evaluated in, for every FPS entry point of the library
 * (process-wide, read at launch).  0 (default): the source's -- (dx*dx + dy*dy) + dz*dz, one rounding per operation: the parity
 * contract of this build, independent of any compiler's contraction choices.  1: the reference's KERNEL BINARY as hipcc builds
 * that file for gfx950 -- (fma(dy, dy, dx*dx)) + dz*dz, read off its disassembly: a user who migrates from the hipcc-compiled
 * reference and wants its picks bit for bit (near-ties included) selects this one.  (What nvcc made of the expression on the
 * reference's original platform is not observable here; oracle/fma_table.py counts how rarely any contraction moves a pick.) */
int prcnn_set_fps_arithmetic(int mode);

/* FPS with the selected points' coordinates written beside their indices: the result of furthest_point_sample + gather_operation
 * (pointnet2_modules.py:40-46) in one launch (no scratch fill, index cast or gather by the caller).  Served shapes: n <= 1024 (many small
 * clouds, a wave or four per cloud) and, since round 4, 2048 < n <= 16384 with m >= 256 (the speculative kernel; round 5: 16384 < n <= 32768 on two workgroups per cloud, and every other shape through an internal distance scratch + gather).  Same selection, same
 * tie rule as prcnn_furthest_point_sampling. */
int prcnn_fps_new_xyz(int b, int n, int m, const float *xyz, int *idx, float *new_xyz, void *stream);

/* three_nn_wrapper_fast  src/interpolate.cpp:14-23 -> src/interpolate_gpu.cu:9-52.
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) SQUARED distances, idx (b,n,3). */
int prcnn_three_nn(int b, int n, int m, const float *unknown, const float *known,
                   float *dist2, int *idx, void *stream);

/* three_nn followed by the weights of PointnetFPModule.forward (pointnet2_lib/pointnet2/pointnet2_modules.py:139-144;
 * pointnet2_utils.py:97 takes the square root) in the kernel's epilogue: idx (b,n,3) i32, weight (b,n,3) f32 with
 * r_k = 1 / (sqrt(dist2_k) + 1e-8), weight_k = r_k / ((r_0 + r_1) + r_2), one rounding per operation.  The engine's form: the
 * reference's Python runs this as sqrt, add, reciprocal, sum, divide -- five launches per FP level. */
int prcnn_three_nn_weights(int b, int n, int m, const float *unknown, const float *known, int *idx, float *weight,
                           void *stream);
/* three_interpolate_wrapper_fast  src/interpolate.cpp:26-39 -> src/interpolate_gpu.cu:77-97.
 * points (b,c,m), idx/weight (b,n,3) -> out (b,c,n). */
int prcnn_three_interpolate(int b, int c, int m, int n, const float *points,
                            const int *idx, const float *weight, float *out, void *stream);
/* three_interpolate_grad_wrapper_fast  src/interpolate.cpp:42-54 -> interpolate_gpu.cu:120-142. */
int prcnn_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                 const int *idx, const float *weight, float *grad_points, void *stream);

/* Fused QueryAndGroup.forward (pointnet2_utils.py:241-264 = K1 + K2 x2 + centre subtract + cat):
 * out (b,3+c,m,nsample) = cat(xyz[idx]-new_xyz, features[idx]); features may be NULL (c = 0).
 * idx (b,m,nsample) is an OUTPUT here and is fully written (empty balls -> 0, the value the
 * reference's zero-filled idx holds).  Not part of the reference ABI; it is what the
 * reference's Python composes from it. */
int prcnn_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                          const float *new_xyz, const float *xyz, const float *features,
                          int *idx, float *out, void *stream);

/* ---- shared-MLP epilogues (fused forms of pytorch_utils.py Conv2d -> BN(eval) -> ReLU and of
 *      the max over nsample of pointnet2_modules.py:41-44; not in the reference ABI) ------------ */

/* x (outer, c, inner) f32 in place: x = max(x + bias[c], 0).  Replaces the separate bias-add and
 * ReLU passes after a bias-free 1x1 convolution (pytorch_utils.py:35-101). */
int prcnn_bias_relu_inplace(long outer, int c, long inner, const float *bias, float *x, void *stream);

/* in (b, c, npoint, nsample) = raw output of the LAST 1x1 convolution of a shared MLP ->
 * out (b, c, npoint) = relu(max_s in + bias[c]), which equals max_s relu(in + bias[c]) exactly
 * (F.max_pool2d over nsample, pointnet2_modules.py:41-44). */
int prcnn_maxpool_bias_relu(int b, int c, int npoint, int nsample, const float *bias,
                            const float *in, float *out, void *stream);

/* ---- point-major (channels-last) forms used by the MI355X inference path (csrc/pointmajor.hip);
 *      same values as K2 / the max-pool / K8, features stored (b, n, c) instead of (b, c, n) -------- */

/* Grouping for QueryAndGroup (pointnet2_utils.py:241-264) with point-major features (b,n,c):
 * out (b, m*nsample, kpad), kpad = round_up(c,4)+4, row = [features[idx] | 0-pad | xyz[idx]-centre | 0]. */
int prcnn_group_cat_pm(int b, int n, int m, int c, int nsample, const float *new_xyz, const float *xyz,
                       const float *features, const int *idx, float *out, void *stream);
/* First shared-MLP layer on grouped points without materialising the grouped input (the layer is
 * linear before its ReLU): P (b,n,cout) = features @ W1f^T + b1 per point, wxyz (3,cout) = the xyz
 * columns of W1 -> out (b, m*nsample, cout) = relu(P[idx] + wxyz . (xyz[idx] - new_xyz)). */
int prcnn_gather_affine_relu_pm(int b, int n, int m, int cout, int nsample, const float *new_xyz,
                                const float *xyz, const float *P, const float *wxyz, const int *idx,
                                float *out, void *stream);
/* One kernel for a set-abstraction MLP on grouped points (hand-written v_mfma_f32_32x32x2_f32 tiles):
 * gather -> layer 1 (P[idx] + wxyz.(xyz[idx]-centre), ReLU) -> layer 2 (w2t,b2,ReLU) -> layer 3
 * (w3t,b3,ReLU) -> max over nsample; nothing of the grouped activations touches HBM
 * (pointnet2_modules.py:37-53 for one scale).  w2t (c1,c2) / w3t (c2,c3) are stored input-channel-major.
 * Supported: c1 = c2 = 128, c3 in {128,256}, nsample = 64.  out[(b*m)][out_col..out_col+c3), row stride
 * out_stride. */
int prcnn_sa_mlp_fused(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *new_xyz,
                       const float *xyz, const float *P, const float *wxyz, const int *idx, const float *w2t,
                       const float *b2, const float *w3t, const float *b3, float *out, int out_stride,
                       int out_col, void *stream);

/* The same fused set-abstraction MLP over the DISTINCT grouped rows only.  The reference's ball query back-fills the
 * slots beyond the hit count with the first hit (ball_query_gpu.cu:35-39) and the grouped MLP + max_pool2d
 * (pointnet2_utils.py:241-264, pointnet2_modules.py:37-53) evaluates those copies again; the max over a group does not
 * change when they are dropped.  prcnn_ball_pack: idx (b,m,nsample) -> per cloud, cnt[c] = 1 + (last slot that differs
 * from slot 0) rows per centre, written as a dense list of 64-row tiles:
 *   rowinfo  u32 [b * ceil(m*nsample/64) * 64]: (centre within cloud) << 16 | (point within cloud)
 *   rowdxyz  float4 [same length]: xyz[point] - new_xyz[centre] of the row (the grouped, centre-relative coordinates of
 *            pointnet2_utils.py:252), so that the MLP kernels do not gather coordinates again
 *   tilecloud i32 [b * ceil(m*nsample/64)]:     cloud of each tile
 *   hdr      u32 [4]: [0] tiles, [1] distinct rows (device-resident; no host sync)
 *   limit    i32 [b], optional: points k >= limit[cloud] of a cloud are copies of point k % limit[cloud] (the wrap-around
 *            fill of RoI pooling, roipool3d_kernel.cu:152-159); their rows are duplicates as well and are dropped
 * prcnn_sa_packed_mlp: layers as prcnn_sa_mlp_fused (c1 = c2 = 128; narrower levels zero-padded by the caller), c3 in
 * {128, 256}; zeroes out[(b*m)][out_col..out_col+c3) and accumulates the per-centre maxima with atomicMax (values are
 * >= 0 after ReLU).  Bit-identical to prcnn_sa_mlp_fused on the same idx.  max_tiles = b * ceil(m*nsample/64). */
int prcnn_ball_pack(int b, int n, int m, int nsample, const int *idx, const int *limit, const float *xyz,
                    const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud, unsigned int *hdr,
                    void *stream);
/* The same for b = lists x group clouds in one launch: list l = clouds [l group, (l+1) group) gets its own row list and header --
 * rowinfo / rowdxyz [lists][group * tiles_cap * 64], tilecloud [lists][group * tiles_cap] (cloud index INSIDE the list),
 * hdr [lists][4].  Used for the packed row lists of the batches of one geometry group (each batch's kernels walk their own tiles). */
int prcnn_ball_pack_groups(int b, int group, int n, int m, int nsample, const int *idx, const int *limit, const float *xyz,
                           const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud, unsigned int *hdr,
                           void *stream);

/* Every form of prcnn_ball_pack behind one entry (round 5): group = clouds per list (b: one list), limit / rep / crep optional (NULL),
 * hdr_is_zero != 0: hdr [lists][4] already holds zeros -- a slice of an arena the caller zeroes once per chain of launches instead of
 * one memset per row list. */
int prcnn_ball_pack_ex(int b, int group, int n, int m, int nsample, const int *idx, const int *limit, const int *rep, const int *crep,
                       const float *xyz, const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud, unsigned int *hdr,
                       int hdr_is_zero, void *stream);

/* prcnn_ball_pack with a representative map (round 3): rep (b,n) i32, rep[cloud][k] = the lowest-indexed point of the cloud
 * that is an exact copy of point k (coordinates and features; k when it is the first of its kind).  A copy lies in a ball
 * iff its representative does and the representative is listed earlier in the same row, so the slots whose point is not
 * its own representative are dropped from the row list as well (a max-pool over copies is the max-pool over the
 * originals: same bits as pointnet2_modules.py:37-53 over all nsample rows).  nsample <= 64 with rep.
 * crep (b,m) i32, optional: the same kind of map over the CENTRES (new_xyz): a centre that is an exact copy of an earlier
 * one gets no rows at all -- its pooled output is never computed (the caller's zero stays) and must not be read; the level
 * above passes this map as its `rep`, so it never is.  rep and crep may each be null. */
int prcnn_ball_pack_rep(int b, int n, int m, int nsample, const int *idx, const int *limit, const int *rep, const int *crep,
                        const float *xyz, const float *new_xyz, unsigned int *rowinfo, float *rowdxyz, int *tilecloud,
                        unsigned int *hdr, void *stream);

/* The representative map of the points an FPS call sampled: sel (b,m) i32 indexes clouds of n points of which the points
 * k >= limit[cloud] are copies of point k % limit[cloud] (RoI pooling's wrap-around fill, roipool3d_kernel.cu:152-159)
 * and / or prev (b,n) i32 is the representative map of those n points; rep (b,m) i32 <- the first sampled point with the
 * same source as sampled point j. */
int prcnn_dup_rep(int b, int n, int m, const int *sel, const int *limit, const int *prev, int *rep, void *stream);

/* The whole geometry chain of the RCNN's RoI clouds in ONE launch, a wave per RoI (csrc/fps.hip, round 3): xyz (b,512,3) pooled
 * coordinates whose points k >= limit[cloud] are copies of point k % limit[cloud] ->
 *   new_xyz1 (b,128,3), idx1 (b,128,ns1), rep1 (b,128)  = prcnn_fps_new_xyz(128), prcnn_ball_query_limit(r1, ns1), prcnn_dup_rep(limit)
 *   new_xyz2 (b,32,3),  idx2 (b,32,ns2),  rep2 (b,32)   = the same one level up over the 128 centres: prcnn_fps_new_xyz(32),
 *                                                         prcnn_ball_query(r2, ns2) into a zero-filled tensor, prcnn_dup_rep(prev = rep1)
 * -- rcnn_net.py:165-175 (SA modules 1 and 2 of the RCNN: pointnet2_modules.py:37-46 sampling + grouping indices) under
 * default.yaml's RCNN.NUM_POINTS 512, SA_CONFIG.NPOINTS [128, 32, -1].  n == 512, m1 == 128, m2 == 32, ns1, ns2 <= 64. */
int prcnn_rcnn_roi_geometry(int b, int n, int m1, float r1, int ns1, int m2, float r2, int ns2, const float *xyz,
                            const int *limit, float *new_xyz1, int *idx1, int *rep1, float *new_xyz2, int *idx2, int *rep2,
                            void *stream);
/* ... and the distinct-row lists of both levels out of the same launch (round 5; csrc/fps.hip roi_pack_out): what
 *   prcnn_ball_pack_ex(b, b, 512, 128, ns1, idx1, limit, NULL, rep1, xyz, new_xyz1, rowinfo1, rowdxyz1, tilecloud1, hdr1, ...) and
 *   prcnn_ball_pack_ex(b, b, 128, 32, ns2, idx2, NULL, rep1, rep2, new_xyz1, new_xyz2, rowinfo2, rowdxyz2, tilecloud2, hdr2, ...)
 * write -- every cloud's rows in the same order, cut into the same tiles; the order of the clouds' tiles inside a list is whatever
 * the list's counter hands out (as for prcnn_ball_pack).  Buffers sized as for prcnn_ball_pack: b * ceil(m * ns / 64) tiles per list;
 * hdr1 / hdr2 (4 u32 each) are zeroed by this call unless hdr_is_zero != 0 (the caller zeroed them: see prcnn_ball_pack_ex).
 * idx1 and idx2 may both be NULL: the index tensors are then not written (the packed MLP kernels read the row lists only).
 * tilecloud1 and tilecloud2 may both be NULL: the lists are then written in the form whose ROWS carry their cloud -- descriptor
 * (cloud << 16) | (centre << 9) | point, hdr[1] = rows, hdr[0] unused -- with every cloud's rows right behind another cloud's: tiles are
 * cut wherever the rows fall, no padded last tile per cloud.  prcnn_sa_packed_mlp takes such a list when IT is given tilecloud == NULL.
 * rowinfo3 / rowdxyz3 / hdr3 (all three or none; only with the row-carried form): the list of the GroupAll level above the two sampled ones
 * (rcnn_net.py:64-92, SA_CONFIG.NPOINTS[2] = -1: one group of all m2 centres, no centre subtraction) -- per cloud a row (centre 0, point j)
 * for every level-2 centre j that is its own representative, relative coordinates = new_xyz2[j]: what
 * prcnn_ball_pack_ex(b, b, m2, 1, m2, {0 .. m2-1}, NULL, rep2, NULL, new_xyz2, zeros, ...) lists; prcnn_sa_wide_fused3 reads it (tilecloud NULL).
 * crows1 / hdr_c1 (both or none): the level-1 centres that are their own representatives, as rows cloud * m1 + centre of a (b * m1)-row
 * matrix, hdr_c1[1] of them: the list prcnn_rows_gemm128_rows takes for level 2's per-point layer (the other centres copy an earlier one:
 * no row list names them, nobody reads their rows).
 * The reference has no counterpart: it groups all nsample rows (pointnet2_utils.py:241-264); see prcnn_ball_pack. */
int prcnn_rcnn_roi_geometry_packs(int b, int n, int m1, float r1, int ns1, int m2, float r2, int ns2, const float *xyz,
                                  const int *limit, float *new_xyz1, int *idx1, int *rep1, float *new_xyz2, int *idx2, int *rep2,
                                  unsigned int *rowinfo1, float *rowdxyz1, int *tilecloud1, unsigned int *hdr1,
                                  unsigned int *rowinfo2, float *rowdxyz2, int *tilecloud2, unsigned int *hdr2,
                                  unsigned int *rowinfo3, float *rowdxyz3, unsigned int *hdr3, int *crows1, unsigned int *hdr_c1,
                                  int hdr_is_zero, void *stream);
/* out_is_zero (this entry, prcnn_sa_xyz_mlp_packed, prcnn_packed_layer_segmax): the results arrive through atomicMax into a
 * zeroed slice; 0 = the entry zeroes out[..., out_col : out_col + width) itself, 1 = the caller has zeroed it (one fill for all
 * the scales of a level instead of one strided fill per scale). */
/* tilecloud == NULL (this entry only, round 5): a list whose rows carry their cloud, as prcnn_rcnn_roi_geometry_packs writes it when it
 * is given no tilecloud either (n <= 512, m <= 128, b <= 65536): hdr[1] rows, tiles cut wherever they fall, the last tile's missing rows
 * read as copies of the list's last row.  Same per-row arithmetic: same bits as over the list with a tile per cloud. */
int prcnn_sa_packed_mlp(int b, int n, int m, int c3, long max_tiles, const float *P, const float *wxyz,
                        const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud,
                        const unsigned int *hdr, const float *w2t, const float *b2, const float *w3t,
                        const float *b3, float *out, int out_stride, int out_col, int out_is_zero, void *stream);

/* Shared-MLP layers of any width over packed row lists (csrc/packed_layer.hip): the levels whose weights do not fit the
 * register-resident fused kernels (RPN SA3/SA4: 128-196-256, 256-256-512, 256-384-512, pointrcnn/lib/config.py:58-61;
 * RCNN GroupAll level 256-256-512) run layer by layer on the distinct rows only.  Widths are zero-padded to multiples of
 * 128 by the caller; W is (K,N) input-channel-major; f32 MFMA, fixed summation order (oracle/mlp_oracle.c).
 *   prcnn_packed_gather_affine: A1 (max_tiles*64, c1) = relu(P[point] + wxyz.(xyz[point] - centre))   (layer 1)
 *   prcnn_packed_layer:         out[:, 0..n_store) = act(A @ W + bias)[:, 0..n_store) (n_store <= N: a head's narrow last layer,
 *                               weights zero-padded to N = 128) over hdr[0]*64 rows (hdr != NULL) or `rows` rows (hdr == NULL:
 *                               a plain row-major GEMM layer with a host-side row count, e.g. P = features @ W1f + b1)
 *   prcnn_packed_layer_segmax:  out[(b*m)][out_col..+N) = max over each centre's rows of relu(A @ W + bias)
 *                               (pointnet2_modules.py:37-53: last layer + max_pool2d) */
/* One scale of a wide set-abstraction level (layers 1-3 after the per-point part, max pool) over a packed row list in ONE
 * kernel (csrc/sa_wide.hip): what prcnn_packed_gather_affine + prcnn_packed_layer + prcnn_packed_layer_segmax compute, bit for
 * bit, without their activations going through HBM.  P (b,n,c1), wxyz (3,c1), w2 (c1,c2), w3 (c2,c3) k-major, widths multiples of
 * 128 as accepted by prcnn_sa_wide_fused_supported (RPN SA3 / SA4, RCNN GroupAll: pointrcnn/lib/config.py:58-61,118-120). */
int prcnn_sa_wide_fused_supported(int c1, int c2, int c3);
int prcnn_sa_wide_fused(int b, int n, int m, int c1, int c2, int c3, long max_tiles, const float *P, const float *wxyz,
                        const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud, const unsigned int *hdr,
                        const float *w2, const float *b2, const float *w3, const float *b3, float *out, int out_stride,
                        int out_col, int out_is_zero, void *stream);

/* The same scale with layer 1 inside as well (csrc/sa_wide3.hip): for a level that groups every point exactly once -- the RCNN's GroupAll
 * level, rcnn_net.py:64-92 with pointnet2_modules.py:19-55 -- the per-point part of layer 1 is the layer itself, and the separate
 * prcnn_packed_layer launch that made P (and P's trip through HBM) goes away.  F (b,n,c0) point features; wcat = w1 (c0,c1) | w2 (c1,c2) |
 * w3 (c2,c3), k-major, in ONE allocation; b1 / b2 / b3 the biases; the rest as prcnn_sa_wide_fused.  Results: prcnn_packed_layer (P = F w1 + b1)
 * followed by prcnn_sa_wide_fused, bit for bit. */
int prcnn_sa_wide_fused3_supported(int c0, int c1, int c2, int c3);
int prcnn_sa_wide_fused3(int b, int n, int m, int c0, int c1, int c2, int c3, long max_tiles, const float *F, const float *wxyz,
                         const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud, const unsigned int *hdr,
                         const float *wcat, const float *b1, const float *b2, const float *b3, float *out, int out_stride,
                         int out_col, int out_is_zero, void *stream);

/* Batched forms: up to 4 independent problems (the scales of one MSG level, pointnet2_modules.py:19-55 loops over them) in ONE
 * launch -- the sparse levels are latency-bound, side by side they cost one launch instead of one each.  The single-problem
 * entries below are these with n = 1.  segmax = 1: every problem is a level's last layer + max pool (fields b, m, rowinfo,
 * tilecloud, out_col, out_is_zero; ldo = row stride of the level's output); segmax = 0: plain layers (fields rows, n_store,
 * relu; hdr NULL = host row count). */
typedef struct prcnn_gather_problem {
    int b, n, c1; long max_tiles; const float *P; const float *wxyz; const unsigned int *rowinfo; const float *rowdxyz;
    const int *tilecloud; const unsigned int *hdr; float *out;
} prcnn_gather_problem;
typedef struct prcnn_layer_problem {
    const unsigned int *hdr; long rows; long max_tiles; int K, N, n_store; const float *A; long lda; const float *W;
    const float *bias; int relu; float *out; long ldo;
    int b, m; const unsigned int *rowinfo; const int *tilecloud; int out_col, out_is_zero;
} prcnn_layer_problem;
/* one 128-wide problem of prcnn_sa_packed_mlp (c3 = 128): the two scales of an MSG level go into ONE launch (round 5) */
typedef struct prcnn_sa_problem {
    int b, n, m, c3; long max_tiles; const float *P; const float *wxyz; const unsigned int *rowinfo; const float *rowdxyz;
    const int *tilecloud; const unsigned int *hdr; const float *w2t; const float *b2; const float *w3t; const float *b3;
    float *out; int out_stride, out_col, out_is_zero;
    int c1, c2;     /* the REAL widths of layers 1 and 2 when the 128-wide arrays are zero-padded (0 = 128): 64-64 and 64-96, the scales of
                     * RPN SA2 (tools/cfgs/default.yaml SA_CONFIG.MLPS[1]), skip the padding's MFMAs -- same bits as the padded chain */
} prcnn_sa_problem;
int prcnn_sa_packed_mlp_batch(int nprob, const prcnn_sa_problem *problems, void *stream);
int prcnn_packed_gather_affine_batch(int nprob, const prcnn_gather_problem *problems, void *stream);
int prcnn_packed_layer_batch(int nprob, const prcnn_layer_problem *problems, int segmax, void *stream);
int prcnn_packed_gather_affine(int b, int n, int c1, long max_tiles, const float *P, const float *wxyz,
                               const unsigned int *rowinfo, const float *rowdxyz, const int *tilecloud,
                               const unsigned int *hdr, float *out, void *stream);
int prcnn_packed_layer(const unsigned int *hdr, long rows, long max_tiles, int K, int N, int n_store, const float *A,
                       long lda, const float *W, const float *bias, int relu, float *out, long ldo, void *stream);
/* EXPERIMENT (round 6, numerics switch PRCNN_SPLIT_BF16, default off; csrc/split_bf16.hip): the plain per-point layer of prcnn_packed_layer
 * (hdr == NULL form) on the bf16 matrix cores, every operand split exactly into three bf16 pieces, six products per k-step of 16,
 * f32 accumulation: ~1e-7 relative to the f32 fma chain, not its bits.  prcnn_split_weights_bf16x3: W (K, N) f32 -> `out`
 * ((K / 16) * (N / 32) * 3 * 64 * 16 bytes: the B operands of v_mfma_f32_32x32x16_bf16 in issue order), once per weight matrix;
 * prcnn_rows_layer_bf16x3: out[:, 0..n_store) = act(A @ W + bias).  K % 16 == 0, N % 128 == 0, lda % 4 == 0, A 16-byte aligned.
 * Replaces nothing of the reference's (its convolutions are a GEMM library's: pytorch_utils.py:35-101); measured in profiles/r06_split_bf16.md. */
int prcnn_split_weights_bf16x3(int K, int N, const float *W, void *out, void *stream);
int prcnn_rows_layer_bf16x3(long rows, int K, int N, int n_store, const float *A, long lda, const void *wsplit, const float *bias,
                            int relu, float *out, long ldo, void *stream);
/* First layer of a feature-propagation module (pointnet2_modules.py:139-156) with the interpolation moved behind the layer's
 * linear part: out[r] = act((A[r] @ W + bias) + ((w0 G[i0] + w1 G[i1]) + w2 G[i2])), A (rows,K) = the skip features, W (K,N) = the
 * skip columns of the layer, G = coarse features @ the interpolated columns of the layer (clouds * m_known rows, N wide),
 * idx / weight (rows,3) from prcnn_three_nn, row r in cloud r / n_per_cloud.  K, N multiples of 128. */
int prcnn_packed_layer_interp(long rows, int K, int N, const float *A, long lda, const float *W, const float *bias, int relu,
                              float *out, long ldo, int n_per_cloud, int m_known, const float *G, long ldg, const int *idx,
                              const float *weight, void *stream);
/* out[r][0..n) = A[r] @ W + bias for n <= 4 outputs (the 1-wide last layer of the classification heads, rpn.py:36-50,
 * rcnn_net.py:94-103): W (K,n) k-major; 32 lanes per row, fixed summation order (oracle: orc_rows_dot). */
/* The coordinates-only first SA level (prcnn_sa_xyz_mlp) over a packed row list: distinct rows only, bit-identical. */
int prcnn_sa_xyz_mlp_packed(int b, int m, int c1, int c2, int c3, long max_tiles, const unsigned int *rowinfo,
                            const float *rowdxyz, const int *tilecloud, const unsigned int *hdr, const float *w1,
                            const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                            float *out, int out_stride, int out_col, int out_is_zero, void *stream);
int prcnn_rows_dot(long rows, int K, int n, const float *A, long lda, const float *W, const float *bias, float *out,
                   long ldo, void *stream);
/* The last stretch of the RPN over all input points in one kernel (csrc/rpn_tail.hip): three_interpolate of the coarse
 * features (pointnet2_modules.py:136-160, FP module 0, no skip features) -> SharedMLP 256-128-128 -> the backbone features,
 * and on them the cls head 128-128-1 and the reg head 128-128-n_reg (rpn.py:28-50).  known (b,m,256), idx / weight (b,n,3)
 * from three_nn; wcat (768,128) = [FP layer 1 (256 rows) | FP layer 2 | cls layer 1 | reg layer 1 | reg layer 2 zero-padded
 * beyond n_reg columns], bcat (5,128) their biases, wc2 (128) / bc2 (1) the score layer; BN folded, k-major.
 * feats (b*n,128), cls (b*n), reg (b*n,n_reg), n_reg % 4 == 0.  Same arithmetic as prcnn_three_interpolate_pm ->
 * prcnn_packed_layer x5 -> prcnn_rows_dot, bit for bit. */
int prcnn_rpn_tail(int b, int n, int m, const float *known, const int *idx, const float *weight, const float *wcat,
                   const float *bcat, const float *wc2, const float *bc2, int n_reg, float *feats, float *cls, float *reg,
                   void *stream);
/* prcnn_rpn_tail with the FP module's first layer applied at the COARSE level (round 3): relu(W1 interp(f) + b1) =
 * relu(interp(W1 f) + b1).  G (b,m,128) = f @ W1 (no bias; prcnn_packed_layer over the 4096 coarse points of a scene, a
 * quarter of the rows); wcat (512,128) = [FP layer 2 | cls layer 1 | reg layer 1 | reg layer 2 zero-padded beyond n_reg
 * columns], bcat (5,128) = the biases of FP layer 1 (added after the interpolation), FP layer 2, cls 1, reg 1, reg 2.
 * Another association of the same sums than pointnet2_modules.py:139-156 (~1e-7 relative). */
int prcnn_rpn_tail_lin(int b, int n, int m, const float *G, const int *idx, const float *weight, const float *wcat,
                       const float *bcat, const float *wc2, const float *bc2, int n_reg, float *feats, float *cls,
                       float *reg, void *stream);
/* prcnn_rpn_tail_lin with the proposal layer's decode INSIDE (round 5): boxes (b*n,7) = decode_bbox_target(xyz, reg, anchor,
 * get_xz_fine = True, get_y_by_bin = False, get_ry_fine = False) with y += h / 2 (lib/utils/bbox_transform.py:24-121,
 * lib/rpn/proposal_layer.py:23-31), operation for operation what prcnn_rpn_proposals' decode computes from the stored rows; the
 * (b*n, n_reg) regression rows themselves never reach HBM.  Served regression layout: prcnn_rpn_tail_boxes_supported (12 x / z bins,
 * 12 heading bins, fine residuals: the 76 channels of every shipped yaml); anchor_size_host: (h, w, l) in HOST memory; xyz (b,n,3). */
int prcnn_rpn_tail_boxes_supported(int channels, float loc_scope, float loc_bin_size, int num_head_bin, int xz_fine);
/* test hook of that decode's branch-free f32 fmod by 2 pi: out_mine[i] = its value, out_lib[i] = fmodf(a[i], (float)(2 pi)) */
int prcnn_selftest_fmod_two_pi(long n, const float *a, float *out_mine, float *out_lib, void *stream);
int prcnn_rpn_tail_lin_boxes(int b, int n, int m, const float *G, const int *idx, const float *weight, const float *wcat,
                             const float *bcat, const float *wc2, const float *bc2, int n_reg, float loc_scope, float loc_bin_size,
                             int num_head_bin, int xz_fine, const float *anchor_size_host, const float *xyz, float *feats,
                             float *cls, float *boxes, void *stream);
int prcnn_packed_layer_segmax(int b, int m, long max_tiles, int K, int N, const float *A, long lda, const float *W,
                              const float *bias, const unsigned int *rowinfo, const int *tilecloud,
                              const unsigned int *hdr, float *out, int out_stride, int out_col, int out_is_zero, void *stream);

/* Entrance of the RCNN as MFMA kernels (lib/net/rcnn_net.py:139-163 xyz_up_layer + concat + merge_down_layer,
 * fused with the per-point part of SA1's first layer): rows (r, ld) f32 = pooled rows
 * [x',y',z',mask,depth,0,0,0 | 128 RPN features at column fcol] as prcnn_roipool3d_canonical writes them, r % 64 == 0;
 * wu1 (8,128), wu2 (128,128), wm (256,128), wp (128,128) k-major with BN folded;
 * xfeat (r,128) = xyz_up output, merged (r,128) = relu([xfeat | feats] wm + bm), p (r,128) = merged wp + bp:
 * three launches of one tiled MFMA layer kernel, all buffers caller-allocated.
 * tilemap / ntiles (optional, both or neither): only the 64-row tiles listed in tilemap[0 .. *ntiles) are computed -- the
 * tiles that hold DISTINCT pooled rows (prcnn_pooled_tiles from prcnn_roipool3d_canonical's pooled_cnt); the rows of the
 * other tiles are wrap-around copies (roipool3d_kernel.cu:152-159) that no consumer reads, and are left unwritten. */
int prcnn_rcnn_point_mlp(long r, int ld, int fcol, const float *rows, const float *wu1, const float *bu1,
                         const float *wu2, const float *bu2, const float *wm, const float *bm, const float *wp,
                         const float *bp, float *xfeat, float *merged, float *p, const int *tilemap,
                         const unsigned int *ntiles, void *stream);
int prcnn_pooled_tiles(int clouds, int rows_per_cloud, const int *cnt, int *tilemap, unsigned int *hdr, void *stream);
/* The same chain (only p) over a LIST of rows instead of whole tiles (round 5): prcnn_pooled_rows lists the distinct pooled rows of all
 * clouds back to back -- rowmap[0 .. hdr[1]), row c * rows_per_cloud + j for j < max(cnt[c], 1), the clouds in the order a device counter
 * hands out; hdr (4 u32) is zeroed by the call unless hdr_is_zero -- and prcnn_rcnn_point_mlp_rows computes p for exactly those rows,
 * 64 list entries per tile whichever RoIs they belong to (a RoI's 47 distinct rows no longer fill a tile of 64).  p rows that are not
 * listed are left as they are.  Per row the arithmetic is prcnn_rcnn_point_mlp's: same bits. */
int prcnn_pooled_rows(int clouds, int rows_per_cloud, const int *cnt, int *rowmap, unsigned int *hdr, int hdr_is_zero, void *stream);
int prcnn_rcnn_point_mlp_rows(long r, int ld, int fcol, const float *rows, const float *wu1, const float *bu1,
                              const float *wu2, const float *bu2, const float *wm, const float *bm, const float *wp,
                              const float *bp, float *p, const int *rowmap, const unsigned int *hdr, void *stream);

/* One 128-wide shared-MLP / Conv1d layer (pytorch_utils.py:35-101 with BN folded) on the same tiled MFMA kernel:
 * out (r,128) = act(A0 w[0:128] [+ A1 w[128:256]] + bias), r % 64 == 0; A0 = src0 rows (128 floats at column col0, row
 * stride ld0), npanel = 2 adds A1 = src1 rows (col1, ld1): a K = 256 layer over two 128-wide halves. w k-major. */
/* the same 128 -> 128 layer (npanel = 1) over a LIST of rows: out[r] = act(src[r] @ w + bias) for r = rowmap[0 .. hdr[1]), rows that are not
 * listed are neither read nor written (round 5; per row the arithmetic of prcnn_rows_gemm128 / prcnn_packed_layer: same bits) */
int prcnn_rows_gemm128_rows(long r, const float *src, int ld, int col, const float *w, const float *bias, int relu, float *out,
                            const int *rowmap, const unsigned int *hdr, void *stream);
int prcnn_rows_gemm128(long r, int npanel, const float *src0, int ld0, int col0, const float *src1, int ld1, int col1,
                       const float *w, const float *bias, int relu, float *out, void *stream);

/* One whole coordinates-only set-abstraction scale (first RPN SA level: QueryAndGroup without input features ->
 * 3-layer shared MLP + ReLU -> max over nsample; pointnet2_modules.py:37-53, pointnet2_utils.py:241-264) in one
 * VALU kernel: a grouped row lives in one lane from the gather to the max, weights are scalar operands.
 * xyz (b,n,3), new_xyz (b,m,3), idx (b,m,nsample); w1 (>=3,c1), w2 (c1,c2), w3 (c2,c3) k-major, BN folded;
 * out[(b*m rows)][out_col .. out_col+c3), row stride out_stride.
 * Supported (c1,c2,c3,nsample): (16,16,32,16), (32,32,64,32) -- prcnn_sa_xyz_mlp_supported returns 1 for those. */
int prcnn_sa_xyz_mlp_supported(int c1, int c2, int c3, int nsample);
int prcnn_sa_xyz_mlp(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *new_xyz,
                     const float *xyz, const int *idx, const float *w1, const float *b1, const float *w2,
                     const float *b2, const float *w3, const float *b3, float *out, int out_stride, int out_col,
                     void *stream);
/* max over ns consecutive rows: in (rows_out*ns, c) -> out[r][out_col..out_col+c), row stride out_stride
 * (F.max_pool2d over nsample, pointnet2_modules.py:41-44, on the point-major MLP output). */
int prcnn_maxpool_pm(long rows_out, int ns, int c, const float *in, float *out, int out_stride,
                     int out_col, void *stream);
/* three_interpolate (interpolate_gpu.cu:77-97) on point-major features (b,m,c); result written into a
 * column slice of out (b, n, out_stride). */
int prcnn_three_interpolate_pm(int b, int c, int m, int n, const float *features, const int *idx,
                               const float *weight, float *out, int out_stride, int out_col, void *stream);
/* The input of a feature-propagation module in one pass (pointnet2_modules.py:139-151: three_interpolate, then torch.cat with
 * the skip features): out (b, n, c + c_skip) = [interpolated (same arithmetic as above) | skip (b, n, c_skip)];
 * c, c_skip multiples of 4, pointers 16-byte aligned. */
int prcnn_three_interpolate_cat_pm(int b, int c, int m, int n, const float *features, const int *idx, const float *weight,
                                   const float *skip, int c_skip, float *out, void *stream);

/* ---- iou3d_cuda ---------------------------------------------------------------------- */

/* boxes_overlap_bev_gpu  src/iou3d.cpp:31-50 -> src/iou3d_kernel.cu:223-234.
 * boxes_a (na,5), boxes_b (nb,5) [x1,y1,x2,y2,ry] -> ans (na,nb) intersection area. */
int prcnn_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                            float *ans_overlap, void *stream);
/* boxes_iou_bev_gpu  src/iou3d.cpp:52-71 -> src/iou3d_kernel.cu:236-248. */
int prcnn_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                        float *ans_iou, void *stream);
/* nms_gpu  src/iou3d.cpp:73-120 -> src/iou3d_kernel.cu:250-292 + host reduce.
 * boxes (n,5) DEVICE, score-sorted; keep (n) HOST int64.  BLOCKING (as the reference's
 * cudaMemcpy is).  Returns num_to_keep >= 0, or a negative PRCNN_E* code. */
int prcnn_nms(int boxes_num, const float *boxes, long long *keep_host, float thresh, void *stream);
/* nms_normal_gpu  src/iou3d.cpp:123-170 -> src/iou3d_kernel.cu:306-348. */
int prcnn_nms_normal(int boxes_num, const float *boxes, long long *keep_host, float thresh, void *stream);

/* Device-resident, batched greedy NMS (no host round trip): `nprob` independent problems.
 * boxes (nprob, n_max, 5) score-sorted per problem; counts (nprob) DEVICE i32 = valid rows of
 * each problem (NULL -> all n_max).  Writes keep (nprob, max_keep) i32 (row indices, padded
 * with -1) and num_keep (nprob) i32 = min(#kept, max_keep): exactly the first max_keep
 * entries the reference's full greedy pass would return.  rotated != 0 -> iou_bev, else
 * iou_normal.  Asynchronous. */
int prcnn_nms_device(int nprob, int n_max, const int *counts, const float *boxes, float thresh,
                     int rotated, int max_keep, int *keep, int *num_keep, void *stream);

/* The per-point RCNN inputs of point_rcnn.py:44-52 and rcnn_net.py:131-137 in one launch: seg[r] = sigmoid(scores[r]) > thresh ? 1 : 0,
 * depth[r] = |xyz[r]|, depth_norm[r] = depth[r] / 70 - 0.5  (rows = b * n; the reference's Python: sigmoid, compare, cast, norm, divide,
 * subtract -- six launches). */
int prcnn_point_aux(long rows, float thresh, const float *scores, const float *xyz, float *seg, float *depth, float *depth_norm,
                    void *stream);

/* Whole RPN proposal layer (lib/rpn/proposal_layer.py:15-119 + decode_bbox_target of
 * lib/utils/bbox_transform.py:24-121, distance-based variant) in five launches and no host sync:
 * decode -> per-scene score sort -> (0,40] / (40,80] m band selection (top 70 % / 30 % of pre_nms_top_n,
 * with the "no far points" fallback) -> batched NMS -> rois (b, post_nms_top_n, 7) + raw scores, zero padded.
 * xyz (b,n,3), scores (b,n), reg (b,n,channels); anchor_size_host = 3 floats in HOST memory (h,w,l).
 * n <= 65536 per scene (round 5: the chunked sort and the band scan take any n; tools/cfgs/double.yaml has 32768). */
int prcnn_rpn_proposals(int b, int n, int channels, float loc_scope, float loc_bin_size, int num_head_bin,
                        int xz_fine, const float *anchor_size_host, int pre_nms_top_n, int post_nms_top_n,
                        float nms_thresh, int rotated_nms, const float *xyz, const float *scores,
                        const float *reg, float *rois, float *roi_scores, void *stream);
/* The same layer over boxes decoded already (prcnn_rpn_tail_lin_boxes): boxes (b,n,7), scores (b,n). */
int prcnn_rpn_proposals_boxes(int b, int n, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int rotated_nms,
                              const float *scores, const float *boxes, float *rois, float *roi_scores, void *stream);

/* Final detection stage of eval_rcnn.py (tools/eval_rcnn.py:506-530 decode with get_xz_fine = get_ry_fine
 * = True, :611-629 score threshold + rotated NMS) in three launches and no host sync.
 * rois (b,m,7), rcnn_reg (b,m,channels), rcnn_cls (b,m) raw scores, m <= 128.
 * pred_boxes3d (b,m,7) = every RoI's decoded box (RoI order); boxes (b,m,7) / scores (b,m) raw = survivors of
 * sigmoid(score) > score_thresh and rotated NMS, descending score, zero padded; num (b) i32. */
int prcnn_rcnn_postprocess(int b, int m, int channels, float loc_scope, float loc_bin_size, int num_head_bin,
                           int y_by_bin, float loc_y_scope, float loc_y_bin_size, const float *anchor_size_host,
                           float score_thresh, float nms_thresh, const float *rois, const float *rcnn_reg,
                           const float *rcnn_cls, float *pred_boxes3d, float *boxes, float *scores, int *num,
                           void *stream);
/* The same with the results as one BLOB per batch of scenes_per_blob scenes (b a multiple of it): blobs (b / spb, spb (8 m + 1)) f32,
 * each [spb m 7 boxes | spb m scores | spb num as i32 bits]: a launch over several batches hands every batch's detections to the
 * host with one copy (round 5). */
int prcnn_rcnn_postprocess_blobs(int b, int m, int channels, float loc_scope, float loc_bin_size, int num_head_bin,
                                 int y_by_bin, float loc_y_scope, float loc_y_bin_size, const float *anchor_size_host,
                                 float score_thresh, float nms_thresh, const float *rois, const float *rcnn_reg,
                                 const float *rcnn_cls, float *pred_boxes3d, float *blobs, int scenes_per_blob, void *stream);

/* ---- roipool3d_cuda ------------------------------------------------------------------ */

/* forward  src/roipool3d.cpp:48-79 -> roipool3dLauncher src/roipool3d_kernel.cu:209-237.
 * xyz (B,N,3), boxes3d (B,M,7) already enlarged, pts_feature (B,N,C) ->
 * pooled_features (B,M,S,3+C) and pooled_empty_flag (B,M), both zero-filled by the caller;
 * rows of empty boxes are left untouched. */
int prcnn_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                    int sampled_pts_num, const float *xyz, const float *boxes3d,
                    const float *pts_feature, float *pooled_features, int *pooled_empty_flag,
                    void *stream);

/* RCNN input assembly in one pass (lib/net/rcnn_net.py:139-163 + roipool3d_utils.py:7-28 + kitti_utils.py:150-160):
 * enlarge the RoIs by pool_extra_width, pool the first `sampled` points per box (same selection as prcnn_roipool3d),
 * move the pooled coordinates into the RoI's canonical frame and write rows
 * [x', y', z', seg mask, depth, 0, 0, 0 | c features] (c % 4 == 0).  rois (b,m,7) are the UN-enlarged proposals;
 * feats (b,n,c) point-major; seg_mask, depth (b,n); pooled (b,m,sampled,8+c) need not be cleared; empty (b,m) i32.
 * pooled_cnt (b,m) i32, optional (NULL = off): number of DISTINCT rows of each box (min(#points in box, sampled), >= 1;
 * rows s >= cnt are the wrap-around copies of row s % cnt, roipool3d_kernel.cu:152-159).  When given, the feature
 * columns are written only for rows < round_up(cnt, 64): the coordinate / mask / depth columns of all rows are. */
/* pxyz / aabb (optional, both or neither): the cloud's spatial groups from prcnn_point_groups -- the selection then tests
 * pts_num / 64 group boxes and reads the few groups that can hold a point of the box instead of sweeping all points; same
 * points, same order (the hits are sorted by original index).  pts_num % 64 == 0, <= 65536. */
int prcnn_roipool3d_canonical(int batch_size, int pts_num, int boxes_num, int feature_len, int sampled_pts_num,
                              float pool_extra_width, const float *xyz, const float *rois, const float *feats,
                              const float *seg_mask, const float *depth, float *pooled, int *pooled_empty_flag,
                              int *pooled_cnt, const float *pxyz, const float *aabb, void *stream);
/* the same with xyz_out (b,m,sampled,3), optional (NULL = off): the rows' canonical coordinates once more as dense clouds -- what the
 * RCNN stage's sampling and ball queries (rcnn_net.py:165-175: pointnet2 SA modules over xyz = pooled[..., 0:3]) read. */
int prcnn_roipool3d_canonical_xyz(int batch_size, int pts_num, int boxes_num, int feature_len, int sampled_pts_num,
                                  float pool_extra_width, const float *xyz, const float *rois, const float *feats,
                                  const float *seg_mask, const float *depth, float *pooled, int *pooled_empty_flag,
                                  int *pooled_cnt, const float *pxyz, const float *aabb, float *xyz_out, void *stream);
/* Spatial groups of clouds xyz (b,n,3), n % 64 == 0, n <= 65536: pxyz (b,n,4) = the points in Morton order over (x,z) with the
 * original index in the 4th lane (int bits), aabb (b, n/64, 2, 4) = min / max corner of every 64-point group.  Any box-vs-cloud
 * sweep (RoI pooling here) can cull by group.  Not part of the reference ABI. */
int prcnn_point_groups(int b, int n, const float *xyz, float *pxyz, float *aabb, void *stream);

/* The reference module's two HOST utilities (CPU tensors, unbatched; they serve its dataset / GT-database code):
 * pts_in_boxes3d_cpu  roipool3d.cpp:97-125 -> flags (boxes_num, pts_num) i64 in {0,1};
 * roipool3d_cpu       roipool3d.cpp:127-195 -> pooled_pts (boxes_num,sampled,3), pooled_features
 * (boxes_num,sampled,feature_len), empty (boxes_num) i64; rows of empty boxes stay as the caller left them.
 * HOST pointers, no stream. */
int prcnn_host_pts_in_boxes3d(int boxes_num, int pts_num, const float *pts, const float *boxes3d, long long *flags);
int prcnn_host_roipool3d(int boxes_num, int pts_num, int feature_len, int sampled, const float *pts,
                         const float *boxes3d, const float *pts_feature, float *pooled_pts, float *pooled_features,
                         long long *empty);

/* ---- evaluate/rotate_iou.py ---------------------------------------------------------- */

/* rotate_iou_gpu_eval  evaluate/rotate_iou.py:294-329 (kernel :261-291).
 * boxes (n,5), query_boxes (k,5) [cx,cy,w,h,angle] DEVICE -> iou (n,k);
 * criterion -1 IoU, 0 /area(query), 1 /area(box), 2 raw intersection (as the kernel passes
 * (query, box) to the device function, :287-291). */
int prcnn_rotate_iou_eval(int n, int k, const float *boxes, const float *query_boxes,
                          float *iou, int criterion, void *stream);

/* Block-diagonal rotate_iou_gpu_eval: segment s (one image) pairs boxes[box_off[s]..box_off[s+1]) with
 * query_boxes[q_off[s]..q_off[s+1]) and writes its row-major block at out_off[s]; one launch for a whole split
 * instead of the ~50 dense parts of evaluate/eval2.py:352-424 (calculate_iou_partly), whose cross-image pairs
 * are discarded.  out_off (nseg+1) i64, box_off / q_off (nseg+1) i32, all DEVICE; total = out_off[nseg]. */
int prcnn_rotate_iou_eval_segmented(int nseg, long long total, const long long *out_off, const int *box_off,
                                    const int *q_off, const float *boxes, const float *query_boxes, float *iou,
                                    int criterion, void *stream);

/* ---- lib/datasets/kitti_rcnn_dataset.py: the network-input stage on the device ---------- */

/* get_lidar + get_valid_flag + the near/far sampler of get_rpn_sample (kitti_rcnn_dataset.py:249-324) with
 * calibration.py:51-71, one workgroup per scene.  raw (b,n_max,stride) f32, stride 3 or 4, as read from velodyne .bin
 * (lidar_frame = 1) or already rectified (0); counts (b) i32; calib (b,35) f32 = V2C 3x4 | R0 3x3 | P2 3x4 | img h, w;
 * image_filter 0/1 (in-image + depth >= 0 test); scope_host = 6 HOST floats x0,x1,y0,y1,z0,z1 or NULL;
 * seeds (b) u64 DEVICE.  -> out (b,npoints,3), stats (b,3) i32 = #valid, #near, #far, choice (b,npoints) i32 = raw
 * index of each output point (may be NULL).  npoints <= 16384.
 * The subset is random (distinct-key selection + key-sorted shuffle): same distribution as the reference's
 * np.random.choice / shuffle, not the same draws; transform and filter are bitwise the reference's (see prcnn_valid_flags). */
int prcnn_input_stage(int b, int n_max, int stride, int lidar_frame, int image_filter, const int *counts,
                      const float *raw, const float *calib, const float *scope_host, int npoints, float far_depth,
                      int npoints_faraway, const unsigned long long *seeds, float *out, int *stats, int *choice,
                      void *stream);

/* get_valid_flag (kitti_rcnn_dataset.py:201-222) + Calibration.lidar_to_rect / rect_to_img (calibration.py:51-71) for whole
 * batches -- the front half of prcnn_input_stage as an operator, same arguments.  -> cls (b,n_max) u8: 0 = not valid,
 * 1 = valid and z < far_depth, 2 = valid beyond (rows >= counts[b]: 0); rect (b,n_max,3) f32 rectified coordinates (may be
 * NULL).  Bit-identical to the reference's numpy results: float32 np.dot is a chain of fused multiply-adds over the inner
 * index, which the kernel reproduces (tests/golden g11, recorded by running the reference's code). */
int prcnn_valid_flags(int b, int n_max, int stride, int lidar_frame, int image_filter, const int *counts, const float *raw,
                      const float *calib, const float *scope_host, float far_depth, float *rect, unsigned char *cls,
                      void *stream);

/* ---- evaluate/eval2.py: host-side matching of the AP evaluator (HOST pointers, f64 / i64) ---- */

/* compute_statistics_jit  evaluate/eval2.py:170-289 for one image.  overlaps (n_dt,n_gt) row-major,
 * gt_datas (n_gt,5) = [bbox x4, alpha], dt_datas (n_dt,6) = [bbox x4, alpha, score], ignored_* in {-1,0,1},
 * dc_bboxes (n_dc,4).  -> tp_fp_fn[3], similarity, thresholds (room for n_gt) and their count. */
int prcnn_kitti_image_stats(int n_gt, int n_dt, int n_dc, const double *overlaps, const double *gt_datas,
                            const double *dt_datas, const long long *ignored_gt, const long long *ignored_det,
                            const double *dc_bboxes, int metric, double min_overlap, double thresh, int compute_fp,
                            int compute_aos, long long *tp_fp_fn, double *similarity, double *thresholds,
                            int *n_thresholds);

/* First pass of eval_class (evaluate/eval2.py:512-527) over all images: scores of matched detections
 * (compute_fp = False, thresh = 0).  Per-image arrays are concatenated; overlaps_flat holds the (n_dt,n_gt)
 * blocks back to back.  scores_out has room for sum(gt_nums). */
int prcnn_kitti_collect_scores(int n_img, const long long *gt_nums, const long long *dt_nums,
                               const double *overlaps_flat, const double *gt_datas, const double *dt_datas,
                               const long long *ignored_gts, const long long *ignored_dets, int metric,
                               double min_overlap, double *scores_out, long long *n_scores);

/* fused_compute_statistics  evaluate/eval2.py:300-349 over all images:
 * pr (n_thresh,4) += [tp, fp, fn, similarity] at every score threshold. */
int prcnn_kitti_accumulate_pr(int n_img, const long long *gt_nums, const long long *dt_nums, const long long *dc_nums,
                              const double *overlaps_flat, const double *gt_datas, const double *dt_datas,
                              const double *dontcares, const long long *ignored_gts, const long long *ignored_dets,
                              int metric, double min_overlap, const double *thresholds, int n_thresh, int compute_aos,
                              double *pr);

#ifdef __cplusplus
}
#endif
#endif
