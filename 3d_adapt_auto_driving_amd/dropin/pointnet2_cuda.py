"""Drop-in for the reference's ``pointnet2_cuda`` pybind module
(pointrcnn/pointnet2_lib/pointnet2/src/pointnet2_api.cpp:10-24): same function names, argument
order and pre-allocated-output convention, bound to libprcnn_hip.so through the C ABI
(include/prcnn_hip.h).  Put this directory on ``sys.path`` and the reference's
``pointnet2_utils.py`` (``import pointnet2_cuda as pointnet2``, :7) imports it unmodified.

Like the reference wrappers, kernels go to torch's current stream.  Inputs must be CUDA(HIP),
contiguous, f32/i32 (ball_query.cpp:10-12 checks only ball_query's inputs; we check all).
"""
import ctypes
import importlib
import os
import sys

import torch

_pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg_dir) not in sys.path:
    sys.path.insert(0, os.path.dirname(_pkg_dir))
_lib = importlib.import_module(os.path.basename(_pkg_dir) + "._lib")
IS_HIP_EXTENSION = True     # marks the real extension (the test suite's CPU stand-ins do not carry it)


def _chk(dtype, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("pointnet2_cuda: tensor must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError("pointnet2_cuda: tensor must be contiguous")
        if t.dtype != dtype:
            raise RuntimeError("pointnet2_cuda: expected %s, got %s" % (dtype, t.dtype))


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _chk(torch.float32, new_xyz, xyz); _chk(torch.int32, idx)
    _lib.call("prcnn_ball_query", b, n, m, radius, nsample, new_xyz.data_ptr(), xyz.data_ptr(),
              idx.data_ptr(), _lib.current_stream(xyz))
    return 1


def fps_new_xyz_supported(n, m):
    """prcnn_fps_new_xyz serves every shape since round 5 (one launch for n <= 1024 and in the speculative kernel's range
    2048 < n <= 16384, m >= 256; elsewhere FPS over an internal
    distance scratch + a gather launch inside the library)."""
    return True


def ball_query_full_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """ball_query_wrapper with every slot of idx written (an empty ball's row holds zeros): idx may be uninitialised memory"""
    _chk(torch.float32, new_xyz, xyz); _chk(torch.int32, idx)
    _lib.call("prcnn_ball_query_full", b, n, m, radius, nsample, new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(),
              _lib.current_stream(xyz))
    return 1


def point_aux_wrapper(scores, xyz, thresh, seg, depth, depth_norm):
    """scores (B,N), xyz (B,N,3) -> seg (B,N) 0/1 floats = sigmoid(score) > thresh, depth (B,N) = |xyz|, depth_norm = depth / 70 - 0.5"""
    _chk(torch.float32, scores, xyz, seg, depth, depth_norm)
    _lib.call("prcnn_point_aux", scores.numel(), float(thresh), scores.data_ptr(), xyz.data_ptr(), seg.data_ptr(), depth.data_ptr(),
              depth_norm.data_ptr(), _lib.current_stream(xyz))


def fps_new_xyz_wrapper(xyz, m):
    """xyz (b,n,3) -> (idx (b,m) i32, new_xyz (b,m,3)): furthest_point_sampling_wrapper + the gather of the selected
    coordinates in one launch (csrc/fps.hip); shapes: fps_new_xyz_supported."""
    _chk(torch.float32, xyz)
    b, n, _ = xyz.shape
    idx = torch.empty((b, m), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((b, m, 3), dtype=torch.float32, device=xyz.device)
    _lib.call("prcnn_fps_new_xyz", b, n, m, xyz.data_ptr(), idx.data_ptr(), new_xyz.data_ptr(), _lib.current_stream(xyz))
    return idx, new_xyz


def ball_query_limit_wrapper(b, n, m, radius, nsample, new_xyz, xyz, limit, idx):
    """ball_query_wrapper over clouds whose points k >= limit[cloud] are copies of point k % limit[cloud] (pooled RoI rows):
    scans the first limit[cloud] points only -- the same distinct points per ball, slots past them repeat the first hit."""
    _chk(torch.float32, new_xyz, xyz); _chk(torch.int32, idx, limit)
    _lib.call("prcnn_ball_query_limit", b, n, m, radius, nsample, new_xyz.data_ptr(), xyz.data_ptr(), limit.data_ptr(),
              idx.data_ptr(), _lib.current_stream(xyz))
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _chk(torch.float32, points, out); _chk(torch.int32, idx)
    _lib.call("prcnn_group_points", b, c, n, npoints, nsample, points.data_ptr(), idx.data_ptr(),
              out.data_ptr(), _lib.current_stream(points))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _chk(torch.float32, grad_out, grad_points); _chk(torch.int32, idx)
    _lib.call("prcnn_group_points_grad", b, c, n, npoints, nsample, grad_out.data_ptr(), idx.data_ptr(),
              grad_points.data_ptr(), _lib.current_stream(grad_out))
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _chk(torch.float32, points, out); _chk(torch.int32, idx)
    _lib.call("prcnn_gather_points", b, c, n, npoints, points.data_ptr(), idx.data_ptr(),
              out.data_ptr(), _lib.current_stream(points))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _chk(torch.float32, grad_out, grad_points); _chk(torch.int32, idx)
    _lib.call("prcnn_gather_points_grad", b, c, n, npoints, grad_out.data_ptr(), idx.data_ptr(),
              grad_points.data_ptr(), _lib.current_stream(grad_out))
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _chk(torch.float32, points, temp); _chk(torch.int32, idx)
    _lib.call("prcnn_furthest_point_sampling", b, n, m, points.data_ptr(), temp.data_ptr(),
              idx.data_ptr(), _lib.current_stream(points))
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _chk(torch.float32, unknown, known, dist2); _chk(torch.int32, idx)
    _lib.call("prcnn_three_nn", b, n, m, unknown.data_ptr(), known.data_ptr(), dist2.data_ptr(),
              idx.data_ptr(), _lib.current_stream(unknown))


def set_fps_arithmetic(mode):
    """0: the source's arithmetic (default, the parity contract); 1: the reference's hipcc-built kernel binary's (include/prcnn_hip.h)"""
    _lib.call("prcnn_set_fps_arithmetic", int(mode))


def three_nn_weights_wrapper(b, n, m, unknown, known, idx, weight):
    """three_nn + the FP module's inverse-distance weights in one kernel (engine-side entry, not in the reference's API)"""
    _chk(torch.float32, unknown, known, weight); _chk(torch.int32, idx)
    _lib.call("prcnn_three_nn_weights", b, n, m, unknown.data_ptr(), known.data_ptr(), idx.data_ptr(), weight.data_ptr(),
              _lib.current_stream(unknown))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _chk(torch.float32, points, weight, out); _chk(torch.int32, idx)
    _lib.call("prcnn_three_interpolate", b, c, m, n, points.data_ptr(), idx.data_ptr(),
              weight.data_ptr(), out.data_ptr(), _lib.current_stream(points))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _chk(torch.float32, grad_out, weight, grad_points); _chk(torch.int32, idx)
    _lib.call("prcnn_three_interpolate_grad", b, c, n, m, grad_out.data_ptr(), idx.data_ptr(),
              weight.data_ptr(), grad_points.data_ptr(), _lib.current_stream(grad_out))


# -- extension beyond the reference module: the fused QueryAndGroup path -------------------
def query_and_group_wrapper(b, n, m, c, radius, nsample, new_xyz, xyz, features, idx, out):
    _chk(torch.float32, new_xyz, xyz, out); _chk(torch.int32, idx)
    if features is not None:
        _chk(torch.float32, features)
    _lib.call("prcnn_query_and_group", b, n, m, c, radius, nsample, new_xyz.data_ptr(), xyz.data_ptr(),
              _lib.ptr(features), idx.data_ptr(), out.data_ptr(), _lib.current_stream(xyz))
    return 1


# -- extension beyond the reference module: shared-MLP epilogues -----------------------------------
def bias_relu_inplace_wrapper(x, bias):
    """x (B, C, ...) in place: x = relu(x + bias[c])."""
    _chk(torch.float32, x, bias)
    outer, c = x.size(0), x.size(1)
    inner = x.numel() // max(outer * c, 1)
    _lib.call("prcnn_bias_relu_inplace", outer, c, inner, bias.data_ptr(), x.data_ptr(), _lib.current_stream(x))
    return x


def maxpool_bias_relu_wrapper(x, bias, out):
    """x (B, C, npoint, nsample) raw conv output -> out (B, C, npoint) = relu(max_s x + bias[c])."""
    _chk(torch.float32, x, bias, out)
    _lib.call("prcnn_maxpool_bias_relu", x.size(0), x.size(1), x.size(2), x.size(3), bias.data_ptr(),
              x.data_ptr(), out.data_ptr(), _lib.current_stream(x))
    return out


# -- extension beyond the reference module: point-major (channels-last) forms ----------------------
def group_cat_pm_wrapper(b, n, m, c, nsample, new_xyz, xyz, features, idx, out):
    """features (b,n,c) point-major or None -> out (b, m*nsample, round_up(c,4)+4)."""
    _chk(torch.float32, new_xyz, xyz, out); _chk(torch.int32, idx)
    if features is not None:
        _chk(torch.float32, features)
    _lib.call("prcnn_group_cat_pm", b, n, m, c, nsample, new_xyz.data_ptr(), xyz.data_ptr(), _lib.ptr(features),
              idx.data_ptr(), out.data_ptr(), _lib.current_stream(xyz))
    return out


def maxpool_pm_wrapper(x, ns, out, out_col):
    """x (rows*ns, c) -> out (rows, stride)[:, out_col:out_col+c] = max over each run of ns rows."""
    _chk(torch.float32, x, out)
    rows = x.size(0) // ns
    _lib.call("prcnn_maxpool_pm", rows, ns, x.size(1), x.data_ptr(), out.data_ptr(), out.size(-1), out_col,
              _lib.current_stream(x))
    return out


def three_interpolate_pm_wrapper(features, idx, weight, out, out_col):
    """features (b,m,c), idx/weight (b,n,3) -> out (b,n,stride)[..., out_col:out_col+c]."""
    _chk(torch.float32, features, weight, out); _chk(torch.int32, idx)
    b, m, c = features.shape
    _lib.call("prcnn_three_interpolate_pm", b, c, m, idx.size(1), features.data_ptr(), idx.data_ptr(),
              weight.data_ptr(), out.data_ptr(), out.size(-1), out_col, _lib.current_stream(features))
    return out


def three_interpolate_cat_pm_wrapper(features, idx, weight, skip, out):
    """features (b,m,c), idx/weight (b,n,3), skip (b,n,c_skip) -> out (b,n,c+c_skip) = [interpolated | skip]: the input of a
    feature-propagation module in one launch."""
    _chk(torch.float32, features, weight, skip, out); _chk(torch.int32, idx)
    b, m, c = features.shape
    if out.size(-1) != c + skip.size(-1):
        raise ValueError("three_interpolate_cat_pm: out must be (b, n, c + c_skip)")
    _lib.call("prcnn_three_interpolate_cat_pm", b, c, m, idx.size(1), features.data_ptr(), idx.data_ptr(), weight.data_ptr(),
              skip.data_ptr(), skip.size(-1), out.data_ptr(), _lib.current_stream(features))
    return out


def gather_affine_relu_pm_wrapper(new_xyz, xyz, P, wxyz, idx, out):
    """P (b,n,cout), wxyz (3,cout), idx (b,m,ns) -> out (b, m*ns, cout) = relu(P[idx] + wxyz.(xyz[idx]-centre))."""
    _chk(torch.float32, new_xyz, xyz, P, wxyz, out); _chk(torch.int32, idx)
    b, n, cout = P.shape
    _lib.call("prcnn_gather_affine_relu_pm", b, n, idx.size(1), cout, idx.size(2), new_xyz.data_ptr(), xyz.data_ptr(),
              P.data_ptr(), wxyz.data_ptr(), idx.data_ptr(), out.data_ptr(), _lib.current_stream(xyz))
    return out


def sa_mlp_fused_supported(c1, c2, c3, nsample):
    return c1 == 128 and c2 == 128 and c3 in (128, 256) and nsample == 64


def sa_mlp_fused_wrapper(new_xyz, xyz, P, wxyz, idx, w2t, b2, w3t, b3, out, out_col):
    """gather + 3-layer shared MLP + max over nsample in one MFMA kernel (csrc/sa_mlp_fused.hip).
    P (b,n,128), wxyz (3,128), idx (b,m,64), w2t (128,128), w3t (128,c3), out (b,m,stride)."""
    _chk(torch.float32, new_xyz, xyz, P, wxyz, w2t, b2, w3t, b3, out); _chk(torch.int32, idx)
    b, n, c1 = P.shape
    _lib.call("prcnn_sa_mlp_fused", b, n, idx.size(1), idx.size(2), c1, w2t.size(1), w3t.size(1),
              new_xyz.data_ptr(), xyz.data_ptr(), P.data_ptr(), wxyz.data_ptr(), idx.data_ptr(), w2t.data_ptr(),
              b2.data_ptr(), w3t.data_ptr(), b3.data_ptr(), out.data_ptr(), out.size(-1), out_col,
              _lib.current_stream(xyz))
    return out


class BallPack:
    """Distinct grouped rows of an index tensor as 64-row tiles (prcnn_ball_pack); device-resident, no host sync."""
    __slots__ = ("idx", "limit", "rep", "crep", "rowinfo", "rowdxyz", "tilecloud", "hdr", "max_tiles")

    def tensors(self):
        # (an index tensor that was never written is a shape on the "meta" device: nothing to keep alive or to mark)
        return tuple(t for t in (self.idx, self.rowinfo, self.rowdxyz, self.tilecloud, self.hdr) if t is not None and t.is_cuda)

    def record_stream(self, stream):
        for t in self.tensors():
            t.record_stream(stream)


def ball_pack_wrapper(idx, xyz, new_xyz, limit=None, rep=None, crep=None, hdr=None):
    """idx (b,m,nsample) i32 from a ball query of new_xyz (b,m,3) in xyz (b,n,3) -> BallPack for sa_packed_mlp_wrapper.
    limit (b) i32, optional: the points k >= limit[cloud] of a cloud are copies of point k % limit[cloud] (RoI pooling's
    wrap-around fill): dropped as well.  rep (b,n) i32, optional: rep[cloud][k] = the lowest-indexed exact copy of point k
    (dup_rep_wrapper): the slots whose point is not its own representative are dropped too (prcnn_ball_pack_rep).  crep (b,m) i32,
    optional: the same map over the centres -- a centre that copies an earlier one gets no rows (its output row is never written).
    hdr (4) i32, optional: a header that IS ZERO already (a slice of an arena the caller zeroed: no memset launch per row list)."""
    _chk(torch.int32, idx); _chk(torch.float32, xyz, new_xyz)
    if limit is not None:
        _chk(torch.int32, limit)
    if rep is not None:
        _chk(torch.int32, rep)
        if tuple(rep.shape) != (idx.shape[0], xyz.shape[1]):
            raise ValueError("ball_pack: rep must be (b, n)")
    if crep is not None:
        _chk(torch.int32, crep)
        if tuple(crep.shape) != (idx.shape[0], idx.shape[1]):
            raise ValueError("ball_pack: crep must be (b, m)")
    b, m, ns = idx.shape
    cap = (m * ns + 63) // 64
    pk = BallPack()
    pk.idx, pk.limit, pk.rep, pk.crep = idx, limit, rep, crep
    pk.rowinfo = torch.empty((b * cap * 64,), dtype=torch.int32, device=idx.device)
    pk.rowdxyz = torch.empty((b * cap * 64, 4), dtype=torch.float32, device=idx.device)
    pk.tilecloud = torch.empty((b * cap,), dtype=torch.int32, device=idx.device)
    pk.max_tiles = b * cap
    if hdr is not None:
        _chk(torch.int32, hdr)
        if hdr.numel() != 4:
            raise ValueError("ball_pack: hdr must hold 4 int32")
        pk.hdr = hdr
        _lib.call("prcnn_ball_pack_ex", b, b, xyz.size(1), m, ns, idx.data_ptr(), _lib.ptr(limit), _lib.ptr(rep), _lib.ptr(crep),
                  xyz.data_ptr(), new_xyz.data_ptr(), pk.rowinfo.data_ptr(), pk.rowdxyz.data_ptr(), pk.tilecloud.data_ptr(),
                  pk.hdr.data_ptr(), 1, _lib.current_stream(idx))
        return pk
    pk.hdr = torch.empty((4,), dtype=torch.int32, device=idx.device)
    if rep is not None or crep is not None:
        _lib.call("prcnn_ball_pack_rep", b, xyz.size(1), m, ns, idx.data_ptr(), _lib.ptr(limit), _lib.ptr(rep), _lib.ptr(crep),
                  xyz.data_ptr(), new_xyz.data_ptr(), pk.rowinfo.data_ptr(), pk.rowdxyz.data_ptr(), pk.tilecloud.data_ptr(),
                  pk.hdr.data_ptr(), _lib.current_stream(idx))
        return pk
    _lib.call("prcnn_ball_pack", b, xyz.size(1), m, ns, idx.data_ptr(), _lib.ptr(limit), xyz.data_ptr(), new_xyz.data_ptr(),
              pk.rowinfo.data_ptr(), pk.rowdxyz.data_ptr(), pk.tilecloud.data_ptr(),
              pk.hdr.data_ptr(), _lib.current_stream(idx))
    return pk


def rcnn_roi_geometry_supported(n, m1, ns1, m2, ns2):
    return n == 512 and m1 == 128 and m2 == 32 and 1 <= ns1 <= 64 and 1 <= ns2 <= 64


def rcnn_roi_geometry_wrapper(xyz, limit, m1, r1, ns1, m2, r2, ns2):
    """The geometry of the RCNN's two sampled SA levels for every RoI cloud in ONE launch (prcnn_rcnn_roi_geometry): xyz (b,512,3),
    limit (b) i32 -> (new_xyz1 (b,128,3), idx1 (b,128,ns1), rep1 (b,128), new_xyz2 (b,32,3), idx2 (b,32,ns2), rep2 (b,32)) -- what
    fps_new_xyz_wrapper, ball_query_limit_wrapper, dup_rep_wrapper, fps_new_xyz_wrapper, ball_query_wrapper (into zeros),
    dup_rep_wrapper return one after the other."""
    _chk(torch.float32, xyz); _chk(torch.int32, limit)
    b, n, _ = xyz.shape
    dev = xyz.device
    new1 = torch.empty((b, m1, 3), dtype=torch.float32, device=dev)
    idx1 = torch.empty((b, m1, ns1), dtype=torch.int32, device=dev)
    rep1 = torch.empty((b, m1), dtype=torch.int32, device=dev)
    new2 = torch.empty((b, m2, 3), dtype=torch.float32, device=dev)
    idx2 = torch.empty((b, m2, ns2), dtype=torch.int32, device=dev)
    rep2 = torch.empty((b, m2), dtype=torch.int32, device=dev)
    _lib.call("prcnn_rcnn_roi_geometry", b, n, m1, float(r1), ns1, m2, float(r2), ns2, xyz.data_ptr(), limit.data_ptr(), new1.data_ptr(),
              idx1.data_ptr(), rep1.data_ptr(), new2.data_ptr(), idx2.data_ptr(), rep2.data_ptr(), _lib.current_stream(xyz))
    return new1, idx1, rep1, new2, idx2, rep2


_ARANGE = {}


def rcnn_roi_geometry_packs_wrapper(xyz, limit, m1, r1, ns1, m2, r2, ns2, hdr1=None, hdr2=None, want_idx=True, row_clouds=False, hdr3=None,
                                    group_all=False, hdr_c1=None, centre_rows=False):
    """rcnn_roi_geometry_wrapper + the two levels' distinct-row lists out of the same launch (prcnn_rcnn_roi_geometry_packs) ->
    (new_xyz1, idx1, rep1, new_xyz2, idx2, rep2, pack1, pack2): pack1 == ball_pack_wrapper(idx1, xyz, new_xyz1, limit, None, rep1),
    pack2 == ball_pack_wrapper(idx2, new_xyz1, new_xyz2, None, rep1, rep2) -- the same rows per cloud, the same tiles.
    hdr1 / hdr2 (4) i32, optional: headers that ARE ZERO already (slices of an arena the caller zeroed).  want_idx=False: idx1 / idx2 are
    not written -- what comes back in their place (and in the packs' .idx) are tensors of the right SHAPE on the "meta" device (no
    storage: reading a value raises): the packed MLP wrappers only ask them for their shape.  row_clouds=True: the lists in the form whose ROWS carry their
    cloud (packs with .tilecloud None: rows of all clouds back to back, no padded last tile per cloud) -- sa_packed_mlp_wrapper takes
    them, the other consumers of a BallPack do not.  group_all=True (with row_clouds; hdr3 as hdr1 / hdr2): a ninth value, the list of the
    GroupAll level above -- every cloud ONE group of its m2 level-2 centres, the copies among them dropped: the BallPack of the index
    tensor (b, 1, m2) = 0 .. m2-1 around the origin with rep = rep2 (sa_wide_fused3_wrapper takes it, with new_xyz = zeros (b, 1, 3)).
    centre_rows=True (hdr_c1 as the other headers): one more value at the end, (rowmap i32, hdr i32[4]) = the level-1 centres that are their
    own representatives as rows cloud * m1 + centre, hdr[1] of them -- the list rows_gemm128_rows_wrapper takes."""
    _chk(torch.float32, xyz); _chk(torch.int32, limit)
    b, n, _ = xyz.shape
    dev = xyz.device
    new1 = torch.empty((b, m1, 3), dtype=torch.float32, device=dev)
    if want_idx:
        idx1 = torch.empty((b, m1, ns1), dtype=torch.int32, device=dev)
        idx2 = torch.empty((b, m2, ns2), dtype=torch.int32, device=dev)
    else:
        # shapes only, on the "meta" device (a stride-0 view of one uninitialised int handed garbage to whoever read VALUES):
        # .shape and slices work, every operator of this module refuses the tensor (_chk: not a CUDA tensor), torch raises on reads
        idx1 = torch.empty((b, m1, ns1), dtype=torch.int32, device="meta")
        idx2 = torch.empty((b, m2, ns2), dtype=torch.int32, device="meta")
    rep1 = torch.empty((b, m1), dtype=torch.int32, device=dev)
    new2 = torch.empty((b, m2, 3), dtype=torch.float32, device=dev)
    rep2 = torch.empty((b, m2), dtype=torch.int32, device=dev)
    if (hdr1 is None) != (hdr2 is None):
        raise ValueError("rcnn_roi_geometry_packs: both headers or none")
    packs = []
    for idx, lim_, rep_, crep_, hdr in ((idx1, limit, None, rep1, hdr1), (idx2, None, rep1, rep2, hdr2)):
        m, ns = idx.shape[1], idx.shape[2]
        cap = (m * ns + 63) // 64
        pk = BallPack()
        pk.idx, pk.limit, pk.rep, pk.crep = idx, lim_, rep_, crep_
        pk.rowinfo = torch.empty((b * cap * 64,), dtype=torch.int32, device=dev)
        pk.rowdxyz = torch.empty((b * cap * 64, 4), dtype=torch.float32, device=dev)
        pk.tilecloud = None if row_clouds else torch.empty((b * cap,), dtype=torch.int32, device=dev)
        pk.max_tiles = b * cap
        if hdr is not None:
            _chk(torch.int32, hdr)
            if hdr.numel() != 4:
                raise ValueError("rcnn_roi_geometry_packs: a header must hold 4 int32")
        pk.hdr = hdr if hdr is not None else torch.empty((4,), dtype=torch.int32, device=dev)
        packs.append(pk)
    p1, p2 = packs
    p3 = None
    if group_all:
        if not row_clouds or (hdr3 is None) != (hdr1 is None):
            raise ValueError("rcnn_roi_geometry_packs: the GroupAll list comes in the row-carried form, its header like the others")
        p3 = BallPack()
        key = (m2, str(dev))
        if key not in _ARANGE:                               # (a constant of the shape: not a launch per call)
            _ARANGE[key] = torch.arange(m2, dtype=torch.int32, device=dev)
        p3.idx = _ARANGE[key].view(1, 1, m2).expand(b, 1, m2)
        p3.limit, p3.rep, p3.crep = None, rep2, None
        p3.rowinfo = torch.empty((b * m2,), dtype=torch.int32, device=dev)
        p3.rowdxyz = torch.empty((b * m2, 4), dtype=torch.float32, device=dev)
        p3.tilecloud = None
        p3.max_tiles = (b * m2 + 63) // 64
        if hdr3 is not None:
            _chk(torch.int32, hdr3)
        p3.hdr = hdr3 if hdr3 is not None else torch.empty((4,), dtype=torch.int32, device=dev)
    crows = None
    if centre_rows:
        if (hdr_c1 is None) != (hdr1 is None):
            raise ValueError("rcnn_roi_geometry_packs: the centre rows' header comes like the others")
        if hdr_c1 is not None:
            _chk(torch.int32, hdr_c1)
        crows = (torch.empty((b * m1,), dtype=torch.int32, device=dev), hdr_c1 if hdr_c1 is not None else torch.empty((4,), dtype=torch.int32, device=dev))
    _lib.call("prcnn_rcnn_roi_geometry_packs", b, n, m1, float(r1), ns1, m2, float(r2), ns2, xyz.data_ptr(), limit.data_ptr(), new1.data_ptr(),
              idx1.data_ptr() if want_idx else None, rep1.data_ptr(), new2.data_ptr(), idx2.data_ptr() if want_idx else None, rep2.data_ptr(),
              p1.rowinfo.data_ptr(), p1.rowdxyz.data_ptr(), _lib.ptr(p1.tilecloud), p1.hdr.data_ptr(),
              p2.rowinfo.data_ptr(), p2.rowdxyz.data_ptr(), _lib.ptr(p2.tilecloud), p2.hdr.data_ptr(),
              None if p3 is None else p3.rowinfo.data_ptr(), None if p3 is None else p3.rowdxyz.data_ptr(),
              None if p3 is None else p3.hdr.data_ptr(), None if crows is None else crows[0].data_ptr(),
              None if crows is None else crows[1].data_ptr(), 1 if hdr1 is not None else 0, _lib.current_stream(xyz))
    return (new1, idx1, rep1, new2, idx2, rep2, p1, p2) + ((p3,) if p3 is not None else ()) + ((crows,) if crows is not None else ())


def dup_rep_wrapper(sel, n, limit=None, prev=None):
    """sel (b,m) i32 = an FPS answer over clouds of n points whose copies are described by limit (b) i32 (points k >= limit are
    copies of k % limit) and / or prev (b,n) i32 (their representative map) -> rep (b,m) i32: for every sampled point the
    first sampled point with the same source (prcnn_dup_rep)."""
    _chk(torch.int32, sel)
    for t in (limit, prev):
        if t is not None:
            _chk(torch.int32, t)
    b, m = sel.shape
    rep = torch.empty((b, m), dtype=torch.int32, device=sel.device)
    _lib.call("prcnn_dup_rep", b, n, m, sel.data_ptr(), _lib.ptr(limit), _lib.ptr(prev), rep.data_ptr(), _lib.current_stream(sel))
    return rep


def ball_pack_groups_wrapper(idx, xyz, new_xyz, group, hdr=None):
    """ball_pack_wrapper for b = lists x group clouds in ONE launch -> a list of BallPack, one per `group` consecutive clouds
    (views into shared buffers), each exactly what ball_pack_wrapper returns for idx[l*group:(l+1)*group]."""
    _chk(torch.int32, idx); _chk(torch.float32, xyz, new_xyz)
    b, m, ns = idx.shape
    if group < 1 or b % group:
        raise ValueError("ball_pack_groups: %d clouds do not split into lists of %d" % (b, group))
    lists, cap = b // group, (m * ns + 63) // 64
    L = group * cap
    rowinfo = torch.empty((lists, L * 64), dtype=torch.int32, device=idx.device)
    rowdxyz = torch.empty((lists, L * 64, 4), dtype=torch.float32, device=idx.device)
    tilecloud = torch.empty((lists, L), dtype=torch.int32, device=idx.device)
    if hdr is not None:                    # (lists, 4) i32 that IS ZERO already: see ball_pack_wrapper
        _chk(torch.int32, hdr)
        if tuple(hdr.shape) != (lists, 4):
            raise ValueError("ball_pack_groups: hdr must be (lists, 4)")
        _lib.call("prcnn_ball_pack_ex", b, group, xyz.size(1), m, ns, idx.data_ptr(), None, None, None, xyz.data_ptr(), new_xyz.data_ptr(),
                  rowinfo.data_ptr(), rowdxyz.data_ptr(), tilecloud.data_ptr(), hdr.data_ptr(), 1, _lib.current_stream(idx))
    else:
        hdr = torch.empty((lists, 4), dtype=torch.int32, device=idx.device)
        _lib.call("prcnn_ball_pack_groups", b, group, xyz.size(1), m, ns, idx.data_ptr(), None, xyz.data_ptr(), new_xyz.data_ptr(),
                  rowinfo.data_ptr(), rowdxyz.data_ptr(), tilecloud.data_ptr(), hdr.data_ptr(), _lib.current_stream(idx))
    out = []
    for l in range(lists):
        pk = BallPack()
        pk.idx, pk.limit, pk.rep, pk.crep = idx[l * group:(l + 1) * group], None, None, None
        pk.rowinfo, pk.rowdxyz, pk.tilecloud, pk.hdr, pk.max_tiles = rowinfo[l], rowdxyz[l], tilecloud[l], hdr[l], L
        out.append(pk)
    return out


def sa_packed_mlp_wrapper(new_xyz, xyz, P, wxyz, pack, w2t, b2, w3t, b3, out, out_col, zeroed=False):
    """sa_mlp_fused_wrapper over the distinct rows only (csrc/sa_packed.hip): same arguments with the BallPack of the
    index tensor instead of the index tensor; bit-identical results."""
    _chk(torch.float32, new_xyz, xyz, P, wxyz, w2t, b2, w3t, b3, out)
    b, n, c1 = P.shape
    if c1 != 128 or w2t.shape != (128, 128) or w3t.size(0) != 128:
        raise RuntimeError("pointnet2_cuda: sa_packed_mlp needs 128-wide (zero-padded) layers 1 and 2")
    _lib.call("prcnn_sa_packed_mlp", b, n, new_xyz.size(1), w3t.size(1), pack.max_tiles,
              P.data_ptr(), wxyz.data_ptr(), pack.rowinfo.data_ptr(), pack.rowdxyz.data_ptr(), _lib.ptr(pack.tilecloud), pack.hdr.data_ptr(),
              w2t.data_ptr(), b2.data_ptr(), w3t.data_ptr(), b3.data_ptr(), out.data_ptr(), out.size(-1), out_col, int(zeroed),
              _lib.current_stream(xyz))
    return out


def sa_packed_mlp_batch_wrapper(problems):
    """sa_packed_mlp_wrapper for the (up to two) 128-wide scales of one MSG level in ONE launch (round 5):
    [(new_xyz, xyz, P, wxyz, pack, w2t, b2, w3t, b3, out, out_col, zeroed[, (c1, c2)]), ...]; same results as one call per problem.
    (c1, c2): the real widths of layers 1 and 2 under the zero padding -- 64-64 and 64-96 (RPN SA2) skip the padding's MFMAs, same bits."""
    arr = (_lib.SaProblem * len(problems))()
    for q, prob in zip(arr, problems):
        new_xyz, xyz, P, wxyz, pack, w2t, b2, w3t, b3, out, out_col, zeroed = prob[:12]
        q.c1, q.c2 = (int(v) for v in prob[12]) if len(prob) > 12 and prob[12] is not None else (0, 0)
        _chk(torch.float32, new_xyz, xyz, wxyz, w2t, b2, w3t, b3, out)
        b, n, c1 = P.shape
        if c1 == 64:
            # a 64-column slice of a (b, n, 128) tensor that holds the per-point parts of BOTH scales side by side: the narrow
            # kernel reads 64 columns of a 128-float row, the padded one would read its neighbour's
            if not (P.is_cuda and P.dtype == torch.float32 and P.stride() == (n * 128, 128, 1) and P.data_ptr() % 16 == 0):
                raise RuntimeError("pointnet2_cuda: sa_packed_mlp_batch: a 64-wide P must be a column slice of a contiguous (b, n, 128) tensor")
            if not (len(problems) == 2 and all(len(pp) > 12 and pp[12] is not None and pp[12][0] == 64 and pp[12][1] in (64, 96) for pp in problems)):
                raise RuntimeError("pointnet2_cuda: sa_packed_mlp_batch: a 64-wide P needs two problems of real widths 64-64 / 64-96")
        else:
            _chk(torch.float32, P)
        if c1 not in (64, 128) or w2t.shape != (128, 128) or tuple(w3t.shape) != (128, 128):
            raise RuntimeError("pointnet2_cuda: sa_packed_mlp_batch takes 128-128-128 (zero-padded) problems")
        q.b, q.n, q.m, q.c3, q.max_tiles = b, n, new_xyz.size(1), 128, pack.max_tiles
        q.P, q.wxyz, q.rowinfo, q.rowdxyz = P.data_ptr(), wxyz.data_ptr(), pack.rowinfo.data_ptr(), pack.rowdxyz.data_ptr()
        q.tilecloud, q.hdr = pack.tilecloud.data_ptr(), pack.hdr.data_ptr()
        q.w2t, q.b2, q.w3t, q.b3 = w2t.data_ptr(), b2.data_ptr(), w3t.data_ptr(), b3.data_ptr()
        q.out, q.out_stride, q.out_col, q.out_is_zero = out.data_ptr(), out.size(-1), int(out_col), int(bool(zeroed))
    _lib.call("prcnn_sa_packed_mlp_batch", len(problems), arr, _lib.current_stream(problems[0][1]))
    return [p[9] for p in problems]


def packed_gather_affine_wrapper(new_xyz, xyz, P, wxyz, pack, out):
    """Layer 1 over a packed row list: out (pack.max_tiles*64, c1) = relu(P[point] + wxyz.(xyz[point] - centre))."""
    _chk(torch.float32, new_xyz, xyz, P, wxyz, out)
    b, n, c1 = P.shape
    _lib.call("prcnn_packed_gather_affine", b, n, c1, pack.max_tiles,
              P.data_ptr(), wxyz.data_ptr(), pack.rowinfo.data_ptr(), pack.rowdxyz.data_ptr(), pack.tilecloud.data_ptr(), pack.hdr.data_ptr(),
              out.data_ptr(), _lib.current_stream(xyz))
    return out


def packed_layer_wrapper(a, wt, bias, relu, out, pack=None):
    """out = act(a @ wt + bias)[:, :out.size(1)], a (R, K) / wt (K, N) / out (R, <= N) with K, N multiples of 128 (csrc/packed_layer.hip).
    pack given: only the first pack.hdr[0]*64 rows (a packed row list, count on the device); else all R rows."""
    _chk(torch.float32, wt, bias)
    for t in (a, out):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError("pointnet2_cuda: packed_layer expects 2-D float32 CUDA matrices with unit column stride")
    R, K = a.shape
    N = wt.size(1)
    n_store = out.size(1)                     # <= N: a narrow last layer whose weights were zero-padded to N
    if wt.size(0) != K or out.size(0) != R or n_store > N:
        raise RuntimeError("pointnet2_cuda: packed_layer shape mismatch")
    if pack is not None and pack.tilecloud is None:
        raise RuntimeError("pointnet2_cuda: packed_layer does not take a list whose rows carry their cloud (its tile count is not in hdr[0])")
    _lib.call("prcnn_packed_layer", None if pack is None else pack.hdr.data_ptr(), R, 0 if pack is None else pack.max_tiles,
              K, N, n_store, a.data_ptr(), a.stride(0), wt.data_ptr(), bias.data_ptr(), int(bool(relu)), out.data_ptr(), out.stride(0),
              _lib.current_stream(a))
    return out


_SPLIT_W = {}


def rows_layer_bf16x3_wrapper(a, wt, bias, relu, out):
    """EXPERIMENT (numerics switch PRCNN_SPLIT_BF16, csrc/split_bf16.hip): out = act(a @ wt + bias)[:, :out.size(1)] on the bf16 matrix
    cores with every operand split into three bf16 pieces (six products per k-step, f32 accumulation): ~1e-7 relative to the f32 fma
    chain of packed_layer_wrapper, NOT its bits.  The split weights are cached per weight tensor."""
    _chk(torch.float32, wt, bias)
    R, K = a.shape
    N = wt.size(1)
    key = (wt.data_ptr(), wt._version, K, N, str(wt.device))
    ws = _SPLIT_W.get(key)
    if ws is None:
        ws = torch.empty(((K // 16) * (N // 32) * 3 * 64 * 16,), dtype=torch.uint8, device=wt.device)     # three 16-byte pieces per lane, k-step and column block
        _lib.call("prcnn_split_weights_bf16x3", K, N, wt.data_ptr(), ws.data_ptr(), _lib.current_stream(wt))
        _SPLIT_W[key] = ws
    _lib.call("prcnn_rows_layer_bf16x3", R, K, N, out.size(1), a.data_ptr(), a.stride(0), ws.data_ptr(), bias.data_ptr(), int(bool(relu)),
              out.data_ptr(), out.stride(0), _lib.current_stream(a))
    return out


def packed_layer_interp_wrapper(a, wt, bias, relu, out, G, idx, weight):
    """out = act((a @ wt + bias) + interp3(G)): the first layer of a feature-propagation module with the interpolation behind the
    layer's linear part (prcnn_packed_layer_interp).  a (B*n, K) skip features, wt (K, N), G (B, m, N) = coarse features @ the
    interpolated columns of the layer, idx / weight (B, n, 3) from three_nn."""
    _chk(torch.float32, wt, bias, G, weight)
    _chk(torch.int32, idx)
    for t in (a, out):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError("pointnet2_cuda: packed_layer_interp expects 2-D float32 CUDA matrices with unit column stride")
    R, K = a.shape
    N = wt.size(1)
    B, n = idx.shape[0], idx.shape[1]
    if wt.size(0) != K or tuple(out.shape) != (R, N) or R != B * n or G.shape[0] != B or G.shape[2] != N or tuple(weight.shape) != tuple(idx.shape):
        raise RuntimeError("pointnet2_cuda: packed_layer_interp shape mismatch")
    _lib.call("prcnn_packed_layer_interp", R, K, N, a.data_ptr(), a.stride(0), wt.data_ptr(), bias.data_ptr(), int(bool(relu)),
              out.data_ptr(), out.stride(0), n, G.shape[1], G.data_ptr(), G.stride(1), idx.data_ptr(), weight.data_ptr(),
              _lib.current_stream(a))
    return out


def sa_wide_fused_supported(c1, c2, c3):
    return bool(_lib.load().prcnn_sa_wide_fused_supported(int(c1), int(c2), int(c3)))


def sa_wide_fused_wrapper(new_xyz, xyz, P, wxyz, pack, w2t, b2, w3t, b3, out, out_col, zeroed=False):
    """One scale of a wide SA level over a packed row list in ONE kernel (csrc/sa_wide.hip): gather + affine, layer 2, layer 3 +
    max pool -- packed_gather_affine_wrapper -> packed_layer_wrapper -> packed_layer_segmax_wrapper, bit for bit."""
    _chk(torch.float32, new_xyz, xyz, P, wxyz, w2t, b2, w3t, b3, out)
    b, n, c1 = P.shape
    if w2t.size(0) != c1 or w3t.size(0) != w2t.size(1) or wxyz.size(1) != c1:
        raise RuntimeError("pointnet2_cuda: sa_wide_fused shape mismatch")
    _lib.call("prcnn_sa_wide_fused", b, n, new_xyz.size(1), c1, w2t.size(1), w3t.size(1), pack.max_tiles, P.data_ptr(), wxyz.data_ptr(),
              pack.rowinfo.data_ptr(), pack.rowdxyz.data_ptr(), pack.tilecloud.data_ptr(), pack.hdr.data_ptr(), w2t.data_ptr(),
              b2.data_ptr(), w3t.data_ptr(), b3.data_ptr(), out.data_ptr(), out.size(-1), out_col, int(bool(zeroed)),
              _lib.current_stream(xyz))
    return out


def sa_wide_fused3_supported(c0, c1, c2, c3):
    return bool(_lib.load().prcnn_sa_wide_fused3_supported(int(c0), int(c1), int(c2), int(c3)))


def sa_wide_fused3_wrapper(new_xyz, xyz, feats, wcat, b1, wxyz, pack, b2, b3, widths, out, out_col, zeroed=False):
    """sa_wide_fused_wrapper with layer 1 inside (csrc/sa_wide3.hip): feats (b, n, c0) point features, wcat = w1 | w2 | w3 (k-major, one
    tensor), widths = (c0, c1, c2, c3) -- packed_layer_wrapper (P = feats w1 + b1) -> sa_wide_fused_wrapper, bit for bit, for a level
    that groups every point once."""
    _chk(torch.float32, new_xyz, xyz, feats, wcat, b1, wxyz, b2, b3, out)
    b, n, c0 = feats.shape
    c0w, c1, c2, c3 = (int(v) for v in widths)
    if c0 != c0w or wcat.numel() != c0 * c1 + c1 * c2 + c2 * c3 or wxyz.size(1) != c1 or b1.numel() != c1 or b2.numel() != c2 or b3.numel() != c3:
        raise RuntimeError("pointnet2_cuda: sa_wide_fused3 shape mismatch")
    _lib.call("prcnn_sa_wide_fused3", b, n, new_xyz.size(1), c0, c1, c2, c3, pack.max_tiles, feats.data_ptr(), wxyz.data_ptr(),
              pack.rowinfo.data_ptr(), pack.rowdxyz.data_ptr(), _lib.ptr(pack.tilecloud), pack.hdr.data_ptr(), wcat.data_ptr(),
              b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), out.data_ptr(), out.size(-1), out_col, int(bool(zeroed)),
              _lib.current_stream(xyz))
    return out


def packed_gather_affine_batch_wrapper(problems):
    """packed_gather_affine_wrapper for up to 4 independent problems [(new_xyz, xyz, P, wxyz, pack, out), ...] -- the scales of one
    MSG level -- in ONE launch."""
    arr = (_lib.GatherProblem * len(problems))()
    for q, (new_xyz, xyz, P, wxyz, pack, out) in zip(arr, problems):
        _chk(torch.float32, new_xyz, xyz, P, wxyz, out)
        q.b, q.n, q.c1 = P.shape
        q.max_tiles = pack.max_tiles
        q.P, q.wxyz, q.out = P.data_ptr(), wxyz.data_ptr(), out.data_ptr()
        q.rowinfo, q.rowdxyz, q.tilecloud, q.hdr = pack.rowinfo.data_ptr(), pack.rowdxyz.data_ptr(), pack.tilecloud.data_ptr(), pack.hdr.data_ptr()
    _lib.call("prcnn_packed_gather_affine_batch", len(problems), arr, _lib.current_stream(problems[0][2]))
    return [p[5] for p in problems]


def packed_layer_batch_wrapper(problems):
    """packed_layer_wrapper for up to 4 independent problems [(a, wt, bias, relu, out, pack-or-None), ...] in ONE launch (those
    that do not land on the same kernel are launched one by one by the library)."""
    arr = (_lib.LayerProblem * len(problems))()
    for q, (a, wt, bias, relu, out, pack) in zip(arr, problems):
        _chk(torch.float32, wt, bias)
        if a.dim() != 2 or out.dim() != 2 or a.stride(1) != 1 or out.stride(1) != 1 or not a.is_cuda:
            raise RuntimeError("pointnet2_cuda: packed_layer expects 2-D float32 CUDA matrices with unit column stride")
        q.K, q.N = wt.shape
        q.n_store = out.size(1)
        q.A, q.lda, q.W, q.bias, q.relu = a.data_ptr(), a.stride(0), wt.data_ptr(), bias.data_ptr(), int(bool(relu))
        q.out, q.ldo = out.data_ptr(), out.stride(0)
        q.hdr, q.rows, q.max_tiles = (None, a.size(0), 0) if pack is None else (pack.hdr.data_ptr(), 0, pack.max_tiles)
    _lib.call("prcnn_packed_layer_batch", len(problems), arr, 0, _lib.current_stream(problems[0][0]))
    return [p[4] for p in problems]


def packed_layer_segmax_batch_wrapper(problems):
    """packed_layer_segmax_wrapper for up to 4 problems [(a, wt, bias, pack, b, m, out, out_col, zeroed), ...] in ONE launch."""
    arr = (_lib.LayerProblem * len(problems))()
    for q, (a, wt, bias, pack, b, m, out, out_col, zeroed) in zip(arr, problems):
        _chk(torch.float32, a, wt, bias, out)
        q.K, q.N = wt.shape
        q.n_store = q.N
        q.A, q.lda, q.W, q.bias, q.relu = a.data_ptr(), a.stride(0), wt.data_ptr(), bias.data_ptr(), 1
        q.out, q.ldo, q.out_col, q.out_is_zero = out.data_ptr(), out.size(-1), out_col, int(bool(zeroed))
        q.b, q.m = b, m
        q.hdr, q.max_tiles, q.rowinfo, q.tilecloud = pack.hdr.data_ptr(), pack.max_tiles, pack.rowinfo.data_ptr(), pack.tilecloud.data_ptr()
    _lib.call("prcnn_packed_layer_batch", len(problems), arr, 1, _lib.current_stream(problems[0][0]))
    return [p[6] for p in problems]


def rows_dot_wrapper(a, wt, bias, out):
    """out (R, n) = a (R, K) @ wt (K, n) + bias for n <= 4 (a classification head's last layer) -- csrc/packed_layer.hip."""
    _chk(torch.float32, wt, bias)
    if a.dim() != 2 or a.stride(1) != 1 or out.stride(1) != 1 or not a.is_cuda:
        raise RuntimeError("pointnet2_cuda: rows_dot expects 2-D float32 CUDA matrices with unit column stride")
    _lib.call("prcnn_rows_dot", a.size(0), a.size(1), wt.size(1), a.data_ptr(), a.stride(0), wt.data_ptr(), bias.data_ptr(),
              out.data_ptr(), out.stride(0), _lib.current_stream(a))
    return out


def rpn_tail_wrapper(known, idx, weight, wcat, bcat, wc2, bc2, feats, cls, reg):
    """Finest FP module + both RPN heads in one kernel (csrc/rpn_tail.hip): known (b,m,256), idx / weight (b,n,3) ->
    feats (b,n,128), cls (b,n,1), reg (b,n,n_reg)."""
    _chk(torch.float32, known, weight, wcat, bcat, wc2, bc2, feats, cls, reg); _chk(torch.int32, idx)
    b, m, c = known.shape
    if c != 256 or tuple(wcat.shape) != (768, 128) or tuple(bcat.shape) != (5, 128) or wc2.numel() != 128 or feats.size(-1) != 128:
        raise RuntimeError("pointnet2_cuda: rpn_tail is written for 256 interpolated channels and 128-wide layers")
    _lib.call("prcnn_rpn_tail", b, idx.size(1), m, known.data_ptr(), idx.data_ptr(), weight.data_ptr(), wcat.data_ptr(),
              bcat.data_ptr(), wc2.data_ptr(), bc2.data_ptr(), reg.size(-1), feats.data_ptr(), cls.data_ptr(), reg.data_ptr(),
              _lib.current_stream(known))
    return feats, cls, reg


def rpn_tail_lin_wrapper(G, idx, weight, wcat, bcat, wc2, bc2, feats, cls, reg):
    """rpn_tail_wrapper with the FP module's first layer already applied at the coarse level: G (b,m,128) = coarse features @ W1 (no
    bias), wcat (512,128) = [FP layer 2 | cls 1 | reg 1 | reg 2], bcat (5,128) as in rpn_tail_wrapper (prcnn_rpn_tail_lin)."""
    _chk(torch.float32, G, weight, wcat, bcat, wc2, bc2, feats, cls, reg); _chk(torch.int32, idx)
    b, m, c = G.shape
    if c != 128 or tuple(wcat.shape) != (512, 128) or tuple(bcat.shape) != (5, 128) or wc2.numel() != 128 or feats.size(-1) != 128:
        raise RuntimeError("pointnet2_cuda: rpn_tail_lin is written for a 128-wide coarse product and 128-wide layers")
    _lib.call("prcnn_rpn_tail_lin", b, idx.size(1), m, G.data_ptr(), idx.data_ptr(), weight.data_ptr(), wcat.data_ptr(),
              bcat.data_ptr(), wc2.data_ptr(), bc2.data_ptr(), reg.size(-1), feats.data_ptr(), cls.data_ptr(), reg.data_ptr(),
              _lib.current_stream(G))
    return feats, cls, reg


def rpn_tail_boxes_supported(channels, loc_scope, loc_bin_size, num_head_bin, xz_fine):
    """does prcnn_rpn_tail_lin_boxes serve this regression layout? (the 76 channels of the shipped configurations)"""
    return bool(_lib.call("prcnn_rpn_tail_boxes_supported", int(channels), float(loc_scope), float(loc_bin_size), int(num_head_bin),
                          int(bool(xz_fine))))


def rpn_tail_lin_boxes_wrapper(G, idx, weight, wcat, bcat, wc2, bc2, n_reg, loc_scope, loc_bin_size, num_head_bin, xz_fine, anchor_size,
                               xyz, feats, cls, boxes):
    """rpn_tail_lin_wrapper with the proposal layer's decode inside (round 5): boxes (b,n,7) = decode_bbox_target(xyz, reg) with
    y += h / 2 (bbox_transform.py:24-121, proposal_layer.py:23-31); the (b,n,n_reg) regression rows are never stored.
    anchor_size: 3 python floats (h, w, l); xyz (b,n,3)."""
    _chk(torch.float32, G, weight, wcat, bcat, wc2, bc2, feats, cls, boxes, xyz); _chk(torch.int32, idx)
    b, m, c = G.shape
    if c != 128 or tuple(wcat.shape) != (512, 128) or tuple(bcat.shape) != (5, 128) or wc2.numel() != 128 or feats.size(-1) != 128:
        raise RuntimeError("pointnet2_cuda: rpn_tail_lin is written for a 128-wide coarse product and 128-wide layers")
    if boxes.size(-1) != 7 or boxes.numel() != 7 * b * idx.size(1) or xyz.numel() != 3 * b * idx.size(1):
        raise RuntimeError("pointnet2_cuda: rpn_tail_lin_boxes wants xyz (b,n,3) and boxes (b,n,7)")
    anchor = (ctypes.c_float * 3)(*[float(v) for v in anchor_size])
    _lib.call("prcnn_rpn_tail_lin_boxes", b, idx.size(1), m, G.data_ptr(), idx.data_ptr(), weight.data_ptr(), wcat.data_ptr(),
              bcat.data_ptr(), wc2.data_ptr(), bc2.data_ptr(), int(n_reg), float(loc_scope), float(loc_bin_size), int(num_head_bin),
              int(bool(xz_fine)), ctypes.cast(anchor, ctypes.c_void_p), xyz.data_ptr(), feats.data_ptr(), cls.data_ptr(),
              boxes.data_ptr(), _lib.current_stream(G))
    return feats, cls, boxes


def selftest_fmod_two_pi(a):
    """-> (the fused decode's branch-free fmod(a, 2 pi), the device library's fmodf(a, 2 pi)) for a float32 tensor"""
    _chk(torch.float32, a)
    mine, lib = torch.empty_like(a), torch.empty_like(a)
    _lib.call("prcnn_selftest_fmod_two_pi", a.numel(), a.data_ptr(), mine.data_ptr(), lib.data_ptr(), _lib.current_stream(a))
    return mine, lib


def packed_layer_segmax_wrapper(a, wt, bias, pack, b, m, out, out_col, zeroed=False):
    """Last layer of a level + max pool over a packed row list: out (b,m,stride)[..., out_col:out_col+N]."""
    _chk(torch.float32, a, wt, bias, out)
    K, N = wt.shape
    _lib.call("prcnn_packed_layer_segmax", b, m, pack.max_tiles, K, N, a.data_ptr(), a.stride(0), wt.data_ptr(), bias.data_ptr(),
              pack.rowinfo.data_ptr(), pack.tilecloud.data_ptr(), pack.hdr.data_ptr(), out.data_ptr(), out.size(-1), out_col, int(zeroed),
              _lib.current_stream(a))
    return out


def sa_xyz_mlp_supported(c1, c2, c3, nsample):
    return bool(_lib.load().prcnn_sa_xyz_mlp_supported(int(c1), int(c2), int(c3), int(nsample)))


def sa_xyz_mlp_wrapper(new_xyz, xyz, idx, w1, b1, w2, b2, w3, b3, out, out_col):
    """Coordinates-only SA scale in one VALU kernel (csrc/sa_xyz_mlp.hip): gather -> 3 layers -> max.
    xyz (b,n,3), new_xyz (b,m,3), idx (b,m,ns), w1 (>=3,c1), w2 (c1,c2), w3 (c2,c3) k-major, out (b,m,stride)."""
    _chk(torch.float32, new_xyz, xyz, w1, b1, w2, b2, w3, b3, out); _chk(torch.int32, idx)
    b, n, _ = xyz.shape
    _lib.call("prcnn_sa_xyz_mlp", b, n, idx.size(1), idx.size(2), w1.size(1), w2.size(1), w3.size(1),
              new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
              b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), out.data_ptr(), out.size(-1), out_col,
              _lib.current_stream(xyz))
    return out


def pooled_tiles_wrapper(cnt, rows_per_cloud):
    """cnt (clouds) i32 distinct rows per pooled cloud -> (tilemap i32, hdr i32[4]): the 64-row tiles that hold them."""
    _chk(torch.int32, cnt)
    clouds = cnt.numel()
    tilemap = torch.empty((max(1, clouds * (rows_per_cloud // 64)),), dtype=torch.int32, device=cnt.device)
    hdr = torch.empty((4,), dtype=torch.int32, device=cnt.device)
    _lib.call("prcnn_pooled_tiles", clouds, rows_per_cloud, cnt.data_ptr(), tilemap.data_ptr(), hdr.data_ptr(), _lib.current_stream(cnt))
    return tilemap, hdr


def pooled_rows_wrapper(cnt, rows_per_cloud, hdr=None):
    """cnt (clouds) i32 distinct rows per pooled cloud -> (rowmap i32, hdr i32[4]): those rows of ALL clouds back to back, hdr[1] of
    them (prcnn_pooled_rows), for rcnn_point_mlp_rows_wrapper.  hdr (4) i32, optional: a header that IS ZERO already."""
    _chk(torch.int32, cnt)
    clouds = cnt.numel()
    rowmap = torch.empty((max(1, clouds * rows_per_cloud),), dtype=torch.int32, device=cnt.device)
    zero = hdr is not None
    if zero:
        _chk(torch.int32, hdr)
        if hdr.numel() != 4:
            raise ValueError("pooled_rows: hdr must hold 4 int32")
    else:
        hdr = torch.empty((4,), dtype=torch.int32, device=cnt.device)
    _lib.call("prcnn_pooled_rows", clouds, rows_per_cloud, cnt.data_ptr(), rowmap.data_ptr(), hdr.data_ptr(), int(zero), _lib.current_stream(cnt))
    return rowmap, hdr


def rcnn_point_mlp_rows_wrapper(rows, fcol, wu1, bu1, wu2, bu2, wm, bm, wp, bp, p, rowlist):
    """rcnn_point_mlp_wrapper's fused form (only p) over the row list of pooled_rows_wrapper: p[r] for the listed rows r, the others are
    left as they are (prcnn_rcnn_point_mlp_rows)."""
    _chk(torch.float32, rows, wu1, bu1, wu2, bu2, wm, bm, wp, bp, p)
    rowmap, hdr = rowlist
    _chk(torch.int32, rowmap, hdr)
    _lib.call("prcnn_rcnn_point_mlp_rows", rows.size(0), rows.size(1), int(fcol), rows.data_ptr(), wu1.data_ptr(), bu1.data_ptr(),
              wu2.data_ptr(), bu2.data_ptr(), wm.data_ptr(), bm.data_ptr(), wp.data_ptr(), bp.data_ptr(), p.data_ptr(),
              rowmap.data_ptr(), hdr.data_ptr(), _lib.current_stream(rows))
    return p


def sa_xyz_mlp_packed_wrapper(new_xyz, xyz, pack, w1, b1, w2, b2, w3, b3, out, out_col, zeroed=False):
    """sa_xyz_mlp_wrapper over the distinct rows of the level's index tensor (BallPack) -- bit-identical results."""
    _chk(torch.float32, w1, b1, w2, b2, w3, b3, out)
    _lib.call("prcnn_sa_xyz_mlp_packed", new_xyz.size(0), new_xyz.size(1), w1.size(1), w2.size(1), w3.size(1), pack.max_tiles,
              pack.rowinfo.data_ptr(), pack.rowdxyz.data_ptr(), pack.tilecloud.data_ptr(), pack.hdr.data_ptr(), w1.data_ptr(),
              b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), out.data_ptr(), out.size(-1), out_col,
              int(zeroed), _lib.current_stream(out))
    return out


def rcnn_point_mlp_wrapper(rows, fcol, wu1, bu1, wu2, bu2, wm, bm, wp, bp, xfeat, merged, p, tiles=None):
    """RCNN entrance chain as tiled MFMA layer kernels (csrc/rcnn_point_mlp.hip): rows (R, ld) pooled rows
    [x',y',z',mask,depth,0,0,0 | 128 feats at column fcol] -> xfeat = xyz_up(in5), merged = relu([xfeat | feats] wm + bm),
    p = merged wp + bp, each (R,128).  tiles = (tilemap, hdr) from pooled_tiles_wrapper: only those 64-row tiles are computed."""
    _chk(torch.float32, rows, wu1, bu1, wu2, bu2, wm, bm, wp, bp, p)
    if (xfeat is None) != (merged is None):
        raise RuntimeError("pointnet2_cuda: rcnn_point_mlp takes xfeat and merged together or neither")
    if xfeat is not None:
        _chk(torch.float32, xfeat, merged)
    # xfeat = merged = None: only p is wanted -> the whole chain in ONE kernel, a 64-row tile never leaves LDS
    _lib.call("prcnn_rcnn_point_mlp", rows.size(0), rows.size(1), int(fcol), rows.data_ptr(), wu1.data_ptr(), bu1.data_ptr(),
              wu2.data_ptr(), bu2.data_ptr(), wm.data_ptr(), bm.data_ptr(), wp.data_ptr(), bp.data_ptr(), _lib.ptr(xfeat),
              _lib.ptr(merged), p.data_ptr(), None if tiles is None else tiles[0].data_ptr(),
              None if tiles is None else tiles[1].data_ptr(), _lib.current_stream(rows))
    return p


def rows_gemm128_rows_wrapper(a, wt, bias, relu, out, rowlist):
    """rows_gemm128_wrapper (K = 128) over a LIST of rows: out[r] = act(a[r] @ wt + bias) for r = rowmap[0 .. hdr[1]); the other rows of
    `out` are left as they are (prcnn_rows_gemm128_rows)."""
    if a.dim() != 2 or a.stride(1) != 1 or not a.is_cuda or a.dtype != torch.float32 or a.shape[1] != 128:
        raise RuntimeError("pointnet2_cuda: rows_gemm128_rows expects a 2-D float32 CUDA matrix of 128 columns with unit column stride")
    _chk(torch.float32, wt, bias, out)
    rowmap, hdr = rowlist
    _chk(torch.int32, rowmap, hdr)
    _lib.call("prcnn_rows_gemm128_rows", a.shape[0], a.data_ptr(), a.stride(0), 0, wt.data_ptr(), bias.data_ptr(), int(bool(relu)), out.data_ptr(),
              rowmap.data_ptr(), hdr.data_ptr(), _lib.current_stream(a))
    return out


def rows_gemm128_wrapper(a, wt, bias, relu, out=None):
    """a (R, K) with K in (128, 256), row stride a.stride(0), unit column stride -> act(a @ wt + bias) (R, 128) on the tiled
    MFMA layer kernel (csrc/rcnn_point_mlp.hip); R % 64 == 0, wt (K,128) k-major."""
    if a.dim() != 2 or a.stride(1) != 1 or not a.is_cuda or a.dtype != torch.float32:
        raise RuntimeError("pointnet2_cuda: rows_gemm128 expects a 2-D float32 CUDA matrix with unit column stride")
    _chk(torch.float32, wt, bias)
    R, K = a.shape
    if out is None:
        out = torch.empty((R, 128), dtype=torch.float32, device=a.device)
    ld = a.stride(0)
    off1 = a.data_ptr() + 128 * 4
    _lib.call("prcnn_rows_gemm128", R, K // 128, a.data_ptr(), ld, 0, off1 if K == 256 else None, ld, 0, wt.data_ptr(),
              bias.data_ptr(), int(bool(relu)), out.data_ptr(), _lib.current_stream(a))
    return out
