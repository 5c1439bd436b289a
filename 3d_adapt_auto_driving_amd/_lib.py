"""ctypes binding of libprcnn_hip.so -- the C ABI declared in include/prcnn_hip.h.

There is NO fallback: if the shared library is missing or an entry point fails, an exception is
raised.  Nothing in this package imports the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprcnn_hip.so")

_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_D = C.c_double
_L = C.c_longlong

class GatherProblem(C.Structure):
    """prcnn_gather_problem (include/prcnn_hip.h)"""
    _fields_ = [("b", _I), ("n", _I), ("c1", _I), ("max_tiles", C.c_long), ("P", _P), ("wxyz", _P), ("rowinfo", _P), ("rowdxyz", _P),
                ("tilecloud", _P), ("hdr", _P), ("out", _P)]


class LayerProblem(C.Structure):
    """prcnn_layer_problem (include/prcnn_hip.h)"""
    _fields_ = [("hdr", _P), ("rows", C.c_long), ("max_tiles", C.c_long), ("K", _I), ("N", _I), ("n_store", _I), ("A", _P), ("lda", C.c_long),
                ("W", _P), ("bias", _P), ("relu", _I), ("out", _P), ("ldo", C.c_long), ("b", _I), ("m", _I), ("rowinfo", _P), ("tilecloud", _P),
                ("out_col", _I), ("out_is_zero", _I)]


class SaProblem(C.Structure):
    """prcnn_sa_problem (include/prcnn_hip.h)"""
    _fields_ = [("b", _I), ("n", _I), ("m", _I), ("c3", _I), ("max_tiles", C.c_long), ("P", _P), ("wxyz", _P), ("rowinfo", _P), ("rowdxyz", _P),
                ("tilecloud", _P), ("hdr", _P), ("w2t", _P), ("b2", _P), ("w3t", _P), ("b3", _P), ("out", _P), ("out_stride", _I),
                ("out_col", _I), ("out_is_zero", _I), ("c1", _I), ("c2", _I)]


# name -> argument types (return type is always int except where noted)
SIGNATURES = {
    "prcnn_version": [],
    "prcnn_opt_n_threads": [_I],
    "prcnn_set_ball_query_mode": [_I],
    "prcnn_ball_query": [_I, _I, _I, _F, _I, _P, _P, _P, _P],
    "prcnn_fps_new_xyz": [_I, _I, _I, _P, _P, _P, _P],
    "prcnn_ball_query_limit": [_I, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    "prcnn_group_points": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "prcnn_group_points_grad": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "prcnn_gather_points": [_I, _I, _I, _I, _P, _P, _P, _P],
    "prcnn_gather_points_grad": [_I, _I, _I, _I, _P, _P, _P, _P],
    "prcnn_furthest_point_sampling": [_I, _I, _I, _P, _P, _P, _P],
    "prcnn_set_fps_arithmetic": [_I],
    "prcnn_ball_query_full": [_I, _I, _I, _F, _I, _P, _P, _P, _P],
    "prcnn_point_aux": [_L, _F, _P, _P, _P, _P, _P, _P],
    "prcnn_three_nn": [_I, _I, _I, _P, _P, _P, _P, _P],
    "prcnn_three_nn_weights": [_I, _I, _I, _P, _P, _P, _P, _P],
    "prcnn_three_interpolate": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "prcnn_three_interpolate_grad": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "prcnn_query_and_group": [_I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P],
    "prcnn_bias_relu_inplace": [C.c_long, _I, C.c_long, _P, _P, _P],
    "prcnn_maxpool_bias_relu": [_I, _I, _I, _I, _P, _P, _P, _P],
    "prcnn_group_cat_pm": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "prcnn_gather_affine_relu_pm": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_sa_mlp_fused": [_I] * 7 + [_P] * 10 + [_I, _I, _P],
    "prcnn_ball_pack": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_ball_pack_ex": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "prcnn_rcnn_postprocess_blobs": [_I, _I, _I, _F, _F, _I, _I, _F, _F, _P, _F, _F, _P, _P, _P, _P, _P, _I, _P],
    "prcnn_ball_pack_groups": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_ball_pack_rep": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_dup_rep": [_I, _I, _I, _P, _P, _P, _P, _P],
    "prcnn_rcnn_roi_geometry": [_I, _I, _I, _F, _I, _I, _F, _I] + [_P] * 9,
    "prcnn_rcnn_roi_geometry_packs": [_I, _I, _I, _F, _I, _I, _F, _I] + [_P] * 21 + [_I, _P],
    "prcnn_rows_gemm128_rows": [_L, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P],
    "prcnn_sa_packed_mlp": [_I, _I, _I, _I, C.c_long] + [_P] * 11 + [_I, _I, _I, _P],
    "prcnn_packed_gather_affine": [_I, _I, _I, C.c_long] + [_P] * 7 + [_P],
    "prcnn_packed_layer": [_P, C.c_long, C.c_long, _I, _I, _I, _P, C.c_long, _P, _P, _I, _P, C.c_long, _P],
    "prcnn_split_weights_bf16x3": [_I, _I, _P, _P, _P],
    "prcnn_rows_layer_bf16x3": [C.c_long, _I, _I, _I, _P, C.c_long, _P, _P, _I, _P, C.c_long, _P],
    "prcnn_packed_layer_interp": [C.c_long, _I, _I, _P, C.c_long, _P, _P, _I, _P, C.c_long, _I, _I, _P, C.c_long, _P, _P, _P],
    "prcnn_sa_xyz_mlp_packed": [_I, _I, _I, _I, _I, C.c_long] + [_P] * 11 + [_I, _I, _I, _P],
    "prcnn_rows_dot": [C.c_long, _I, _I, _P, C.c_long, _P, _P, _P, C.c_long, _P],
    "prcnn_sa_wide_fused_supported": [_I, _I, _I],
    "prcnn_sa_wide_fused": [_I, _I, _I, _I, _I, _I, C.c_long] + [_P] * 11 + [_I, _I, _I, _P],
    "prcnn_sa_wide_fused3_supported": [_I, _I, _I, _I],
    "prcnn_sa_wide_fused3": [_I, _I, _I, _I, _I, _I, _I, C.c_long] + [_P] * 11 + [_I, _I, _I, _P],
    "prcnn_sa_packed_mlp_batch": [_I, C.POINTER(SaProblem), _P],
    "prcnn_packed_gather_affine_batch": [_I, C.POINTER(GatherProblem), _P],
    "prcnn_packed_layer_batch": [_I, C.POINTER(LayerProblem), _I, _P],
    "prcnn_rpn_tail": [_I, _I, _I] + [_P] * 7 + [_I] + [_P] * 4,
    "prcnn_rpn_tail_lin": [_I, _I, _I] + [_P] * 7 + [_I] + [_P] * 4,
    "prcnn_rpn_tail_boxes_supported": [_I, _F, _F, _I, _I],
    "prcnn_rpn_tail_lin_boxes": [_I, _I, _I] + [_P] * 7 + [_I, _F, _F, _I, _I] + [_P] * 6,
    "prcnn_selftest_fmod_two_pi": [_L, _P, _P, _P, _P],
    "prcnn_rpn_proposals_boxes": [_I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    "prcnn_packed_layer_segmax": [_I, _I, C.c_long, _I, _I, _P, C.c_long, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "prcnn_maxpool_pm": [C.c_long, _I, _I, _P, _P, _I, _I, _P],
    "prcnn_three_interpolate_pm": [_I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P],
    "prcnn_three_interpolate_cat_pm": [_I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    "prcnn_boxes_overlap_bev": [_I, _P, _I, _P, _P, _P],
    "prcnn_boxes_iou_bev": [_I, _P, _I, _P, _P, _P],
    "prcnn_nms": [_I, _P, _P, _F, _P],
    "prcnn_nms_normal": [_I, _P, _P, _F, _P],
    "prcnn_nms_device": [_I, _I, _P, _P, _F, _I, _I, _P, _P, _P],
    "prcnn_rows_gemm128": [_L, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P],
    "prcnn_rcnn_point_mlp": [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_pooled_tiles": [_I, _I, _P, _P, _P, _P],
    "prcnn_pooled_rows": [_I, _I, _P, _P, _P, _I, _P],
    "prcnn_rcnn_point_mlp_rows": [_L, _I, _I] + [_P] * 13,
    "prcnn_sa_xyz_mlp_supported": [_I, _I, _I, _I],
    "prcnn_sa_xyz_mlp": [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "prcnn_rpn_proposals": [_I, _I, _I, _F, _F, _I, _I, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P],
    "prcnn_rcnn_postprocess": [_I, _I, _I, _F, _F, _I, _I, _F, _F, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_roipool3d": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "prcnn_host_pts_in_boxes3d": [_I, _I, _P, _P, _P],
    "prcnn_host_roipool3d": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "prcnn_roipool3d_canonical": [_I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_roipool3d_canonical_xyz": [_I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "prcnn_point_groups": [_I, _I, _P, _P, _P, _P],
    "prcnn_input_stage": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _I, _P, _P, _P, _P, _P],
    "prcnn_valid_flags": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P],
    "prcnn_rotate_iou_eval": [_I, _I, _P, _P, _P, _I, _P],
    "prcnn_rotate_iou_eval_segmented": [_I, _L, _P, _P, _P, _P, _P, _P, _I, _P],
    "prcnn_kitti_image_stats": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _D, _D, _I, _I, _P, _P, _P, _P],
    "prcnn_kitti_collect_scores": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _D, _P, _P],
    "prcnn_kitti_accumulate_pr": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _D, _P, _I, _I, _P],
}

_lib = None


class PrcnnError(RuntimeError):
    pass


def load():
    """Load libprcnn_hip.so (once).  Raises if it has not been built: build it with
    ``python __graft_entry__.py`` or ``make -C 3d_adapt_auto_driving_amd/csrc``."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PrcnnError("%s not found: the HIP extension is not built (no CPU fallback exists)" % LIB_PATH)
        # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so); the streams and device
        # pointers we are handed belong to THAT runtime, so it must be the one this library binds
        # to: import torch first so its copy is already mapped when the loader resolves ours.
        import torch  # noqa: F401
        _hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(_hip):
            C.CDLL(_hip, mode=C.RTLD_GLOBAL)
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.argtypes = argtypes
            fn.restype = _I
        lib.prcnn_last_error.restype = C.c_char_p
        lib.prcnn_last_error.argtypes = []
        _lib = lib
    return _lib


def last_error():
    return load().prcnn_last_error().decode("utf-8", "replace")


_fn_cache = {}


def call(name, *args):
    """Call an entry point; a negative return code raises PrcnnError with the library's message."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args)
    if rc < 0:
        raise PrcnnError("%s failed (%d): %s" % (name, rc, last_error()))
    return rc


def has_entry(ext, name):
    """Does operator backend `ext` offer entry point `name`?  The HIP drop-in modules (IS_HIP_EXTENSION) must offer every
    entry the engine uses: a missing one raises instead of silently selecting a slower formulation.  Only the CPU
    stand-ins of the test suite may lack fused entries (the torch-op formulations are their checked equivalents)."""
    if getattr(ext, "IS_HIP_EXTENSION", False):
        if not hasattr(ext, name):
            raise PrcnnError("HIP extension module %s lost its entry point %r" % (getattr(ext, "__name__", ext), name))
        return True
    return hasattr(ext, name)


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def current_stream(t):
    """hipStream_t of torch's current stream on the tensor's device, as an int (the raw-handle query: the Stream-object
    route costs ~4 us per call, ~0.5 ms per step of the engine)."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        idx = t.device.index
        return _raw_stream(0 if idx is None else idx) if t.is_cuda else 0
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream
