"""MI355X-native (gfx950) implementation of the PointRCNN eval_rcnn hot path of
cxy1997/3D_adapt_auto_driving: PointNet++ set-abstraction ops, RoI point pooling, BEV / rotated
NMS and the rotated-IoU evaluator kernel, behind the reference's own extension-module API.

The directory name starts with a digit, so import it with
``importlib.import_module("3d_adapt_auto_driving_amd")``.
"""
import os as _os

PACKAGE_DIR = _os.path.dirname(_os.path.abspath(__file__))
DROPIN_DIR = _os.path.join(PACKAGE_DIR, "dropin")                  # pointnet2_cuda / iou3d_cuda / roipool3d_cuda as Python modules over ctypes
# the same three names as COMPILED extension modules (pybind11 over the C ABI, csrc/bindings/; built by __graft_entry__.build()):
# the reference's own entry points only, with its bindings' signatures -- put this directory on sys.path instead of DROPIN_DIR
NATIVE_DROPIN_DIR = _os.path.join(PACKAGE_DIR, "dropin_native")
__version__ = "0.1.0"
