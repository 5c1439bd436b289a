"""MI355X-native (gfx950) implementation of the PointRCNN eval_rcnn hot path of
cxy1997/3D_adapt_auto_driving: PointNet++ set-abstraction ops, RoI point pooling, BEV / rotated
NMS and the rotated-IoU evaluator kernel, behind the reference's own extension-module API.

The directory name starts with a digit, so import it with
``importlib.import_module("3d_adapt_auto_driving_amd")``.
"""
import os as _os

PACKAGE_DIR = _os.path.dirname(_os.path.abspath(__file__))
DROPIN_DIR = _os.path.join(PACKAGE_DIR, "dropin")
__version__ = "0.1.0"
