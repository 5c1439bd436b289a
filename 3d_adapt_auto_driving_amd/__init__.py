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


# ---- hipGraph replay and the HIP runtime's "graph packet capture" ---------------------------------------------------------------
# ROCm 7.2's runtime pre-builds the AQL packets of an instantiated graph (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default).  On this
# stack those packets go bad once a few thousand kernels have been launched eagerly on the DEFAULT stream after the instantiation: the
# next replay of ANY graph of the process dies with a memory access fault (profiles/r03_hipgraph_notes.md: 5000 x `x.add_(1)` on the
# null stream between two replays reproduce it; the same launches on another stream do not; with the switch at 0 nothing happens).
# The switch is read when the HIP runtime initialises, so it is set here unless the runtime is already up -- in which case graph
# replay is declared unsafe and eval_rcnn.make_runner() hands out the eager runner instead (same results, 1.2 ms of host time per step).
GRAPH_REPLAY_SAFE = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in _os.environ:           # (an explicit setting of the user is left alone)
    import sys as _sys
    _torch = _sys.modules.get("torch")
    if _torch is None or not _torch.cuda.is_initialized():
        _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
        GRAPH_REPLAY_SAFE = True
if _os.environ.get("PRCNN_GRAPHS_FORCE") == "1":                  # profiles/graph_fault_probe.py: replay although it is unsafe
    GRAPH_REPLAY_SAFE = True
