"""IN-CONTAINER ONLY: import the reference's Python model (read-only, from /root/reference) on
CPU with the oracle as its operator backend.  Used by make_golden.py to produce the committed
fixtures and by tests that are skipped when /root/reference is absent (i.e. on the GPU box).

Shims (none copies reference code):
  1. ``easydict.EasyDict`` stand-in (attribute dict)          -- package absent from the image
  2. ``yaml.load`` defaults to FullLoader                      -- config.py:188 predates the arg
  3. ``torch.cuda.FloatTensor/IntTensor`` -> CPU constructors  -- pointnet2_utils.py:25.. allocate with them
  4. ``Tensor.cuda`` / ``Module.cuda`` -> identity
  5. ``Tensor.get_device`` -> the tensor's device              -- bbox_transform.py:40
  6. ``sys.modules['pointnet2_cuda'|'iou3d_cuda'|'roipool3d_cuda']`` -> oracle-backed modules
"""
import os
import sys
import types

REF = "/root/reference/pointrcnn"


def available():
    return os.path.isdir(REF)


def install():
    import torch
    import yaml
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import ext_cpu

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            d = dict(d or {}, **kw)
            for k, v in d.items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            elif isinstance(v, (list, tuple)):
                v = type(v)(EasyDict(x) if isinstance(x, dict) else x for x in v)
            dict.__setitem__(self, k, v)

        __setitem__ = __setattr__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules.setdefault("easydict", ed)

    if not getattr(yaml, "_prcnn_patched", False):
        _load = yaml.load
        yaml.load = lambda stream, Loader=yaml.FullLoader: _load(stream, Loader=Loader)
        yaml._prcnn_patched = True

    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.IntTensor = torch.IntTensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: self.device

    for name, cls in (("pointnet2_cuda", ext_cpu.pointnet2_cpu), ("iou3d_cuda", ext_cpu.iou3d_cpu),
                      ("roipool3d_cuda", ext_cpu.roipool3d_cpu)):
        mod = types.ModuleType(name)
        for attr in dir(cls):
            if not attr.startswith("_"):
                setattr(mod, attr, getattr(cls, attr))
        sys.modules[name] = mod

    for p in (REF, os.path.join(REF, "lib", "net"), os.path.join(REF, "pointnet2_lib", "pointnet2")):
        if p not in sys.path:
            sys.path.append(p)


def reference_model(yaml_overrides=None, eval_mode="rcnn"):
    """Build the reference PointRCNN (mode='TEST') under cfgs/default.yaml (+ overrides given as a
    dict merged through the reference's own cfg_from_list-style assignment)."""
    install()
    import torch
    from lib.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(REF, "tools", "cfgs", "default.yaml"))
    if eval_mode == "rcnn":
        cfg.RCNN.ENABLED = True
        cfg.RPN.ENABLED = cfg.RPN.FIXED = True

    def assign(node, d):
        for k, v in d.items():
            if isinstance(v, dict):
                assign(node[k], v)
            else:
                node[k] = v
    if yaml_overrides:
        assign(cfg, yaml_overrides)
    from lib.net.point_rcnn import PointRCNN
    model = PointRCNN(num_classes=2, use_xyz=True, mode="TEST")
    model.eval()
    return model, cfg
