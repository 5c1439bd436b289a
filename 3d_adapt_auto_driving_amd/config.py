"""Configuration tree of the PointRCNN path.

Same key names as the reference's global ``cfg`` (pointrcnn/lib/config.py:5-181) so that code
reading ``cfg.RPN.SA_CONFIG.NPOINTS`` etc. works unchanged, but an ordinary object that is
passed around explicitly; ``make_cfg()`` returns a fresh tree holding the library defaults and
``apply_eval_defaults()`` overlays the values the reference evaluates with
(pointrcnn/tools/cfgs/default.yaml + the eval_mode 'rcnn' switches of eval_rcnn.py:883-887).
Only keys that the inference path reads are kept; training-only keys are accepted on merge and
stored, never interpreted.
"""
import copy
from ast import literal_eval

import numpy as np


class Node(dict):
    """dict with attribute access (cfg.RPN.LOC_SCOPE and cfg['TEST'] both work)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Node({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _tree(d):
    return Node({k: _tree(v) if isinstance(v, dict) else v for k, v in d.items()})


_LIBRARY_DEFAULTS = {
    "TAG": "default",
    "CLASSES": "Car",
    "INCLUDE_SIMILAR_TYPE": False,
    "PC_REDUCE_BY_RANGE": True,
    "PC_AREA_SCOPE": np.array([[-40, 40], [-1, 3], [0, 70.4]], dtype=np.float64),
    "CLS_MEAN_SIZE": np.array([[1.52, 1.63, 3.88]], dtype=np.float32),
    "RPN": {
        "ENABLED": True, "FIXED": False, "USE_INTENSITY": True,
        "LOC_XZ_FINE": False, "LOC_SCOPE": 3.0, "LOC_BIN_SIZE": 0.5, "NUM_HEAD_BIN": 12,
        "BACKBONE": "pointnet2_msg", "USE_BN": True, "NUM_POINTS": 16384,
        "SA_CONFIG": {
            "NPOINTS": [4096, 1024, 256, 64],
            "RADIUS": [[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]],
            "NSAMPLE": [[16, 32], [16, 32], [16, 32], [16, 32]],
            "MLPS": [[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]],
                     [[128, 196, 256], [128, 196, 256]], [[256, 256, 512], [256, 384, 512]]],
        },
        "FP_MLPS": [[128, 128], [256, 256], [512, 512], [512, 512]],
        "CLS_FC": [128], "REG_FC": [128], "DP_RATIO": 0.5,
        "LOSS_CLS": "DiceLoss", "NMS_TYPE": "normal", "SCORE_THRESH": 0.3,
    },
    "RCNN": {
        "ENABLED": False, "USE_RPN_FEATURES": True, "USE_MASK": True, "MASK_TYPE": "seg",
        "USE_INTENSITY": False, "USE_DEPTH": True, "USE_SEG_SCORE": False, "ROI_SAMPLE_JIT": False,
        "POOL_EXTRA_WIDTH": 1.0,
        "LOC_SCOPE": 1.5, "LOC_BIN_SIZE": 0.5, "NUM_HEAD_BIN": 9, "LOC_Y_BY_BIN": False,
        "LOC_Y_SCOPE": 0.5, "LOC_Y_BIN_SIZE": 0.25, "SIZE_RES_ON_ROI": False,
        "USE_BN": False, "DP_RATIO": 0.0, "BACKBONE": "pointnet", "XYZ_UP_LAYER": [128, 128],
        "NUM_POINTS": 512,
        "SA_CONFIG": {"NPOINTS": [128, 32, -1], "RADIUS": [0.2, 0.4, 100], "NSAMPLE": [64, 64, 64],
                      "MLPS": [[128, 128, 128], [128, 128, 256], [256, 256, 512]]},
        "CLS_FC": [256, 256], "REG_FC": [256, 256],
        "LOSS_CLS": "BinaryCrossEntropy", "SCORE_THRESH": 0.3, "NMS_THRESH": 0.1,
    },
    "TRAIN": {"SPLIT": "train", "VAL_SPLIT": "smallval", "RPN_PRE_NMS_TOP_N": 12000,
              "RPN_POST_NMS_TOP_N": 2048, "RPN_NMS_THRESH": 0.85, "RPN_DISTANCE_BASED_PROPOSE": True},
    "TEST": {"SPLIT": "val", "RPN_PRE_NMS_TOP_N": 9000, "RPN_POST_NMS_TOP_N": 300,
             "RPN_NMS_THRESH": 0.7, "RPN_DISTANCE_BASED_PROPOSE": True},
}

# values of tools/cfgs/default.yaml that differ from the library defaults and matter at inference
_EVAL_OVERLAY = {
    "INCLUDE_SIMILAR_TYPE": True,
    "CLS_MEAN_SIZE": [[1.52563191462, 1.62856739989, 3.88311640418]],
    "RPN": {"USE_INTENSITY": False, "LOC_XZ_FINE": True, "LOSS_CLS": "SigmoidFocalLoss"},
    "RCNN": {"ENABLED": True, "ROI_SAMPLE_JIT": True},
    "TRAIN": {"RPN_PRE_NMS_TOP_N": 9000, "RPN_POST_NMS_TOP_N": 512},
    "TEST": {"RPN_POST_NMS_TOP_N": 100, "RPN_NMS_THRESH": 0.8},
}


def make_cfg():
    return _tree(copy.deepcopy(_LIBRARY_DEFAULTS))


def merge_into(src, dst, strict=False, path=""):
    """Overlay mapping ``src`` on tree ``dst`` (config.py:193-220 semantics: lists given for
    ndarray-typed keys are converted; type clashes raise).  Unknown keys raise only if ``strict``;
    otherwise they are stored (training-only keys of the yaml files)."""
    for k, v in src.items():
        here = path + k
        if k not in dst:
            if strict:
                raise KeyError("%s is not a valid config key" % here)
            dst[k] = _tree(v) if isinstance(v, dict) else v
            continue
        old = dst[k]
        if isinstance(old, dict):
            if not isinstance(v, dict):
                raise ValueError("config key %s: expected a mapping" % here)
            merge_into(v, old, strict, here + ".")
        elif isinstance(old, np.ndarray):
            dst[k] = np.array(v, dtype=old.dtype)
        elif type(old) is not type(v) and not (isinstance(old, float) and isinstance(v, int)):
            raise ValueError("Type mismatch (%s vs. %s) for config key: %s" % (type(old), type(v), here))
        else:
            dst[k] = v
    return dst


def apply_eval_defaults(cfg, eval_mode="rcnn"):
    merge_into(_EVAL_OVERLAY, cfg)
    if eval_mode == "rcnn":      # eval_rcnn.py:883-887
        cfg.RCNN.ENABLED = True
        cfg.RPN.ENABLED = cfg.RPN.FIXED = True
    elif eval_mode == "rpn":     # eval_rcnn.py:879-882
        cfg.RPN.ENABLED, cfg.RCNN.ENABLED = True, False
    else:
        raise ValueError("unsupported eval_mode %r" % eval_mode)
    return cfg


def cfg_from_file(cfg, filename):
    """Merge a reference-style yaml (tools/cfgs/*.yaml) into ``cfg``."""
    import yaml
    with open(filename, "r") as f:
        data = yaml.safe_load(f)
    return merge_into(data or {}, cfg)


def cfg_from_list(cfg, kv):
    """``--set K V K V ...`` overrides (config.py:223-242)."""
    if len(kv) % 2:
        raise ValueError("--set needs KEY VALUE pairs")
    for key, raw in zip(kv[0::2], kv[1::2]):
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        try:
            val = literal_eval(raw)
        except (ValueError, SyntaxError):
            val = raw
        if parts[-1] not in node:
            raise KeyError(key)
        if type(val) is not type(node[parts[-1]]):
            raise ValueError("type %s does not match original type %s" % (type(val), type(node[parts[-1]])))
        node[parts[-1]] = val
    return cfg


def default_eval_cfg():
    """The configuration eval_rcnn.py --cfg_file cfgs/default.yaml --eval_mode rcnn runs with."""
    return apply_eval_defaults(make_cfg(), "rcnn")
