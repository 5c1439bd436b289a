"""-m gpu: every A/B switch of 3d_adapt_auto_driving_amd/switches.py (kind "ab": "SAME results, bit for bit, either way") is RUN in its
non-default form and held to that promise ("each is a path the default bench never runs").  A child process per switch
(they are read once, at import / first call) runs six batches of 4 scenes -- uniform and LiDAR-shaped alternating -- through the eager
pipelined runner and one batch through the serial engine; RoIs, head outputs, final boxes, scores and counts must equal the
default run's exactly.  docs/SWITCHES.md points here for every "ab" row."""
import concurrent.futures
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "ab_switch_child.py")

# the non-default value of a switch: "unset" -> "1", "1" -> "0", "0" -> "1"; the few numeric ones by hand
ALT = {"PRCNN_EARLY_LEVELS": "2", "PRCNN_EARLY_FP": "1"}


def alternatives():
    SW = pkg("switches")
    out = []
    for name, (kind, default, _where, _what) in sorted(SW.SWITCHES.items()):
        if kind != "ab":
            continue
        out.append((name, ALT.get(name, {"unset": "1", "1": "0", "0": "1"}.get(default))))
    assert all(v is not None for _, v in out), [n for n, v in out if v is None]
    return out


def run_child(path, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["PRCNN_GRAPHS"] = "0"                                   # the eager runner: same kernels, streams and order, no capture time per child
    r = subprocess.run([sys.executable, CHILD, path], env=env, capture_output=True, text=True, timeout=900)
    return r.returncode, (r.stdout + r.stderr)[-2000:]


# A/B switches whose other form shares its bits with the default only under a given setting of a NUMERICS switch: the layer-by-layer
# RPN tail applies the finest FP module in the reference's association (interpolate, then the layer), which is what the fused tail
# does under PRCNN_NO_FP_LINEAR=1 (csrc/rpn_tail.hip rpn_tail_kernel: "arithmetic of the separate kernels, bit for bit")
UNDER = {"PRCNN_NO_RPN_TAIL": {"PRCNN_NO_FP_LINEAR": "1"}}


def test_every_ab_switch_gives_the_same_detections(tmp_path):
    alts = alternatives()
    assert len(alts) == 20                                        # round 6: 43 -> 20 A/B switches
    wants = {}
    for tag, extra in [("default", {})] + [(n, e) for n, e in UNDER.items()]:
        base = str(tmp_path / ("base_%s.npz" % tag))
        rc, log = run_child(base, extra)
        assert rc == 0, log
        wants[tag] = dict(np.load(base))
    want0 = wants["default"]
    assert sum(int(want0["r%d_num" % i].sum()) for i in range(6)) > 50
    failures = []

    def one(item):
        name, value = item
        path = str(tmp_path / (name + ".npz"))
        env = dict(UNDER.get(name, {}))
        env[name] = value
        rc, log = run_child(path, env)
        if rc != 0:
            return "%s=%s: child failed\n%s" % (name, value, log)
        got = np.load(path)
        want = wants[name if name in UNDER else "default"]
        bad = [k for k in want if not np.array_equal(want[k], got[k])]
        return ("%s=%s: differs in %s" % (name, value, bad[:6])) if bad else None

    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as pool:          # children share the GPU: a few at a time
        for res in pool.map(one, alts):
            if res:
                failures.append(res)
    assert not failures, "\n".join(failures)


def test_numerics_switches_stay_inside_the_box_tolerance(tmp_path):
    """kind "numerics": another association of a sum or a library GEMM -- results move, inside BASELINE's 1e-4 on boxes: same
    detection counts, every final box and score within 1e-4 of the default run's (PRCNN_ALLOW_LIB_GEMM only permits, it selects nothing)."""
    SW = pkg("switches")
    names = sorted(n for n, v in SW.SWITCHES.items() if v[0] == "numerics" and n != "PRCNN_ALLOW_LIB_GEMM")
    assert "PRCNN_TAIL_NARROW" in names and "PRCNN_NO_FP_LINEAR" in names and "PRCNN_NO_PACK" in names
    base = str(tmp_path / "default.npz")
    rc, log = run_child(base, {})
    assert rc == 0, log
    want = dict(np.load(base))

    def one(name):
        default = SW.SWITCHES[name][1]
        value = {"unset": "1", "1": "0", "0": "1"}[default]
        path = str(tmp_path / (name + ".npz"))
        env = {name: value}
        if name in ("PRCNN_LIB_GEMM", "PRCNN_ROWS_GEMM", "PRCNN_NO_PACK"):
            env["PRCNN_ALLOW_LIB_GEMM"] = "1"
        rc, log = run_child(path, env)
        if rc != 0:
            return "%s=%s: child failed\n%s" % (name, value, log)
        got = np.load(path)
        msgs = []
        for i in list(range(6)) + ["serial"]:
            pre = ("r%d_" % i) if i != "serial" else "serial_"
            if not np.array_equal(want[pre + "num"], got[pre + "num"]):
                msgs.append("%snum differs" % pre)
                continue
            for k in ("boxes", "scores"):
                d = float(np.abs(want[pre + k] - got[pre + k]).max())
                if d > 1e-4:
                    msgs.append("%s%s max |d| %.3g" % (pre, k, d))
        return ("%s=%s: %s" % (name, value, "; ".join(msgs[:6]))) if msgs else None

    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as pool:
        failures = [r for r in pool.map(one, names) if r]
    assert not failures, "\n".join(failures)
