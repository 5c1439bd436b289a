"""-m gpu: BASELINE configs[2] at its FULL shapes (cfgs/default.yaml: 16384 points, 4096/1024/256/64 backbone levels, 100 RoIs x 512
points per scene) against fixtures recorded from the REFERENCE PointRCNN itself (tests/golden/make_golden.py g12: the reference's
point_rcnn.py:26-70, rcnn_net.py:127-185, proposal_layer.py:15-119 run under the shims of tests/golden/ref_harness.py), B = 2,
on uniform scenes (g12u) and on LiDAR-shaped sweeps (g12l).  Bar (BASELINE.json north_star): every RoI, every head output and every
final box within 1e-4 IN THE REFERENCE'S ORDER (RoIs: up to swaps inside groups whose reference RPN scores agree within the
tolerance -- the reference's own order of such RoIs depends on its convolution library; final boxes: exactly the reference's order),
equal counts -- for the nn.Module graph over the HIP operators (the drop-in path)
AND for the point-major engine (the product path).  Replaces the 1e-3 / >= 90 % bars that round 3 held against a
self-comparison.

On a miss the test names the first stage where the GPU and the reference part (RPN scores -> foreground mask -> RoIs -> pooled
clouds -> sampled centres -> heads -> decoded boxes -> final keep lists)."""
import numpy as np
import pytest
import torch

import helpers
from conftest import pkg
from test_host_logic import full_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(kind, use_engine):
    E, F = pkg("eval_rcnn"), pkg("net.fast_infer")
    model, cfg, g, pts = full_model(DEV, kind)
    x = torch.from_numpy(pts).to(DEV)
    with torch.no_grad():
        ret = F.FastPointRCNN(model, cfg)(x) if use_engine else model({"pts_input": x})
        if "seg_result" not in ret:
            ret["seg_result"] = (torch.sigmoid(ret["rpn_cls"][..., 0]) > cfg.RPN.SCORE_THRESH).float()
        det = E.postprocess(cfg, ret, x.shape[0])
    torch.cuda.synchronize()
    rep = helpers.e2e_report(ret, det, g)
    text = helpers.e2e_text(rep)
    print("g12%s %s:\n%s" % (kind, "engine" if use_engine else "module path", text))
    return rep, text, g


@pytest.mark.parametrize("kind", ["u", "l"])
@pytest.mark.parametrize("use_engine", [False, True], ids=["module", "engine"])
def test_full_size_vs_reference_model_fixture_every_roi_and_box(kind, use_engine):
    rep, text, g = _run(kind, use_engine)
    first_bad = next((r for r in rep if r[2] > 0), None)
    assert first_bad is None, "first stage that parts from the reference: %s\n%s" % (first_bad[0], text)
    assert int(g["final_num"].min()) >= 10


@pytest.mark.parametrize("native", [True, False], ids=["compiled_modules", "ctypes_modules"])
def test_reference_operation_order_over_the_reference_api_only(native):
    """The drop-in path proper (north_star: "models load unmodified"): the nn.Module graph in the REFERENCE'S operation order over
    the reference's 17 entry points only -- separate ball_query / group_points x 2 / subtract / cat, Conv2d + BatchNorm + ReLU
    modules, max_pool2d, per-scene proposal layer and final stage over the blocking nms_gpu / nms_normal_gpu
    (eval_rcnn.reference_api_only) -- against the fixture recorded from the reference model at default.yaml shapes: every RoI,
    head output and final box within 1e-4."""
    E = pkg("eval_rcnn")
    model, cfg, g, pts = full_model(DEV, "u")
    x = torch.from_numpy(pts).to(DEV)
    with E.reference_api_only(native=native), torch.no_grad():
        pu = pkg("pointnet2.pointnet2_utils")
        assert not hasattr(pu.pointnet2, "query_and_group_wrapper") and not hasattr(pkg("iou3d_utils").iou3d_cuda, "nms_device")
        if native:
            assert pu.pointnet2.__file__.endswith(".so")
        ret = model({"pts_input": x})
        ret["seg_result"] = (torch.sigmoid(ret["rpn_cls"][..., 0]) > cfg.RPN.SCORE_THRESH).float()
        det = E.postprocess(cfg, ret, x.shape[0])
    torch.cuda.synchronize()
    assert hasattr(pkg("pointnet2.pointnet2_utils").pointnet2, "query_and_group_wrapper")     # restored
    rep = helpers.e2e_report(ret, det, g)
    text = helpers.e2e_text(rep)
    print("g12u reference order, %s:\n%s" % ("compiled modules" if native else "ctypes modules", text))
    assert all(r[2] == 0 for r in rep), text


def test_configs2_literal_batch_through_the_product_runner():
    """VERDICT r4 'missing 2': BASELINE configs[2] at its LITERAL batch through the PRODUCT runner against reference-made data.
    Fixture g12p = the reference PointRCNN on two batches of B = 8 (eight uniform scenes, eight LiDAR-shaped sweeps).  The same 16
    scenes go through ``eval_rcnn.make_runner()`` = GraphedRunner with graph replay on, pairs of batches per launch (16 scenes /
    1600 RoIs per stage launch: what bench.py times) -- twice, so that the second pass is pure REPLAY of the captured graphs -- and
    every RoI, head output, decoded and final box must lie within 1e-4 of the reference's, counts equal, in the reference's order
    (RoIs up to swaps inside groups whose reference scores agree within the tolerance; the number of swaps is printed)."""
    E = pkg("eval_rcnn")
    model, cfg, g, batches = full_model(DEV, "p")
    xs = [torch.from_numpy(b).to(DEV) for b in batches]
    runner = E.make_runner(model, cfg, DEV)
    assert type(runner).__name__ == "GraphedRunner" and runner.pair == 2, "the product runner is the graphed one with pairs of batches"
    keys = ("rois", "rcnn_cls", "rcnn_reg", "boxes", "scores", "num", "pred_boxes3d")
    for rep_no in range(2):
        dets = []
        for i, x in enumerate(xs):
            d = runner.submit(x, xs[i + 1:])
            if d is not None:
                dets.append(d)
        dets += runner.drain()
        assert len(dets) == 2
        got = []
        for d in dets:
            d["ready"].synchronize()
            got.append({k: d[k].clone() for k in keys})
        torch.cuda.synchronize()
        if rep_no == 0:
            captures = runner.captures
    assert captures == runner.captures and captures > 0                  # nothing was captured behind the first pass: pass 2 replayed
    ret = {k: torch.cat([b[k] for b in got], 0) for k in ("rois", "rcnn_cls", "rcnn_reg")}
    det = {k: torch.cat([b[k] for b in got], 0) for k in ("boxes", "scores", "num", "pred_boxes3d")}
    # what the stages in front of the RoIs left in the member's slot (pass 2 ran in the slot before the next one to be written)
    m = runner.slots[(runner._next_slot - 1) % runner.n_slots]["members"][0]
    for k in ("rpn_cls", "rpn_reg", "seg_result"):
        if m["st"].get(k) is not None:
            ret[k] = m["st"][k]
    if m["tl"].get("roi_scores") is not None:
        ret["roi_scores_raw"] = m["tl"]["roi_scores"]
    rep = helpers.e2e_report(ret, det, g)
    text = helpers.e2e_text(rep)
    print("g12p through GraphedRunner (pairs, replay):\n" + text)
    first_bad = next((r for r in rep if r[2] > 0), None)
    assert first_bad is None, "first stage that parts from the reference: %s\n%s" % (first_bad[0], text)
    assert g["rois"].shape[0] == 16 and int(g["final_num"].min()) >= 5


@pytest.mark.parametrize("how", ["engine", "module", "runner"])
def test_double_yaml_32768_points_vs_reference_model_fixture(how):
    """tools/cfgs/double.yaml:39 (NUM_POINTS 32768; VERDICT r4 'missing 3'): fixture g13 = the REFERENCE PointRCNN on a 32768-point
    scene (B = 1).  Round 5 makes this configuration first class -- sampling on two workgroups per cloud (fps_spec2_kernel), the
    fused proposal path and the spatial groups of the RoI pooling up to 65536 points -- and holds it to the bar of default.yaml:
    every RoI, head output, decoded and final box within 1e-4 of the reference's, counts equal, for the point-major engine, the
    nn.Module graph over the HIP operators, and the product runner (graph replay; the batch twice so that a pair forms)."""
    E, F = pkg("eval_rcnn"), pkg("net.fast_infer")
    model, cfg, g, pts = full_model(DEV, "d")
    assert cfg.RPN.NUM_POINTS == 32768 and pts.shape == (1, 32768, 3)
    x = torch.from_numpy(pts).to(DEV)
    with torch.no_grad():
        if how == "runner":
            runner = E.make_runner(model, cfg, DEV)
            dets = [d for d in (runner.submit(x, [x]), runner.submit(x, [])) if d is not None] + runner.drain()
            assert len(dets) == 2
            for d in dets:
                d["ready"].synchronize()
            assert all(torch.equal(dets[0][k], dets[1][k]) for k in ("rois", "boxes", "scores", "num"))
            ret = {k: dets[1][k].clone() for k in ("rois", "rcnn_cls", "rcnn_reg")}
            det = {k: dets[1][k].clone() for k in ("boxes", "scores", "num", "pred_boxes3d")}
        else:
            ret = F.FastPointRCNN(model, cfg)(x) if how == "engine" else model({"pts_input": x})
            if "seg_result" not in ret:
                ret["seg_result"] = (torch.sigmoid(ret["rpn_cls"][..., 0]) > cfg.RPN.SCORE_THRESH).float()
            det = E.postprocess(cfg, ret, 1)
    torch.cuda.synchronize()
    rep = helpers.e2e_report(ret, det, g)
    text = helpers.e2e_text(rep)
    print("g13 (double.yaml, 32768 points) %s:\n%s" % (how, text))
    first_bad = next((r for r in rep if r[2] > 0), None)
    assert first_bad is None, "first stage that parts from the reference: %s\n%s" % (first_bad[0], text)
    assert int(g["final_num"].min()) >= 10
