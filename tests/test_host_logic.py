"""CPU tests of the host-side Python (glue + model) against fixtures produced by the REFERENCE's
own Python (tests/golden/make_golden.py) and of the C-ABI surface (load + exported symbols)."""
import importlib
import os
import re

import numpy as np
import pytest
import torch

from conftest import pkg, ROOT

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_capi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "prcnn_hip.h")).read()
    declared = set(re.findall(r"\b(prcnn_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = pkg("_lib").load()                       # loads without a GPU; no compute call here
    for name in declared:
        assert hasattr(lib, name), "libprcnn_hip.so does not export %s" % name
    assert declared - {"prcnn_last_error"} == set(pkg("_lib").SIGNATURES)
    assert lib.prcnn_version() >= 100
    assert lib.prcnn_opt_n_threads(1000) == 512


def test_capi_rejects_bad_arguments_without_gpu():
    L = pkg("_lib")
    with pytest.raises(L.PrcnnError, match="null pointer"):
        L.call("prcnn_ball_query", 1, 8, 8, 0.1, 4, None, None, None, None)
    with pytest.raises(L.PrcnnError, match="bad sizes"):
        L.call("prcnn_three_nn", -1, 8, 8, None, None, None, None, None)
    with pytest.raises(L.PrcnnError):
        L.call("prcnn_rotate_iou_eval", 2, 2, None, None, None, 7, None)


def test_no_cpu_fallback_in_product_ops():
    pu = pkg("pointnet2.pointnet2_utils")
    xyz = torch.zeros((1, 16, 3))
    with pytest.raises(RuntimeError, match="CUDA"):
        pu.furthest_point_sample(xyz, 4)
    with pytest.raises(RuntimeError, match="CUDA"):
        pkg("roipool3d_utils").roipool3d_gpu(xyz, torch.zeros((1, 16, 2)), torch.zeros((1, 1, 7)), 1.0, 4)
    # nothing under the package imports the oracle
    pdir = pkg().PACKAGE_DIR
    for dp, _, files in os.walk(pdir):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_host_utilities_match_compiled_reference_fixture():
    """roipool3d_cuda.pts_in_boxes3d_cpu / roipool3d_cpu (the reference module's HOST utilities, host code of the C-ABI
    library) against g5 = outputs of the reference's own compiled roipool3d.cpp on the same inputs: bit for bit; plus the
    Python-level helpers of roipool3d_utils.py:31-108 built on them."""
    g = load("g5_roipool_ref.npz")
    RU, ku = pkg("roipool3d_utils"), pkg("kitti_utils")
    pts, boxes, feat = torch.from_numpy(g["pts"]), torch.from_numpy(g["boxes"]), torch.from_numpy(g["feat"])
    masks = RU.pts_in_boxes3d_cpu(pts, boxes)
    assert len(masks) == boxes.shape[0]
    assert np.array_equal(torch.stack(masks).numpy(), g["pts_flag"] > 0)
    pp, pf, pe = RU.roipool_pc_cpu(pts, feat, boxes, 512)
    assert np.array_equal(pe.numpy(), g["pooled_empty_flag"]) and int(pe.sum()) >= 1
    assert np.array_equal(pp.numpy(), g["pooled_pts"]) and np.array_equal(pf.numpy(), g["pooled_features"])
    # roipool3d_cpu: enlarge + pool + canonical transform, numpy in / out
    extra = np.ascontiguousarray(g["feat"][:, :2])
    rest = np.ascontiguousarray(g["feat"][:, 2:])
    inp, fe = RU.roipool3d_cpu(g["boxes"].copy(), g["pts"], rest, extra, 1.0, sampled_pt_num=64)
    assert inp.shape == (boxes.shape[0], 64, 5) and fe.shape == (boxes.shape[0], 64, rest.shape[1])
    big = ku.enlarge_box3d(g["boxes"], 1.0)
    p2, f2, e2 = RU.roipool_pc_cpu(pts, feat, torch.from_numpy(big), 64)
    k = int(np.nonzero(e2.numpy() == 0)[0][0])
    want = p2[k].numpy() - g["boxes"][k, 0:3]
    want = ku.rotate_pc_along_y(want.copy(), g["boxes"][k, 6] % (2 * np.pi))
    np.testing.assert_allclose(inp[k, :, 0:3], want, atol=1e-6)
    assert np.array_equal(inp[k, :, 3:5], f2[k].numpy()[:, :2]) and np.array_equal(fe[k], f2[k].numpy()[:, 2:])
    with pytest.raises(RuntimeError):
        pkg("dropin.roipool3d_cuda").pts_in_boxes3d_cpu(torch.zeros((1, 4), dtype=torch.int32), pts[:4], boxes[:1])


def test_glue_matches_reference_python():
    g = load("g7_glue_ref.npz")
    bt, ku = pkg("bbox_transform"), pkg("kitti_utils")
    anchor = torch.tensor([1.52563191462, 1.62856739989, 3.88311640418])
    rpn = bt.decode_bbox_target(torch.from_numpy(g["rpn_xyz"]), torch.from_numpy(g["rpn_reg"]), anchor_size=anchor,
                                loc_scope=3.0, loc_bin_size=0.5, num_head_bin=12, get_xz_fine=True,
                                get_y_by_bin=False, get_ry_fine=False)
    assert np.array_equal(rpn.numpy(), g["rpn_boxes"])
    rois = torch.from_numpy(g["rcnn_rois"])
    rc = bt.decode_bbox_target(rois.clone(), torch.from_numpy(g["rcnn_reg"]), anchor_size=anchor, loc_scope=1.5,
                               loc_bin_size=0.5, num_head_bin=9, get_xz_fine=True, get_y_by_bin=False,
                               loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=True)
    assert np.array_equal(rc.numpy(), g["rcnn_boxes"])
    assert np.array_equal(ku.boxes3d_to_bev_torch(rois).numpy(), g["bev"])
    assert np.array_equal(ku.enlarge_box3d(rois, 1.0).numpy(), g["enlarged"])
    assert np.array_equal(ku.enlarge_box3d(rois.numpy(), 1.0), g["enlarged"])
    pc = torch.from_numpy(g["rot_pc_in"].copy())
    assert np.array_equal(ku.rotate_pc_along_y_torch(pc, rois[:, 6]).numpy(), g["rot_pc_out"])
    corners = ku.boxes3d_to_corners3d(g["rcnn_rois"])
    assert np.array_equal(corners, g["corners3d"])
    ib, ic = pkg("synth").SyntheticCalib().corners3d_to_img_boxes(corners)
    np.testing.assert_allclose(ib, g["img_boxes"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(ic, g["img_corners"], rtol=1e-6, atol=1e-4)


def test_engine_with_intensity_matches_reference_fixture_on_cpu(oracle):
    """cfg.RPN.USE_INTENSITY on the point-major engine (round 4; general kernels), oracle operator backend on CPU tensors, against
    the fixture recorded from the REFERENCE model in that configuration (g8i)."""
    from oracle import ext_cpu
    E, F = pkg("eval_rcnn"), pkg("net.fast_infer")
    model, cfg, g = tiny_model(intensity=True)
    with ext_cpu.patch_package():
        eng = F.FastPointRCNN(model, cfg)
        assert eng.in_feat == 1
        det = E.infer_batch(model, cfg, torch.from_numpy(g["pts"]), engine=eng)
        with pytest.raises(ValueError):
            eng.rpn_stage(torch.from_numpy(g["pts"][..., :3].copy()))
    for key, ref in (("rois", "rois"), ("rcnn_cls", "rcnn_cls"), ("rcnn_reg", "rcnn_reg"), ("boxes", "final_boxes"), ("scores", "final_scores")):
        np.testing.assert_allclose(det[key].numpy(), g[ref], rtol=0, atol=2e-5)
    assert np.array_equal(det["num"].numpy(), g["final_num"])
    assert isinstance(E.make_runner(model, cfg, "cpu"), E.EngineRunner)


def test_state_dict_layout_default_cfg():
    cfg = pkg("config").default_eval_cfg()
    model = pkg("eval_rcnn").build_model(cfg, "cpu")
    sd = model.state_dict()
    assert len(sd) == 244 and sum(v.numel() for v in sd.values()) == 3901774
    assert "rpn.backbone_net.SA_modules.0.mlps.0.layer0.bn.bn.running_mean" in sd
    assert "rpn.rpn_cls_layer.2.conv.bias" in sd and "rpn.rpn_cls_layer.1.conv.bias" not in sd
    assert "rcnn_net.cls_layer.3.conv.weight" in sd and "rcnn_net.SA_modules.0.mlps.0.layer0.conv.bias" in sd
    assert sd["rpn.rpn_reg_layer.2.conv.weight"].shape == (76, 128, 1)
    assert sd["rcnn_net.reg_layer.3.conv.weight"].shape == (46, 256, 1)


TINY = {"RPN": {"NUM_POINTS": 2048,
                "SA_CONFIG": {"NPOINTS": [512, 128, 32, 8],
                              "MLPS": [[[8, 8, 16], [8, 8, 16]], [[16, 16, 32], [16, 16, 32]],
                                       [[32, 32, 32], [32, 32, 32]], [[32, 32, 64], [32, 32, 64]]]},
                "FP_MLPS": [[128, 128], [32, 32], [32, 32], [32, 32]], "CLS_FC": [32], "REG_FC": [32]},
        "RCNN": {"XYZ_UP_LAYER": [128, 128], "NUM_POINTS": 128,
                 "SA_CONFIG": {"NPOINTS": [32, 8, -1], "NSAMPLE": [16, 16, 16],
                               "MLPS": [[32, 32, 32], [32, 32, 64], [64, 64, 64]]},
                 "CLS_FC": [32, 32], "REG_FC": [32, 32]},
        "TEST": {"RPN_PRE_NMS_TOP_N": 600, "RPN_POST_NMS_TOP_N": 20}}


def tiny_model(device="cpu", intensity=False):
    """the tiny PointRCNN with the weights of the REFERENCE model the fixture was recorded from; intensity: cfg.RPN.USE_INTENSITY,
    4-channel input (g8i)"""
    C = pkg("config")
    cfg = C.default_eval_cfg()
    C.merge_into(TINY, cfg)
    if intensity:
        C.merge_into({"RPN": {"USE_INTENSITY": True}}, cfg)
    model = pkg("eval_rcnn").build_model(cfg, device)
    g = load("g8i_e2e_tiny_intensity_ref.npz" if intensity else "g8_e2e_tiny_ref.npz")
    sd = {str(k): torch.from_numpy(g["w/" + str(k)]) for k in g["state_keys"]}
    model.load_state_dict(sd)          # strict: the key tree must equal the reference's
    return model, cfg, g


def full_model(device="cpu", kind="u"):
    """default.yaml PointRCNN with the seeded weights of the REFERENCE model that the full-size fixture g12u / g12l was recorded
    from (tests/golden/make_golden.py g12; helpers.seeded_state_dict regenerates the 3.9 M parameters from the seed, the fixture
    holds their checksum and the calibrated RPN classification bias) -> model, cfg, fixture, input batch (2, 16384, 3) numpy
    (kind "p", fixture g12p: a list of two batches (8, 16384, 3))"""
    import helpers
    C, S = pkg("config"), pkg("synth")
    cfg = C.default_eval_cfg()
    model = pkg("eval_rcnn").build_model(cfg, "cpu")
    if kind == "d":       # tools/cfgs/double.yaml:39: NUM_POINTS 32768, everything else default.yaml's (fixture g13, B = 1)
        C.merge_into({"RPN": {"NUM_POINTS": 32768}}, cfg)
        model = pkg("eval_rcnn").build_model(cfg, "cpu")
    g = load("g13_e2e_double_ref.npz" if kind == "d" else "g12%s_e2e_full_ref.npz" % kind)
    sd, checksum = helpers.seeded_state_dict(model.state_dict(), int(g["seed"]))
    assert abs(checksum - float(g["weights_checksum"])) < 1e-6 * checksum, "seeded weights differ from the ones the fixture was made with"
    sd["rpn.rpn_cls_layer.2.conv.bias"] = torch.from_numpy(g["rpn_cls_bias"])
    model.load_state_dict(sd)          # strict: the key tree must equal the reference's
    seed0 = int(g["scene_seed0"])
    if kind == "d":
        pts = np.stack([S.scene(seed0, 32768)], 0)
    elif kind == "p":       # configs[2]'s literal batch, one pair of the graphed runner: 8 uniform scenes, then 8 LiDAR-shaped sweeps
        pts = [np.stack([S.scene(seed0 + i, 16384) for i in range(8)], 0), np.stack([S.lidar_scene(seed0 + 8 + i, 16384) for i in range(8)], 0)]
    else:
        pts = np.stack([(S.scene if kind == "u" else S.lidar_scene)(seed0 + i, 16384) for i in range(2)], 0)
    return model.to(device).eval(), cfg, g, pts


@pytest.mark.parametrize("kind", ["u", "l"])
def test_e2e_full_size_matches_reference_model(oracle, kind):
    """BASELINE configs[2] shapes (default.yaml, N = 16384, 100 RoIs x 512 points, B = 2; uniform and LiDAR-shaped scenes): this
    build's model + the oracle operator backend on CPU, shared MLPs as nn.Modules (the reference's operation order), against
    the fixture recorded from the REFERENCE PointRCNN at the same shapes (g12): every RoI, head output and final box."""
    from oracle import ext_cpu
    fm = pkg("pointnet2.fused_mlp")
    model, cfg, g, pts = full_model("cpu", kind)
    with ext_cpu.patch_package():
        fm.ENABLED = False
        try:
            det = pkg("eval_rcnn").infer_batch(model, cfg, torch.from_numpy(pts))
        finally:
            fm.ENABLED = True
    assert np.array_equal(det["rois"].numpy(), g["rois"])
    np.testing.assert_allclose(det["rcnn_cls"].numpy(), g["rcnn_cls"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(det["rcnn_reg"].numpy(), g["rcnn_reg"], rtol=0, atol=1e-5)
    assert np.array_equal(det["num"].numpy(), g["final_num"]) and g["final_num"].min() >= 10
    np.testing.assert_allclose(det["boxes"].numpy(), g["final_boxes"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(det["scores"].numpy(), g["final_scores"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("intensity", [False, True])
def test_e2e_tiny_matches_reference_model(oracle, intensity):
    """My model + oracle operator backend on CPU vs the reference PointRCNN (run under the shim
    harness when the fixture was made); intensity: cfg.RPN.USE_INTENSITY = True, (B, N, 4) input (rpn.py:17, pointnet2_msg.py:151-160).  With the shared MLPs executed as nn.Modules (reference
    operation order) every tensor is IDENTICAL; with the fused inference path (BN folded into the
    GEMM, fused epilogues) values move by f32 rounding only: within the 1e-4 box tolerance, same
    RoI order, same NMS keep counts."""
    from oracle import ext_cpu
    fm = pkg("pointnet2.fused_mlp")
    model, cfg, g = tiny_model(intensity=intensity)
    pts = torch.from_numpy(g["pts"])
    assert pts.shape[-1] == (4 if intensity else 3)
    with ext_cpu.patch_package():
        fm.ENABLED = False
        try:
            det = pkg("eval_rcnn").infer_batch(model, cfg, pts)
        finally:
            fm.ENABLED = True
        fused = pkg("eval_rcnn").infer_batch(model, cfg, pts)
    assert np.array_equal(det["rois"].numpy(), g["rois"])
    np.testing.assert_allclose(det["rcnn_cls"].numpy(), g["rcnn_cls"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(det["rcnn_reg"].numpy(), g["rcnn_reg"], rtol=0, atol=1e-6)
    assert np.array_equal(det["num"].numpy(), g["final_num"])
    np.testing.assert_allclose(det["boxes"].numpy(), g["final_boxes"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(det["scores"].numpy(), g["final_scores"], rtol=0, atol=1e-6)
    assert g["final_num"].min() >= 1 and g["seg_result"].sum() > 100
    for key, ref in (("rois", "rois"), ("rcnn_cls", "rcnn_cls"), ("rcnn_reg", "rcnn_reg"),
                     ("boxes", "final_boxes"), ("scores", "final_scores")):
        np.testing.assert_allclose(fused[key].numpy(), g[ref], rtol=0, atol=1e-4)
    assert np.array_equal(fused["num"].numpy(), g["final_num"])


def test_fast_engine_point_major_matches_reference_model(oracle):
    """The point-major inference engine (net/fast_infer.py: folded BN, permuted first-layer weights,
    GEMM + fused epilogue, geometry/feature split) on CPU with oracle operators vs the reference
    model's fixture: same RoI order and NMS keep counts, values within the 1e-4 box tolerance."""
    from oracle import ext_cpu
    model, cfg, g = tiny_model()
    E, F = pkg("eval_rcnn"), pkg("net.fast_infer")
    pts = torch.from_numpy(g["pts"])
    with ext_cpu.patch_package():
        eng = F.FastPointRCNN(model, cfg)
        geo = eng.geometry(pts)
        det = E.infer_batch(model, cfg, pts, engine=eng, geo=geo)
        det2 = E.infer_batch(model, cfg, pts, engine=eng)          # geometry computed inline: same thing
    for key, ref in (("rois", "rois"), ("rcnn_cls", "rcnn_cls"), ("rcnn_reg", "rcnn_reg"),
                     ("boxes", "final_boxes"), ("scores", "final_scores")):
        np.testing.assert_allclose(det[key].numpy(), g[ref], rtol=0, atol=1e-4)
        assert torch.equal(det[key], det2[key])
    assert np.array_equal(det["num"].numpy(), g["final_num"])
    assert len(geo["sa"]) == 4 and geo["sa"][0]["idx"][1].shape == (2, 512, 32) and len(geo["fp"]) == 4


def test_proposal_layer_far_band_fallback(oracle):
    """No point beyond 40 m: the far band re-uses the next near proposals (proposal_layer.py:92-99).
    Checked against a literal per-scene restatement with the oracle NMS."""
    from oracle import ext_cpu
    C = pkg("config")
    cfg = C.default_eval_cfg()
    C.merge_into({"TEST": {"RPN_PRE_NMS_TOP_N": 200, "RPN_POST_NMS_TOP_N": 20}}, cfg)
    PL = pkg("net.proposal_layer").ProposalLayer(cfg, mode="TEST")
    rng = np.random.default_rng(4)
    B, N = 2, 600
    xyz = torch.from_numpy(rng.uniform([-10, -1, 2], [10, 3, 38], (B, N, 3)).astype(np.float32))
    xyz[1, :300, 2] += 40.0                                   # scene 1 does have far points
    reg = torch.from_numpy(rng.standard_normal((B, N, 76)).astype(np.float32) * 0.2)
    scores = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32))
    with ext_cpu.patch_package():
        rois, rs = PL(scores, reg, xyz)
        # literal reference order, scene by scene
        bt, ku, iu = pkg("bbox_transform"), pkg("kitti_utils"), pkg("iou3d_utils")
        prop = bt.decode_bbox_target(xyz.view(-1, 3), reg.view(-1, 76), anchor_size=PL.MEAN_SIZE, loc_scope=3.0,
                                     loc_bin_size=0.5, num_head_bin=12, get_xz_fine=True, get_y_by_bin=False,
                                     get_ry_fine=False)
        prop[:, 1] += prop[:, 3] / 2
        prop = prop.view(B, N, 7)
        for b in range(B):
            order = torch.sort(scores[b], descending=True)[1]
            so, po = scores[b][order], prop[b][order]
            dist = po[:, 2]
            first = (dist > 0) & (dist <= 40)
            outs, outb = [], []
            for i, (lo, hi, pre, post) in enumerate(((0, 40, 140, 14), (40, 80, 60, 6))):
                m = (dist > lo) & (dist <= hi)
                if m.sum() != 0:
                    cs, cp = so[m][:pre], po[m][:pre]
                else:
                    assert i == 1
                    cs, cp = so[first][140:][:pre], po[first][140:][:pre]
                keep = iu.nms_normal_gpu(ku.boxes3d_to_bev_torch(cp), cs, 0.8)[:post]
                outs.append(cs[keep]); outb.append(cp[keep])
            wb, ws = torch.cat(outb), torch.cat(outs)
            assert torch.equal(rois[b, :len(wb)], wb) and torch.equal(rs[b, :len(ws)], ws)
            assert (rois[b, len(wb):] == 0).all()


def test_kitti_writer_format(tmp_path):
    E, S = pkg("eval_rcnn"), pkg("synth")
    boxes = np.array([[1.0, 1.6, 20.0, 1.5, 1.6, 3.9, 0.3], [0.0, 1.6, 1.0, 1.5, 1.6, 3.9, 0.0]], np.float32)
    n = E.save_kitti_format(7, S.SyntheticCalib(), boxes, str(tmp_path), np.array([1.5, 0.2]), (375, 1242))
    lines = open(tmp_path / "000007.txt").read().strip().split("\n")
    assert n == 1 and len(lines) == 1                    # the box at z=1 m fills the image and is dropped
    f = lines[0].split()
    assert f[0] == "Car" and len(f) == 16 and f[-1] == "1.5000" and f[11:14] == ["1.0000", "1.6000", "20.0000"]
    E.save_kitti_format(8, S.SyntheticCalib(), boxes[:0], str(tmp_path), np.zeros(0), (375, 1242))
    assert open(tmp_path / "000008.txt").read() == ""


def test_writer_and_host_input_stage_match_reference_executed_fixture(tmp_path):
    """g11 (tests/golden/make_golden.py): the reference's own KittiRCNNDataset.get_rpn_sample / Calibration / get_valid_flag
    (kitti_rcnn_dataset.py:201-342, calibration.py:51-125) and save_kitti_format (tools/eval_rcnn.py:76-101) were RUN on a fake
    KITTI tree (one scene per sampler branch) that regenerates here from the seed.  This build's host stage must give the same
    validity flags, the same chosen rows (legacy np.random stream, per-scene seeding and the sequential single-process stream)
    and the same pts_input bit for bit; the writer the same text, character for character."""
    import helpers
    K, E = pkg("kitti_io"), pkg("eval_rcnn")
    g = load("g11_input_writer_ref.npz")
    cfg = pkg("config").default_eval_cfg()
    ids = helpers.write_fake_kitti_tree(str(tmp_path), int(g["seed"]))
    assert ids == g["ids"].tolist()
    src = K.KittiSource(str(tmp_path), cfg, split="val", npoints_faraway=4000, seed=1024)
    assert src.ids == ids
    for sid in ids:
        lidar, rect, flag, calib, shape = src.rect_and_flags(sid)
        assert abs(float(lidar.astype(np.float64).sum()) - float(g["lidar_sum_%d" % sid])) < 1e-9, "the regenerated tree differs"
        assert tuple(shape[:2]) == tuple(g["shape_%d" % sid]) and len(lidar) == int(g["n_raw_%d" % sid])
        assert rect.dtype == np.float32 and np.array_equal(rect[::97], g["rect_sub_%d" % sid])
        img, depth = calib.rect_to_img(rect)
        assert np.array_equal(img[::97].astype(np.float32), g["img_sub_%d" % sid])
        assert np.array_equal(depth[::97].astype(np.float32), g["depth_sub_%d" % sid])
        want_flag = np.unpackbits(g["valid_%d" % sid])[:len(lidar)].astype(bool)
        assert np.array_equal(flag, want_flag)
        pts, _, _ = src.load(sid)
        want = rect[want_flag][:, 0:3][g["choice_%d" % sid]]
        assert pts.dtype == np.float32 and np.array_equal(pts, want)
        assert float(pts.astype(np.float64).sum()) == float(g["pts_input_sum_%d" % sid])
    stream = np.random.RandomState(1024)                               # tools/eval_rcnn.py:26, one loader process, scenes in order
    seq = [src.load(sid, rng=stream)[0] for sid in ids]
    assert np.array_equal(np.array([p.astype(np.float64).sum() for p in seq]), g["seq_pts_input_sum"])
    assert np.array_equal(np.stack([p[:8] for p in seq], 0), g["seq_first_rows"])
    # the writer: 96 boxes incl. clipped / dropped / x = 0 / x < 0 cases, two calibrations and image sizes
    boxes, scores = helpers.writer_boxes(int(g["seed"]) + 50)
    for k, sid in enumerate(ids[:2]):
        calib, shape = src.calib_and_shape(sid)
        n = E.save_kitti_format(sid, calib, boxes.copy(), str(tmp_path), scores.copy(), shape)
        text = open(tmp_path / ("%06d.txt" % sid)).read()
        assert text == str(g["writer_text"][k]) and n == int(g["writer_lines"][k]) < len(boxes)
        assert E.kitti_result_lines(calib, boxes, scores, shape) == text.split("\n")[:-1]


def test_subsample_rpn_semantics():
    S = pkg("synth")
    raw = S.dense_scene(3, 60000)
    sub = S.subsample_rpn(raw, 16384, 4000, np.random.default_rng(1))
    assert sub.shape == (16384, 3) and (sub[:, 2] >= 40).sum() == min(4000, (raw[:, 2] >= 40).sum())
    small = S.subsample_rpn(raw[:5000], 16384)
    assert small.shape == (16384, 3)
    assert len(np.unique(small, axis=0)) <= 5000


def test_kitti_input_stage(tmp_path):
    """Fake KITTI tree -> KittiSource: calib parsing, lidar->rect->image projection, validity filter
    (image bounds, depth >= 0, PC_AREA_SCOPE) and the 16384-point sampler, against plain numpy."""
    K = pkg("kitti_io")
    cfg = pkg("config").default_eval_cfg()
    root = tmp_path
    d = root / "KITTI" / "object" / "training"
    for sub in ("velodyne", "calib"):
        (d / sub).mkdir(parents=True)
    (root / "KITTI" / "ImageSets").mkdir(parents=True)
    (root / "KITTI" / "ImageSets" / "val.txt").write_text("000003\n000007\n")
    rng = np.random.default_rng(0)
    P2 = np.array([[707.05, 0, 604.08, 45.75], [0, 707.05, 180.5, -0.34], [0, 0, 1, 0.005]], np.float32)
    R0 = np.eye(3, dtype=np.float32)
    Tr = np.array([[0, -1, 0, 0.0], [0, 0, -1, -0.08], [1, 0, 0, -0.27]], np.float32)     # velodyne -> camera axes
    for sid in (3, 7):
        lidar = np.concatenate([rng.uniform([0, -40, -2.5], [75, 40, 1.0], (30000, 3)), rng.uniform(0, 1, (30000, 1))], 1)
        lidar.astype(np.float32).tofile(d / "velodyne" / ("%06d.bin" % sid))
        fmt = lambda name, m: name + ": " + " ".join("%.6e" % v for v in m.reshape(-1)) + "\n"
        (d / "calib" / ("%06d.txt" % sid)).write_text(fmt("P0", P2) + fmt("P1", P2) + fmt("P2", P2) + fmt("P3", P2) +
                                                      fmt("R0_rect", R0) + fmt("Tr_velo_to_cam", Tr) + fmt("Tr_imu_to_velo", Tr))
    src = K.KittiSource(str(root), cfg, "val")
    assert src.ids == [3, 7]
    pts, calib, shape = src.load(7)
    assert pts.shape == (16384, 3) and pts.dtype == np.float32 and shape == (375, 1242, 3)
    np.testing.assert_allclose(calib.P2, P2, rtol=1e-6)
    # every sampled point passes the reference's validity test
    img, depth = calib.rect_to_img(pts)
    assert (img[:, 0] >= 0).all() and (img[:, 0] < 1242).all() and (img[:, 1] >= 0).all() and (img[:, 1] < 375).all()
    assert (depth >= 0).all() and (pts[:, 2] >= 0).all() and (pts[:, 2] <= 70.4).all() and (np.abs(pts[:, 0]) <= 40).all()
    # and they are a subset of the transformed raw cloud
    lidar = np.fromfile(d / "velodyne" / "000007.bin", dtype=np.float32).reshape(-1, 4)
    rect = np.hstack([lidar[:, :3], np.ones((len(lidar), 1), np.float32)]) @ (Tr.T @ R0.T)
    assert len(np.unique(pts, axis=0)) <= len(rect)
    raw = {tuple(np.round(r, 4)) for r in rect}
    assert all(tuple(np.round(p, 4)) in raw for p in pts[:200])
    assert (pts[:, 2] >= 40).sum() <= 4000                       # at most npoints_faraway far points
    # cfg.RPN.USE_INTENSITY: the same rows with the reflectance column shifted to [-0.5, 0.5) (kitti_rcnn_dataset.py:321-338)
    cfg_i = pkg("config").default_eval_cfg()
    pkg("config").merge_into({"RPN": {"USE_INTENSITY": True}}, cfg_i)
    pts4, _, _ = K.KittiSource(str(root), cfg_i, "val").load(7)
    assert pts4.shape == (16384, 4) and pts4.dtype == np.float32
    assert np.array_equal(pts4[:, :3], pts)                      # same seed, same sampler decisions
    refl = {tuple(np.round(r, 4)): v for r, v in zip(rect, lidar[:, 3])}
    for p in pts4[:200]:
        assert abs(refl[tuple(np.round(p[:3], 4))] - 0.5 - p[3]) < 1e-6
    assert pts4[:, 3].min() >= -0.5 and pts4[:, 3].max() < 0.5


def test_runner_choice_follows_the_configuration():
    """make_runner: the stream-pipelined runners for coordinates-only configurations, the serial EngineRunner for cfg.RPN.USE_INTENSITY
    (round 4: the engine covers it on its general kernels), the nn.Module graph (ModuleRunner) for what the engine does not cover
    (cfg.RCNN.USE_INTENSITY); both speak the one-batch-late protocol of the others (here on the CPU with the oracle operator backend)"""
    import copy
    from oracle import ext_cpu
    E = pkg("eval_rcnn")
    model, cfg, g = tiny_model(intensity=True)
    assert E.engine_covers(cfg) and E.engine_covers(pkg("config").default_eval_cfg())
    cfg_r = copy.deepcopy(cfg)
    cfg_r.RCNN.USE_INTENSITY = True
    assert not E.engine_covers(cfg_r) and type(E.make_runner(model, cfg_r, "cpu")) is E.ModuleRunner
    assert type(E.make_runner(model, cfg, "cpu")) is E.EngineRunner
    runner = E.ModuleRunner(model, cfg, "cpu")
    pts = torch.from_numpy(g["pts"])
    with ext_cpu.patch_package():
        assert runner.submit(pts, None) is None
        first = runner.submit(pts[:1].contiguous(), None)
        second = runner.flush()
    assert runner.flush() is None
    np.testing.assert_allclose(first["rois"].numpy(), g["rois"], rtol=0, atol=1e-4)     # (BN folded into the layers: f32 rounding only)
    assert np.array_equal(first["num"].numpy(), g["final_num"])
    np.testing.assert_allclose(first["boxes"].numpy(), g["final_boxes"], rtol=0, atol=1e-4)
    assert second["boxes"].shape[0] == 1
    np.testing.assert_allclose(second["boxes"].numpy(), g["final_boxes"][:1], rtol=0, atol=1e-4)
    src = pkg("kitti_io").SyntheticSource(cfg, 2)
    assert src.load(1)[0].shape == (cfg.RPN.NUM_POINTS, 4)
    with pytest.raises(NotImplementedError):
        pkg("kitti_io").DeviceInputStage(cfg, "cpu")


def test_graph_replay_is_refused_when_the_runtime_switch_came_too_late(monkeypatch):
    """the package sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 at import (tests/conftest.py does it first thing); a process whose HIP runtime was up
    before that must not replay graphs: make_runner() hands out the eager runner with a warning, GraphedRunner itself refuses"""
    import warnings
    P, E = pkg(), pkg("eval_rcnn")
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" and P.GRAPH_REPLAY_SAFE
    monkeypatch.setattr(P, "GRAPH_REPLAY_SAFE", False)
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        E.GraphedRunner(None, None, "cpu")
    made = []
    monkeypatch.setattr(E, "PipelinedRunner", lambda *a: made.append(a) or "eager")
    cfg = pkg("config").default_eval_cfg()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert E.make_runner("model", cfg, "cuda:0") == "eager"
    assert made and any("hipGraph replay disabled" in str(x.message) for x in w)


def test_config_merge_and_set():
    C = pkg("config")
    cfg = C.default_eval_cfg()
    assert cfg.TEST.RPN_NMS_THRESH == 0.8 and cfg.RPN.LOC_XZ_FINE and cfg.RCNN.ENABLED and cfg.RPN.FIXED
    C.cfg_from_list(cfg, ["RCNN.NMS_THRESH", "0.2", "RPN.NUM_POINTS", "32768"])
    assert cfg.RCNN.NMS_THRESH == 0.2 and cfg.RPN.NUM_POINTS == 32768
    with pytest.raises(ValueError):
        C.cfg_from_list(cfg, ["RPN.NUM_POINTS", "'x'"])
    with pytest.raises(KeyError):
        C.merge_into({"NOPE": 1}, cfg, strict=True)


LABELS_GT = """Car 0.00 0 -1.57 600.0 150.0 650.0 200.0 1.50 1.60 3.90 2.00 1.65 20.00 -1.60
Car 0.00 1 -1.20 100.0 150.0 180.0 200.0 1.45 1.55 3.70 -55.00 1.70 30.00 0.30
Car 0.30 2 1.00 700.0 160.0 720.0 175.0 1.50 1.60 4.10 5.00 1.60 85.00 1.57
Car 0.00 0 1.00 700.0 160.0 720.0 175.0 1.50 1.60 4.10 5.00 4.20 35.00 1.57
Van 0.00 0 -1.57 300.0 150.0 380.0 210.0 2.10 1.90 5.20 -6.00 1.75 25.00 -1.50
Pedestrian 0.00 0 0.20 400.0 150.0 420.0 210.0 1.75 0.60 0.80 3.00 1.60 12.00 0.10
DontCare -1 -1 -10 500.0 160.0 540.0 190.0 -1 -1 -1 -1000 -1000 -1000 -10
"""


def test_recall_gt_set_is_class_only_in_eval_like_the_reference(tmp_path):
    """the --recall ground truth must be every labelled object of the class -- the reference's
    filtrate_objects applies check_pc_range in TRAIN mode only (kitti_rcnn_dataset.py:155-176).  The label file holds
    four Cars: one in range, one beyond x = -40, one beyond z = 70.4, one below y = 3 -> eval keeps all 4, train keeps 1.
    In the build container the reference's own filtrate_objects is run on the same file and must agree."""
    K = pkg("kitti_io")
    cfg = pkg("config").default_eval_cfg()
    d = tmp_path / "KITTI" / "object" / "training"
    for sub in ("velodyne", "calib", "label_2"):
        (d / sub).mkdir(parents=True)
    (tmp_path / "KITTI" / "ImageSets").mkdir(parents=True)
    (tmp_path / "KITTI" / "ImageSets" / "val.txt").write_text("000001\n")
    (d / "label_2" / "000001.txt").write_text(LABELS_GT)
    src = K.KittiSource(str(tmp_path), cfg, "val")
    ev = src.gt_boxes3d(1)
    tr = src.gt_boxes3d(1, train_mode=True)
    assert ev.shape == (4, 7) and tr.shape == (1, 7)
    np.testing.assert_allclose(ev[1], [-55.0, 1.70, 30.0, 1.45, 1.55, 3.70, 0.30], rtol=1e-6)
    np.testing.assert_allclose(tr[0], ev[0])
    if os.path.isdir("/root/reference/pointrcnn"):
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
        import ref_harness as H
        H.install()
        from lib.datasets.kitti_rcnn_dataset import KittiRCNNDataset
        import lib.utils.kitti_utils as ku
        from lib.config import cfg as rcfg
        assert rcfg.PC_REDUCE_BY_RANGE
        objs = ku.get_objects_from_label(str(d / "label_2" / "000001.txt"))

        class Fake:
            classes = ("Background", "Car")
            check_pc_range = staticmethod(KittiRCNNDataset.check_pc_range)
        for mode, want in (("EVAL", ev), ("TRAIN", tr)):
            Fake.mode = mode
            kept = KittiRCNNDataset.filtrate_objects(Fake, objs)
            ref = ku.objs_to_boxes3d(kept)
            np.testing.assert_allclose(ref, want, rtol=1e-6)


def test_compiled_extension_modules_load_and_export_the_reference_bindings():
    """The pybind11 modules of csrc/bindings/ (built by __graft_entry__.build() into dropin_native/): importable without a GPU,
    named like the reference's extension modules, exporting exactly the functions the reference binds
    (pointnet2_api.cpp:10-24, iou3d.cpp:174-179, roipool3d.cpp:198-203); device entry points refuse CPU tensors instead of
    computing anything (there is no CPU product path); the two host utilities of roipool3d_cuda do run on CPU tensors and
    reproduce the reference's own compiled roipool3d.cpp (fixture g5)."""
    import importlib
    import sys
    import torch
    p = pkg()
    want = {"pointnet2_cuda": {"ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper", "gather_points_wrapper",
                               "gather_points_grad_wrapper", "furthest_point_sampling_wrapper", "three_nn_wrapper",
                               "three_interpolate_wrapper", "three_interpolate_grad_wrapper"},
            "iou3d_cuda": {"boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu"},
            "roipool3d_cuda": {"pts_in_boxes3d_cpu", "roipool3d_cpu", "forward", "forward_slow"}}
    saved = {n: sys.modules.pop(n, None) for n in want}
    sys.path.insert(0, p.NATIVE_DROPIN_DIR)
    try:
        mods = {n: importlib.import_module(n) for n in want}
        for n, m in mods.items():
            assert m.__file__.endswith(".so") and os.path.dirname(m.__file__) == p.NATIVE_DROPIN_DIR
            assert {f for f in dir(m) if not f.startswith("_")} == want[n], n
        with pytest.raises(RuntimeError):
            mods["pointnet2_cuda"].ball_query_wrapper(1, 4, 2, 1.0, 2, torch.zeros(1, 2, 3), torch.zeros(1, 4, 3),
                                                      torch.zeros(1, 2, 2, dtype=torch.int32))
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g5_roipool_ref.npz"))
        rp = mods["roipool3d_cuda"]
        flag = torch.zeros((16, 4096), dtype=torch.long)
        rp.pts_in_boxes3d_cpu(flag, torch.from_numpy(g["pts"]), torch.from_numpy(g["boxes"]))
        assert np.array_equal(flag.numpy().astype(np.uint8), g["pts_flag"])
        pp, pf, pe = torch.zeros(16, 512, 3), torch.zeros(16, 512, 9), torch.zeros(16, dtype=torch.long)
        rp.roipool3d_cpu(torch.from_numpy(g["pts"]), torch.from_numpy(g["boxes"]), torch.from_numpy(g["feat"]), pp, pf, pe)
        assert np.array_equal(pp.numpy(), g["pooled_pts"]) and np.array_equal(pf.numpy(), g["pooled_features"])
        assert np.array_equal(pe.numpy(), g["pooled_empty_flag"])
    finally:
        sys.path.remove(p.NATIVE_DROPIN_DIR)
        for n in want:
            sys.modules.pop(n, None)
            if saved[n] is not None:
                sys.modules[n] = saved[n]


def test_every_switch_is_registered():
    """switches.py lists exactly the PRCNN_* environment variables the sources read (package, C library, bench.py): a switch that is
    added without a line in the table, or a line whose switch is gone, fails here."""
    import glob
    import re
    import warnings
    SW = pkg("switches")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [f for pat in ("3d_adapt_auto_driving_amd/**/*.py", "3d_adapt_auto_driving_amd/csrc/**/*.hip", "3d_adapt_auto_driving_amd/csrc/**/*.hpp",
                           "3d_adapt_auto_driving_amd/csrc/**/*.cpp", "bench.py") for f in glob.glob(os.path.join(root, pat), recursive=True)]
    read = set()
    for f in files:
        if f.endswith("switches.py"):
            continue
        text = open(f).read()
        read |= set(re.findall(r'getenv\("(PRCNN_[A-Z0-9_]+)"\)', text))
        read |= set(re.findall(r'environ\.get\("(PRCNN_[A-Z0-9_]+)"', text))
    assert read, "no switch found: the scan is broken"
    assert read - set(SW.SWITCHES) == set(), "read but not in switches.py: %s" % sorted(read - set(SW.SWITCHES))
    assert set(SW.SWITCHES) - read == set(), "in switches.py but read nowhere: %s" % sorted(set(SW.SWITCHES) - read)
    for name, (kind, default, where, what) in SW.SWITCHES.items():
        assert kind in ("operational", "numerics", "ab", "tuning", "debug", "bench") and what
        assert any(f.endswith(where) for f in files), (name, where)
    doc = open(os.path.join(root, "docs", "SWITCHES.md")).read()
    assert SW.table() in doc, "docs/SWITCHES.md is stale: regenerate it from switches.table()"
    os.environ["PRCNN_NO_SUCH_SWITCH"] = "1"
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert SW.check_environment() == ["PRCNN_NO_SUCH_SWITCH"] and len(w) == 1
    finally:
        del os.environ["PRCNN_NO_SUCH_SWITCH"]


def test_rcnn_use_intensity_in_the_joint_path_fails_as_the_reference_does(oracle):
    """cfg.RCNN.USE_INTENSITY = True has NO behaviour on the hot path: the reference's joint forward builds the RCNN's input without
    the reflectance (point_rcnn.py:53-57) and rcnn_net.py:131 then reads `rpn_intensity` -- KeyError('rpn_intensity'), verified by
    running the reference under the shims (tests/golden/ref_harness.py; the flag only works in the reference's offline RCNN mode,
    outside SURVEY section 8).  This build's model keeps that: the same exception from the same place, nothing silently different;
    `engine_covers` sends the configuration to the nn.Module graph so that it surfaces."""
    from oracle import ext_cpu
    C, E = pkg("config"), pkg("eval_rcnn")
    cfg = C.default_eval_cfg()
    C.merge_into(TINY, cfg)
    C.merge_into({"RCNN": {"USE_INTENSITY": True}}, cfg)
    assert not E.engine_covers(cfg)
    model = E.build_model(cfg, "cpu")
    assert model.rcnn_net.rcnn_input_channel == 6                       # rcnn_net.py:22: 3 + intensity + mask + depth
    pts = torch.from_numpy(pkg("synth").scenes(1, cfg.RPN.NUM_POINTS, seed0=5))
    with ext_cpu.patch_package(), pytest.raises(KeyError, match="rpn_intensity"):
        with torch.no_grad():
            model({"pts_input": pts})


class _FailingSource:
    """a scene source whose third scene cannot be loaded (picklable: module level)"""
    ids = list(range(16))

    def load(self, i):
        if i == 10:
            raise ValueError("scene %d is broken" % i)
        return np.full((64, 3), float(i), np.float32), pkg("synth").SyntheticCalib(), (375, 1242)


def test_shared_loader_feed_hands_over_the_loaders_bits_in_order(tmp_path):
    """round 6: eval_scenes' loader processes write into ONE shared buffer (eval_rcnn._ShmFeed; page-locked on a GPU box) instead of
    handing tensors over one by one.  On CPU: the batches come back in order with exactly the arrays source.load / load_raw produce
    (sampled clouds; raw clouds of different lengths packed (B, n_max, 4) with their counts), calibration rows and image shapes
    survive the trip, a ragged last batch works, a loader's exception reaches the parent, and a cloud that does not fit a slot says so."""
    E, K, S, C = pkg("eval_rcnn"), pkg("kitti_io"), pkg("synth"), pkg("config")
    cfg = C.default_eval_cfg()
    src = K.SyntheticSource(cfg, 21)                                     # 21 scenes in batches of 8: the last batch holds 5
    feed = E._ShmFeed(src, src.ids, 8, False, 3, "fork", False, 8 * cfg.RPN.NUM_POINTS * 3)
    try:
        for b in range(3):
            host, slot, counts, calibs, shapes = feed.next()
            want = np.stack([src.load(i)[0] for i in src.ids[8 * b:8 * b + 8]], 0)
            assert np.array_equal(host.numpy(), want) and counts == [cfg.RPN.NUM_POINTS] * len(want)
            assert np.allclose(calibs[0].P2, src.calib.P2) and tuple(shapes[0]) == tuple(src.calib.image_shape)
            feed.release(slot, None)
        assert feed.next() is None
    finally:
        feed.close()
    root = str(tmp_path / "tree")
    S.write_kitti_tree(root, 12, pool=4)
    ks = K.KittiSource(root, cfg)
    feed = E._ShmFeed(ks, ks.ids, 8, True, 2, "fork", False, 8 * 60000 * 4)
    try:
        for b in range(2):
            host, slot, counts, calibs, shapes = feed.next()
            for k, i in enumerate(ks.ids[8 * b:8 * b + 8]):
                raw, cal, shape = ks.load_raw(i)
                assert counts[k] == len(raw) and np.array_equal(host[k, :counts[k]].numpy(), raw)
                assert np.allclose(calibs[k].V2C, cal.V2C) and np.allclose(calibs[k].R0, cal.R0) and tuple(shapes[k]) == tuple(shape)
            feed.release(slot, None)
    finally:
        feed.close()
    small = E._ShmFeed(ks, ks.ids, 8, True, 1, "fork", False, 8 * 1000 * 4)    # a slot of 1000 points per raw cloud: too small
    try:
        with pytest.raises(RuntimeError, match="does not fit a loader slot"):
            small.next()
    finally:
        small.close()
    bad = E._ShmFeed(_FailingSource(), _FailingSource.ids, 8, False, 2, "fork", False, 8 * 64 * 3)
    try:
        bad.next()
        with pytest.raises(RuntimeError, match="scene 10 is broken"):
            bad.next()
    finally:
        bad.close()
