"""-m gpu: the BASELINE.json configurations that round 1 left unexercised end to end --
  configs[4]  cross-domain dense clouds (~180 k raw points per scene) through the DEVICE input stage and the engine,
  tools/cfgs/double.yaml  NUM_POINTS = 32768 through the whole pipeline (FPS beyond the pruned kernel's 16384 limit),
and the recall statistics with ground truth (eval_rcnn.py:539-580)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import ext_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_dense_180k_clouds_device_input_stage_then_engine():
    """configs[4] at its stated size on one GPU: 4 scenes x 180 000 raw points -> csrc/input_stage.hip (validity filter +
    near/far sampler) -> 16384 points -> pipelined engine -> detections.  The sampled clouds are subsets of the raw clouds
    without repetition (the scene has > 16384 valid points), the run is deterministic, and the pipelined result equals
    the serial engine on the same sampled clouds."""
    C, E, K, S = pkg("config"), pkg("eval_rcnn"), pkg("kitti_io"), pkg("synth")
    cfg = C.default_eval_cfg()
    model = E.build_model(cfg, DEV, seed=0)
    src = K.SyntheticSource(cfg, 4, raw_points=180000)
    stage = K.DeviceInputStage(cfg, DEV)
    raws = [src.load_raw(i)[0] for i in src.ids]
    meta = [src.calib_and_shape(i) for i in src.ids]
    pts, stats, choice = stage(raws, [m[0] for m in meta], [m[1] for m in meta], src.ids, lidar_frame=False,
                               image_filter=False, return_choice=True)
    torch.cuda.synchronize()
    assert pts.shape == (4, cfg.RPN.NUM_POINTS, 3) and torch.isfinite(pts).all()
    ch = choice.cpu().numpy()
    for k in range(4):
        assert len(np.unique(ch[k])) == cfg.RPN.NUM_POINTS                  # no repetition: enough valid points
        assert np.array_equal(pts[k].cpu().numpy(), raws[k][ch[k], :3])    # every output point IS a raw point
        assert int(stats[k, 0]) > cfg.RPN.NUM_POINTS
    table, counts = E.eval_scenes(model, cfg, DEV, src, src.ids, batch_size=2, device_input=True, workers=0)
    table2, counts2 = E.eval_scenes(model, cfg, DEV, src, src.ids, batch_size=2, device_input=True, workers=0)
    assert torch.equal(table, table2) and torch.equal(counts, counts2)
    runner = E.PipelinedRunner(model, cfg, DEV)
    det = E.infer_batch(model, cfg, pts[:2].contiguous(), engine=runner.engine)
    n = det["num"].cpu()
    assert torch.equal(counts[:2], n)
    for k in range(2):
        assert torch.equal(table[k, :int(n[k]), 0:7], det["boxes"][k, :int(n[k])].cpu())
    assert torch.isfinite(table).all()


def test_double_yaml_32768_points_full_pipeline(oracle):
    """tools/cfgs/double.yaml:39 NUM_POINTS = 32768: FPS 32768 -> 4096 on two workgroups per cloud (fps_spec2_kernel, round 5;
    rounds 1-4: fps_generic_kernel), the hashed-grid ball query and the grid three-NN see n = 32768, the proposal layer its fused
    device path (chunked sort: any n up to 65536).  Indices bit-exact vs the oracle on the xyz chain, engine == module graph
    within 1e-4 (the bar of tests/test_gpu_full_ref.py, where fixture g13 holds the REFERENCE model's outputs at this size),
    deterministic."""
    C, E, F, S = pkg("config"), pkg("eval_rcnn"), pkg("net.fast_infer"), pkg("synth")
    pu = pkg("pointnet2.pointnet2_utils")
    cfg = C.default_eval_cfg()
    C.merge_into({"RPN": {"NUM_POINTS": 32768}}, cfg)
    N = 32768
    xyz = S.scenes(2, N, seed0=900)
    t = torch.from_numpy(xyz).to(DEV)
    sel = pu.furthest_point_sample(t, 4096)
    want = oracle.furthest_point_sample(xyz, 4096)
    assert np.array_equal(sel.cpu().numpy(), want)
    new = np.take_along_axis(xyz, want.astype(np.int64)[..., None].repeat(3, -1), 1)
    newt = torch.from_numpy(new).to(DEV)
    for r, ns in ((0.1, 16), (0.5, 32)):
        assert np.array_equal(pu.ball_query(r, ns, t, newt).cpu().numpy(), oracle.ball_query(r, ns, xyz, new))
    d, i = pu.three_nn(t, newt)
    wd2, wi = oracle.three_nn(xyz, new)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), np.sqrt(wd2))
    model = E.build_model(cfg, DEV, seed=3)
    eng = F.FastPointRCNN(model, cfg)
    d1 = E.infer_batch(model, cfg, t, engine=eng)
    d2 = E.infer_batch(model, cfg, t, engine=eng)
    dm = E.infer_batch(model, cfg, t)
    for k in ("rois", "boxes", "scores", "num"):
        assert torch.equal(d1[k], d2[k]), k
        assert torch.isfinite(d1[k].float()).all()
    assert (d1["rois"] - dm["rois"]).abs().max().item() < 1e-4
    assert torch.equal(d1["num"], dm["num"]) and (d1["boxes"] - dm["boxes"]).abs().max().item() < 1e-4
    # time of the sampling kernel at this size (reported by -s; profiles/r05_double_yaml.md holds the committed numbers)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t8 = torch.from_numpy(S.scenes(8, N, seed0=910)).to(DEV)
    pu.furthest_point_sample(t8, 4096)
    ev[0].record(); pu.furthest_point_sample(t8, 4096); ev[1].record(); torch.cuda.synchronize()
    print("furthest_point_sample 8 x (32768 -> 4096): %.2f ms" % ev[0].elapsed_time(ev[1]))


def test_recall_statistics_with_ground_truth():
    """eval_rcnn.py:539-570 recall statistics through boxes_iou3d_gpu (the extension's BEV overlap kernel): perfect boxes
    recall every gt at every threshold, displaced boxes none, and the device counters equal a host evaluation of the same
    IoU matrices (oracle overlap kernel) on random boxes."""
    C, E, K = pkg("config"), pkg("eval_rcnn"), pkg("kitti_io")
    from helpers import boxes3d
    cfg = C.default_eval_cfg()
    rng = np.random.default_rng(12)
    B, M = 3, 100
    gts = [boxes3d(rng, n, xz_scope=((-10, 10), (8, 30))) for n in (7, 1, 12)]
    gts[1] = np.concatenate([gts[1], np.zeros((3, 7), np.float32)], 0)        # collate padding rows
    pred = np.stack([boxes3d(rng, M, xz_scope=((-10, 10), (8, 30))) for _ in range(B)])
    for k in range(B):                                                           # some predictions near a gt
        n = 7 if k == 0 else (1 if k == 1 else 12)
        pred[k, :n] = gts[k][:n] + rng.normal(0, 0.08, (n, 7)).astype(np.float32)
    rois = pred + rng.normal(0, 0.25, pred.shape).astype(np.float32)
    st = E.RecallStats(DEV)
    st.update(torch.from_numpy(pred).to(DEV), torch.from_numpy(rois).to(DEV), gts)
    res = st.result()
    assert res["total_gt_bbox"] == 20
    # host evaluation with the oracle's overlap kernel
    iu = pkg("iou3d_utils")
    want_rcnn, want_roi = np.zeros(5, int), np.zeros(5, int)
    with ext_cpu.patch_package():
        for k in range(B):
            n = {0: 7, 1: 1, 2: 12}[k]
            g = torch.from_numpy(gts[k][:n])
            for boxes, acc in ((pred[k], want_rcnn), (rois[k], want_roi)):
                best = iu.boxes_iou3d_gpu(torch.from_numpy(boxes), g).max(dim=0).values.numpy()
                for i, th in enumerate(E.RecallStats.THRESH):
                    acc[i] += int((best > th).sum())
    for i, th in enumerate(E.RecallStats.THRESH):
        assert res["rcnn_recalled(thresh=%.2f)" % th] == want_rcnn[i], (th, res, want_rcnn)
        assert res["rpn_recalled(thresh=%.2f)" % th] == want_roi[i], (th, res, want_roi)
    assert want_rcnn[2] >= 15 and want_roi[4] <= want_rcnn[4]
    # perfect / displaced detections
    st2 = E.RecallStats(DEV)
    g = torch.from_numpy(gts[0]).to(DEV).unsqueeze(0)
    st2.update(g, g + torch.tensor([50.0, 0, 0, 0, 0, 0, 0], device=DEV), [gts[0]])
    r2 = st2.result()
    assert r2["rcnn_recall(thresh=0.90)"] == 1.0 and r2["rpn_recall(thresh=0.10)"] == 0.0
    # and through the harness loop on synthetic scenes (random weights: only the plumbing is checked)
    model = E.build_model(cfg, DEV, seed=0)
    src = K.SyntheticSource(cfg, 4)
    st3 = E.RecallStats(DEV)
    E.eval_scenes(model, cfg, DEV, src, src.ids, batch_size=2, workers=0, recall=st3)
    assert st3.result()["total_gt_bbox"] == 40


def test_bench_two_ranks_on_one_gpu_self_launch_barrier_and_gather():
    """BASELINE configs[3] rehearsed on the one GPU of this box (no 8-GPU node was ever available to the driver): `python bench.py
    --gpus 2` as ONE command re-launches itself as two ranks under torch.distributed.run on 127.0.0.1; with
    PRCNN_BENCH_SHARE_GPU=1 both ranks drive cuda:0 and the exchange runs over gloo (RCCL refuses two ranks on one device).
    Proves the self-launch, the rank barriers around the timed region, the max-over-ranks clock and the final all_gather of the
    padded detection tables on hardware: one JSON line from rank 0, n_gpus = 2, twice the scenes of one rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PRCNN_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--prewarm", "6",
                          "--no-roofline", "--no-driver", "--no-cpu-baseline", "--no-lidar"], env=env, cwd=root, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "weak" and line["value"] > 0
    assert abs(line["value"] - 2 * 6 * 8 / (line["ms_per_step"] * 6 * 1e-3)) < 1e-3 * line["value"]     # whole-job scenes / max-over-ranks time
    assert line["config"]["detections_gathered"] > 0                                                   # both ranks' tables arrived on rank 0


def test_rccl_runs_at_world_size_one_and_the_forced_gather_is_the_identity():
    """VERDICT r5 "missing 1": RCCL had never executed on hardware -- all_gather_detections returns early at world size 1 and the
    2-rank rehearsal runs over gloo.  A child process initialises torch.distributed with backend "nccl" (RCCL) as a world of one rank
    on cuda:0 and FORCES the exchange (force=True): padding on the device, the size all_gather, both all_gather_into_tensor calls on
    HIP tensors, strip + id sort; the 472-scene table of rank 0's shard of the val split comes back equal, in scene-id order
    (tests/rccl_world1_child.py holds the asserts)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_world1_child.py"), "472"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ok"] and line["backend"] == "nccl" and line["rows"] == 472


def test_bench_eight_ranks_on_one_gpu():
    """configs[3]'s launch shape rehearsed on the one GPU: `PRCNN_BENCH_SHARE_GPU=1 bench.py --gpus 8` = eight ranks under
    torch.distributed.run driving cuda:0 (gloo: RCCL refuses two ranks per device), barriers, max-over-ranks clock, the gather of
    eight tables: one JSON line, n_gpus 8, eight times the scenes of one rank.  (The scaling CURVE stays unmeasured: no 8-GPU
    node has been available)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PRCNN_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--prewarm", "4",
                          "--windows", "1", "--no-roofline", "--no-driver", "--no-cpu-baseline", "--no-lidar"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 4 and line["scaling"] == "weak" and line["value"] > 0
    assert abs(line["value"] - 8 * 4 * 8 / (line["ms_per_step"] * 4 * 1e-3)) < 1e-3 * line["value"]
    assert line["config"]["detections_gathered"] > 0
