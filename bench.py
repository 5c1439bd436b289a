#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X PointRCNN eval path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one pass (software-pipelined over two HIP streams: geometry of the next batch overlaps the
feature pass of the current one; each step does exactly one of each) of the joint RPN->RCNN hot path (eval_rcnn-equivalent: backbone SA/FP ops,
proposal layer with device NMS, RoI pooling, RCNN, final decode + rotated NMS, async D2H of the
detections) over ONE batch of 8 synthetic KITTI-shaped scenes (16384 points, random-init weights,
default.yaml shapes) per GPU -- BASELINE.json configs[2]; inputs are resident in HBM before the
timed region.  Scenes shard one batch per rank (weak scaling); the only collective is the final
all_gather of the padded detection tables, inside the timed region.

Launch granularity: the geometry of 4 consecutive steps shares one chain of launches (32 clouds), and since the second session of round 4
every stage behind the geometry is launched once per PAIR of consecutive steps (16 scenes, 1600 RoIs: eval_rcnn.GraphedRunner.pair,
`config.steps_per_launch`); each step's batch of 8 scenes still gets its own detections, all K of them inside the closed timed region.

The SA levels run over the DISTINCT grouped rows only (csrc/sa_packed.hip: the reference's ball query back-fills a ball
with copies of its first hit and RoI pooling fills a box with copies of its points; copies do not change a max-pool, the
results are bit-identical -- tests/test_gpu_shadow.py).  How much that saves depends on the data: `config.distinct_rows`
states the measured fraction per level, `config.scenes_per_s_all_rows` is the same step with every nsample row evaluated
the way the reference does (PRCNN_NO_PACK / PRCNN_NO_POOL_DEDUP), measured in this run.

`value` is the MEDIAN of `--windows` (5) closed windows of W + K steps each, all listed in `config.windows`; beside it, never apart
from it, stands `config.lidar_like.scenes_per_s` -- the same engine, same windows, on LiDAR-SHAPED scenes (ray-cast 64-beam sweeps through
the reference's sampler: balls near the sensor are full, 20-60 % of the grouped rows distinct instead of the uniform scene's 3-6 %):
the uniform scene of SURVEY 8d is the distinct-row engine's BEST case, the LiDAR-shaped one its realistic one, and
`config.scenes_per_s_all_rows` the same step with no row dropped.

The JSON line also carries
  roofline      the dominant kernel of the product step -- the largest single launch: the RPN's last stretch over all points
                (prcnn_rpn_tail: interpolation + FP module 0 + both heads, csrc/rpn_tail.hip) on the engine's own inputs, timed
                with HIP events on the launch stream; MFMA-bound: algorithmic flops / average duration against the dense
                f32 MFMA peak; traffic = HBM bytes per launch from the PMC passes (profiles/r02_pmc_product_kernels.md)
  roofline_mfma the hand-written f32 MFMA kernel of the RCNN SA MLP over packed rows (prcnn_sa_packed_mlp) on FULL balls
                (64 distinct rows per centre: every tile does all its flops), against the dense f32 MFMA peak
  roofline_reference_op  fused ball_query+group of the reference's operator API (BASELINE.json configs[1]: B=8, N=16384,
                M=4096, C=128, ns=32, r=0.2; not on the engine's path any more): algorithmic bytes (SURVEY.md section 8d
                formula) / average duration of the launch pair
  roofline_product  the largest HBM-streaming kernel of the product step (prcnn_roipool3d_canonical) the same way
  config.driver_scenes_per_s  the whole driver: loader processes (scene source + 16384-point sampler on the host, as the
                reference's DataLoader workers), pinned upload, the pipelined engine, one D2H per batch, KITTI result
                files written by writer processes -- steady-state rate of eval_rcnn.eval_scenes (rank 0, N = 1 only)
  config.dropin_module_scenes_per_s  the nn.Module graph in the REFERENCE'S operation order over the compiled drop-in modules
                (the reference's 17 entry points only): what a user of the reference's Python gets before touching the engine
  cpu_baseline  the same model code on the host cores with the C oracle as operator backend
                (kind "port": the reference has no CPU path for this pipeline), rank 0, N=1 only.
"""
import argparse
import contextlib
import importlib
import json
import os
import sys
import time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see 3d_adapt_auto_driving_amd/__init__.py (graph replay)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32 MFMA peak (MI355X_MICROARCH.md)
# HBM traffic per launch from the rocprofv3 PMC passes over the product step (FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950
# + WRITE_SIZE), kept with the profile it came from; a kernel that has no entry reports null
PMC_SOURCE = "profiles/r06_pmc_product_kernels.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the product step, bash profiles/measure_r06.sh step)"
# keyed by the scenes per launch the passes ran at (profiles/pmc_step_probe.py: 16 = a pair of batches, the product's launch since the
# second session of round 4; 8: the passes of rounds 2-4 over single batches)
PMC_TRAFFIC = {8: {"rpn_tail_lin_kernel": 148.96e6, "rpn_tail_kernel": 325.87e6, "roipool3d_canonical_kernel": 87.06e6},
               16: {"rpn_tail_lin_kernel": 297.60e6, "rpn_tail_lin_kernel<decode>": 210.66e6, "roipool3d_canonical_kernel": 172.71e6}}      # (round 6 passes: 210.66 / 172.71 MB; round 5: 210.58 / 172.63)
# scenes per launch of the stages behind the geometry in the product runner (eval_rcnn.GraphedRunner pairs batches: PRCNN_PAIR = 2): the
# roofline legs of those kernels run at THIS size, and their PMC traffic comes from passes at this size
def launch_scenes():
    E = importlib.import_module(PKG + ".eval_rcnn")
    pkg = importlib.import_module(PKG)
    paired = E.USE_GRAPHS and getattr(pkg, "GRAPH_REPLAY_SAFE", False)
    group = max(1, int(os.environ.get("PRCNN_GEO_GROUP", "4")))
    pair = max(1, E.RCNN_PAIR) if paired else 1
    return BATCH * (pair if group % pair == 0 else 1)


HOST_LAG = int(os.environ.get("PRCNN_BENCH_LAG", "3"))   # the host consumes a batch's detections this many batches late
BATCH = int(os.environ.get("PRCNN_BENCH_BATCH", "8"))     # scenes per step per GPU (BASELINE configs[2]: 8; the override is for experiments and is echoed in config.env_overrides)
NPOINTS = 16384


def algorithmic_bytes_qg(n, m, c, ns):
    """read xyz once, centres, features once; write idx and the grouped tensor."""
    return 12 * n + 12 * m + 4 * c * n + 4 * m * ns + 4 * (3 + c) * m * ns


def roofline_query_and_group(dev, reps=20):
    """BASELINE's second metric: ball_query + group (prcnn_query_and_group = QueryAndGroup of pointnet2_utils.py:241-264 in one call) at
    B = 8, N = 16384, M = 4096 FPS centres, C = 128, r = 0.2.  The headline object is the uniform scene of SURVEY 8d at nsample = 32 (as
    in rounds 1-5); `lidar_like` holds the same operator on LiDAR-shaped scenes -- full balls, 32 / 64 DISTINCT gathers per centre, the
    regime the reference's grouping kernel (group_points_gpu.cu:47-66) sees on KITTI -- at nsample 32 and 64, and `uniform_ns64`
    the uniform scene at nsample = 64 (VERDICT r5 "missing 2").  HBM traffic cannot be measured from inside the process: it comes
    from the committed rocprofv3 PMC passes over this same operator with the kernels that run NOW (profiles/
    r06_pmc_query_and_group_{uniform,lidar}.json, produced by `bash profiles/measure_r06.sh qg`: separate --pmc FETCH_SIZE /
    WRITE_SIZE passes over profiles/qg_sweep.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), and is reported only
    if shape and kernel names match."""
    pkg = importlib.import_module(PKG)
    if pkg.DROPIN_DIR not in sys.path:
        sys.path.insert(0, pkg.DROPIN_DIR)
    import pointnet2_cuda
    synth = importlib.import_module(PKG + ".synth")
    B, N, M, C, R = BATCH, NPOINTS, 4096, 128, 0.2
    kernels = "dense_build_reg_kernel<16> + dense_query_kernel + group_cat_lds_kernel<1, true>"

    def one(kind, NS):
        make = synth.lidar_scenes if kind == "lidar" else synth.scenes
        xyz = torch.from_numpy(make(B, N, seed0=1000)).to(dev)
        temp = torch.full((B, N), 1e10, device=dev)
        sel = torch.empty((B, M), dtype=torch.int32, device=dev)
        pointnet2_cuda.furthest_point_sampling_wrapper(B, N, M, xyz, temp, sel)
        new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        feats = torch.randn((B, C, N), device=dev)
        idx = torch.empty((B, M, NS), dtype=torch.int32, device=dev)
        out = torch.empty((B, 3 + C, M, NS), device=dev)
        for _ in range(3):
            pointnet2_cuda.query_and_group_wrapper(B, N, M, C, R, NS, new_xyz, xyz, feats, idx, out)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:   # HIP events recorded on the stream the kernels are launched on (torch's current stream)
            a.record()
            pointnet2_cuda.query_and_group_wrapper(B, N, M, C, R, NS, new_xyz, xyz, feats, idx, out)
            b.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        nbytes = B * algorithmic_bytes_qg(N, M, C, NS)
        achieved = nbytes / (ms * 1e-3) / 1e9
        # mean number of DISTINCT neighbours per ball (back-filled slots repeat slot 0: ball_query_gpu.cu:35-39)
        first = idx[:, :, :1]
        distinct = float(((idx != first).sum(dim=2) + 1).float().mean())
        traffic = per_kernel = src = None
        for cand in ("r06_pmc_query_and_group_%s.json" % kind,) + (("r04_pmc_query_and_group.json",) if (kind, NS) == ("uniform", 32) else ()):
            try:
                with open(os.path.join(ROOT, "profiles", cand)) as f:
                    pmc = json.load(f)
                if NS == 64:
                    pmc = pmc.get("ns64", {})
                if pmc.get("algorithmic_bytes_per_launch") == nbytes and pmc.get("kernels") == kernels:
                    traffic, per_kernel, src = pmc["hbm_traffic_bytes_per_launch"], pmc.get("per_kernel"), os.path.join("profiles", cand)
                    break
            except (OSError, ValueError, KeyError):
                pass
        return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": (src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over profiles/qg_sweep.py)") if traffic else None,
                "kernel": "prcnn_query_and_group = " + kernels, "per_kernel_profile": per_kernel,
                "launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": nbytes, "distinct_neighbours_per_ball": round(distinct, 2),
                "shape": {"B": B, "N": N, "M": M, "C": C, "nsample": NS, "radius": R, "scene": kind}}

    line = one("uniform", 32)
    line["uniform_ns64"] = one("uniform", 64)
    line["lidar_like"] = {"ns32": one("lidar", 32), "ns64": one("lidar", 64)}
    return line


def roofline_sa_mlp_fused(dev, reps=10):
    """The dominant kernel of the step: the hand-written MFMA kernel that runs a whole RCNN
    set-abstraction MLP (gather -> 3 layers -> max over nsample) at the RCNN SA1 size of one batch
    (B = 8 scenes x 100 RoIs = 800 clouds of 512 points, 128 centres x 64 samples, 128-128-128).
    Algorithmic flops = the two dense layers (2 * rows * (128*128 + 128*128)); the gathered layer 1 and
    the pooling ride along.  Peak = dense f32 MFMA (MI355X_MICROARCH.md: 157.3 TFLOP/s)."""
    pkg = importlib.import_module(PKG)
    if pkg.DROPIN_DIR not in sys.path:
        sys.path.insert(0, pkg.DROPIN_DIR)
    import pointnet2_cuda
    b, n, m, ns, c3 = 100 * BATCH, 512, 128, 64, 128
    g = torch.Generator(device=dev).manual_seed(0)
    xyz = torch.randn((b, n, 3), device=dev, generator=g)
    new_xyz = xyz[:, :m].contiguous()
    P = torch.randn((b, n, 128), device=dev, generator=g)
    wx = torch.randn((3, 128), device=dev, generator=g)
    idx = torch.randint(0, n, (b, m, ns), dtype=torch.int32, device=dev, generator=g)
    w2 = torch.randn((128, 128), device=dev, generator=g) / 11
    w3 = torch.randn((128, c3), device=dev, generator=g) / 11
    b2 = torch.randn(128, device=dev, generator=g)
    b3 = torch.randn(c3, device=dev, generator=g)
    out = torch.empty((b, m, c3), device=dev)
    # full balls: 64 DISTINCT point indices per centre (a random permutation prefix), so the packed list holds every row
    idx = torch.argsort(torch.rand((b, m, n), device=dev, generator=g), dim=2)[:, :, :ns].to(torch.int32).contiguous()
    pack = pointnet2_cuda.ball_pack_wrapper(idx, xyz, new_xyz)
    assert int(pack.hdr[1]) == b * m * ns
    run = lambda: pointnet2_cuda.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, pack, w2, b2, w3, b3, out, 0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in evs:
        a.record(); run(); e.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(e) for a, e in evs]))
    flops = 2.0 * b * m * ns * (128 * 128 + 128 * c3)
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
            "kernel": "sa_packed_mlp128_kernel (prcnn_sa_packed_mlp) on full balls: 64 distinct rows per centre",
            "launch_ms": round(ms, 4), "algorithmic_flops_per_launch": flops,
            "shape": {"clouds": b, "points": n, "centres": m, "nsample": ns, "mlp": [128, 128, c3]}}


def roofline_rpn_tail(dev, cfg, model, reps=20):
    """The dominant kernel of the product step (the largest single launch, ~9 % of a step): the RPN's last stretch over all
    8 x 16384 points in one kernel (csrc/rpn_tail.hip: interpolation -> FP module 256-128-128 -> cls 128-128-1 and reg
    128-128-76), on the inputs the engine really hands it (neighbour indices / weights of three_nn on a synthetic batch, the
    FP1 output as the table).  MFMA-bound: achieved = algorithmic flops / average HIP-event duration against the dense f32
    MFMA peak.  traffic = HBM bytes per launch from the PMC passes over the product step."""
    F = importlib.import_module(PKG + ".net.fast_infer")
    pu = importlib.import_module(PKG + ".pointnet2.pointnet2_utils")
    synth = importlib.import_module(PKG + ".synth")
    eng = F.FastPointRCNN(model, cfg)
    if eng.rpn_tail is None:
        return None
    pts = torch.from_numpy(synth.scenes(launch_scenes(), NPOINTS, seed0=3000)).to(dev)      # one launch of the product = a pair of batches
    geo = eng.geometry(pts)
    _, (known, idx, weight) = eng._backbone(pts, geo, fuse_tail=True)
    tw = eng.rpn_tail
    B, N = idx.shape[0], idx.shape[1]
    feats = torch.empty((B, N, 128), device=dev); cls = torch.empty((B, N, 1), device=dev); reg = torch.empty((B, N, tw["n_reg"]), device=dev)
    lin = bool(F.USE_FP_LINEAR and hasattr(pu.pointnet2, "rpn_tail_lin_wrapper"))
    dec = None
    if lin:
        # the engine's form since round 3: FP layer 1 applied over the coarse points (its own small launch, not timed here), the fused
        # kernel interpolates the 128-wide product and starts at layer 2
        m = known.shape[1]
        G = F.point_layer(known.view(B * m, known.shape[2]), tw["w1"], tw["zero128"], False).view(B, m, 128)
        dec = eng._tail_decode_cfg(N)
        if dec is not None:
            # the product's form since round 5: the proposal layer's decode inside -- the 7-float box leaves instead of the 76-float row
            boxes = torch.empty((B, N, 7), device=dev)
            run = lambda: pu.pointnet2.rpn_tail_lin_boxes_wrapper(G, idx, weight, tw["wcat_lin"], tw["bcat"], tw["wc2"], tw["bc2"], tw["n_reg"],
                                                                  dec[0], dec[1], dec[2], dec[3], dec[4], pts, feats, cls, boxes)
        else:
            run = lambda: pu.pointnet2.rpn_tail_lin_wrapper(G, idx, weight, tw["wcat_lin"], tw["bcat"], tw["wc2"], tw["bc2"], feats, cls, reg)
    else:
        run = lambda: pu.pointnet2.rpn_tail_wrapper(known, idx, weight, tw["wcat"], tw["bcat"], tw["wc2"], tw["bc2"], feats, cls, reg)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in evs:
        a.record(); run(); e.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(e) for a, e in evs]))
    rows = B * N
    # ALGORITHMIC flops of what THIS kernel computes: [FP layer 1 (256-128) only in the round-2 form] FP layer 2 (128-128), cls head
    # 128-128-1, reg head 128-128-n_reg (76 under default.yaml).  The kernel computes the regression layer as a zero-padded 128-column
    # tile (`padded_flops`): those columns multiply zeros and are NOT counted in `achieved`.
    l1 = 0 if lin else 256 * 128
    flops = 2.0 * rows * (l1 + 3 * 128 * 128 + 128 + 128 * tw["n_reg"])
    narrow = lin and 64 < tw["n_reg"] <= 80 and os.environ.get("PRCNN_TAIL_NARROW", "1") != "0"     # round 5: 80 computed columns, not 128
    padded_flops = 2.0 * rows * (l1 + 3 * 128 * 128 + 128 + 128 * (80 if narrow else 128))
    achieved = flops / (ms * 1e-3) / 1e12
    table = (G if lin else known).numel() * 4
    # read the interpolated table once, indices + weights (+ the coordinates when the boxes are decoded here); write features, score and
    # the regression row -- or, decoded, the 7-float box
    alg_bytes = table + rows * 24 + (rows * 12 if dec is not None else 0) + rows * (128 + 1 + (7 if dec is not None else tw["n_reg"])) * 4
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": PMC_TRAFFIC.get(B, {}).get(("rpn_tail_lin_kernel<decode>" if dec is not None else "rpn_tail_lin_kernel") if lin else "rpn_tail_kernel"),
            "traffic_source": PMC_SOURCE,
            "kernel": "%s (prcnn_rpn_tail%s): the largest single launch on the FEATURE stream (the longest launch of "
                      "the step overall is the sampling kernel on a side stream: see roofline_longest)" % (
                          ("rpn_tail_lin_kernel<decode, narrow>" if dec is not None else "rpn_tail_lin_kernel") if lin else "rpn_tail_kernel",
                          ("_lin_boxes" if dec is not None else "_lin") if lin else ""),
            "launch_ms": round(ms, 4), "algorithmic_flops_per_launch": flops, "padded_flops_per_launch": padded_flops,
            "algorithmic_bytes_per_launch": alg_bytes,
            "shape": {"scenes_per_launch": B, "points": rows, "coarse_points": known.shape[0] * known.shape[1],
                      "layers": (("interp(128) | 128-128 | 128-128-1 | 128-128-%d" + (" | box decode" if dec is not None else "")) if lin
                                 else "256-128-128 | 128-128-1 | 128-128-%d") % tw["n_reg"]}}


def roofline_roipool(dev, cfg, model, reps=20):
    """The largest HBM-streaming kernel of the product step: RoI pooling + canonical transform + RCNN row layout
    (prcnn_roipool3d_canonical) on one batch of 8 scenes x 100 RoIs x 512 points x (8 + 128) floats, in the form the engine
    launches it (pooled_cnt given: the feature columns of the wrap-around copies beyond the first multiple of 64 rows are not
    written -- nobody reads them).  Algorithmic bytes (roipool row, with this kernel's row layout):
    read xyz, features, mask, depth once and the RoIs; write the rows this form writes (counted from pooled_cnt)."""
    pkg = importlib.import_module(PKG)
    if pkg.DROPIN_DIR not in sys.path:
        sys.path.insert(0, pkg.DROPIN_DIR)
    import roipool3d_cuda
    F = importlib.import_module(PKG + ".net.fast_infer")
    synth = importlib.import_module(PKG + ".synth")
    eng = F.FastPointRCNN(model, cfg)
    LB = launch_scenes()                    # one launch of the product = a pair of batches
    pts = torch.from_numpy(synth.scenes(LB, NPOINTS, seed0=2000)).to(dev)
    st = eng.rpn_stage(pts)
    rois, _ = eng.propose(st)
    feats, mask = st["rpn_features"], st["seg_result"].contiguous()
    depth = (st["pts_depth"] / 70.0 - 0.5).contiguous()
    B, M, S, C = LB, rois.shape[1], cfg.RCNN.NUM_POINTS, feats.shape[2]
    pooled = torch.empty((B, M, S, 8 + C), device=dev)
    empty = torch.empty((B, M), dtype=torch.int32, device=dev)
    cnt = torch.empty((B, M), dtype=torch.int32, device=dev)
    groups = st.get("groups")             # the scene's spatial groups (built with the geometry chain): the selection culls by them
    xyz_dense = torch.empty((B, M, S, 3), device=dev)     # round 4: the pooled coordinates once more as dense clouds (what the RCNN's SA levels read)
    run = lambda: roipool3d_cuda.forward_canonical(pts, rois.contiguous(), feats, mask, depth, cfg.RCNN.POOL_EXTRA_WIDTH, pooled, empty, cnt, groups, xyz_dense)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in evs:
        a.record(); run(); e.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(e) for a, e in evs]))
    full_rows = int(torch.clamp((cnt.long() + 63) // 64 * 64, max=S).sum())        # rows written with their feature columns
    distinct_rows = int(torch.clamp(cnt.long(), max=S).sum())                      # pooled points that are not wrap-around copies
    # bytes this FORM must move (round 2 charged all N feature rows, more than the PMC traffic): xyz, mask
    # and depth of every point once; the RoIs; the feature row of every DISTINCT pooled point once (a gather: rows no box holds
    # are never read); every row it writes (with feature columns up to the first multiple of 64 rows, 32 B beyond)
    nbytes = (B * NPOINTS * (12 + 8) + B * M * (28 + 8) + distinct_rows * 4 * C
              + full_rows * (8 + C) * 4 + (B * M * S - full_rows) * 32 + B * M * S * 12)
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": PMC_TRAFFIC.get(B, {}).get("roipool3d_canonical_kernel"), "traffic_source": PMC_SOURCE,
            "kernel": "roipool3d_canonical_kernel (prcnn_roipool3d_canonical, product form)", "launch_ms": round(ms, 4),
            "algorithmic_bytes_per_launch": nbytes, "mean_points_per_roi": round(float(cnt.float().mean()), 1),
            "shape": {"B": B, "N": NPOINTS, "rois": M, "sampled": S, "row_floats": 8 + C}}


def driver_leg(cfg, model, dev, scenes=4096):
    """eval_rcnn.eval_scenes with everything the reference's loop has around the model (eval_rcnn.py:493-649): loader processes
    produce the 16384-point clouds, pinned H2D, pipelined engine, one D2H per batch, KITTI result files by writer processes.
    Steady-state rates (loader / writer process start-up excluded) of three drivers:
      * `value`: the uniform synthetic scene source, host-side generator in the loaders (rounds 2-5's figure);
      * `lidar_kitti_tree`: a KITTI-format tree of LiDAR-shaped sweeps on disk (synth.write_kitti_tree: velodyne .bin + calib files,
        64 distinct sweeps hard-linked to `scenes` ids) through kitti_io.KittiSource -- the reference's get_rpn_sample per scene in the
        loaders (read, lidar_to_rect, rect_to_img, validity filter, near / far sampler: kitti_rcnn_dataset.py:249-324);
      * `lidar_kitti_tree_device_input`: the same files with --device_input: the loaders only read the raw clouds, rectification, filter
        and sampler run on the device (csrc/input_stage.hip)."""
    import shutil
    import tempfile
    E = importlib.import_module(PKG + ".eval_rcnn")
    K = importlib.import_module(PKG + ".kitti_io")
    synth = importlib.import_module(PKG + ".synth")

    def run(src, device_input=False):
        out = tempfile.mkdtemp(prefix="prcnn_bench_")
        stats = {}
        try:
            table, counts = E.eval_scenes(model, cfg, dev, src, src.ids, BATCH, out, device_input=device_input, stats=stats)
            files = len(os.listdir(out))
        finally:
            shutil.rmtree(out, ignore_errors=True)
        return {"value": round(E.steady_state_rate(stats, BATCH), 1), "unit": "scenes/s", "scenes": len(src.ids), "result_files": files,
                "detections": int(counts.sum()), "host_budget": stats.get("host_budget"), "loaders": (stats.get("loader_calibration") or {}).get("loaders"),
                "host_phases_ms_per_batch": stats.get("host_phases_ms_per_batch")}

    line = run(K.SyntheticSource(cfg, scenes))
    line["what"] = ("eval_scenes: synthetic scene source + host 16384-point stage in loader processes, pinned upload, engine, "
                    "D2H, KITTI text files by writer processes; steady state between the first and the last batch")
    tree = tempfile.mkdtemp(prefix="prcnn_tree_")
    try:
        synth.write_kitti_tree(tree, scenes, pool=64)
        line["lidar_kitti_tree"] = run(K.KittiSource(tree, cfg))
        line["lidar_kitti_tree_device_input"] = run(K.KittiSource(tree, cfg), device_input=True)
        line["lidar_kitti_tree"]["what"] = ("the same loop over a KITTI-format tree of LiDAR-shaped sweeps (64 distinct .bin files, ~34 k raw points, "
                                            "hard-linked to %d ids): kitti_io.KittiSource = the reference's get_rpn_sample in the loaders" % scenes)
    except Exception as e:                                          # noqa: BLE001 -- report legs, never the headline
        line["lidar_kitti_tree_error"] = repr(e)[:300]
    finally:
        shutil.rmtree(tree, ignore_errors=True)
    return line


def dropin_module_leg(cfg, model, dev, steps=6):
    """What a user of the REFERENCE'S Python gets from the drop-in modules (north_star: "models load unmodified"): the nn.Module
    graph in the reference's operation order -- FPS -> gather -> ball_query -> group_points x 2 -> subtract -> cat -> Conv2d /
    BatchNorm / ReLU modules -> max_pool2d (pointnet2_modules.py:19-55), three_nn -> three_interpolate -> cat -> Conv1d, the per-scene
    proposal layer and final stage over the blocking nms_gpu / nms_normal_gpu, roipool3d forward -- through the COMPILED modules
    of dropin_native/ (the reference's 17 entry points and nothing else; eval_rcnn.reference_api_only).  One batch of 8 scenes
    per step on the default stream, detections copied out; no engine, no fused entry, no side stream."""
    E = importlib.import_module(PKG + ".eval_rcnn")
    synth = importlib.import_module(PKG + ".synth")
    batches = [torch.from_numpy(synth.scenes(BATCH, NPOINTS, seed0=9000 + BATCH * k)).to(dev) for k in range(2)]
    out = {}
    for name, ctx in (("reference_order_native_modules", lambda: E.reference_api_only(native=True)),
                      ("this_builds_module_graph", contextlib.nullcontext)):
        with ctx():
            for k in range(2):
                E.infer_batch(model, cfg, batches[k])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                det = E.infer_batch(model, cfg, batches[k % 2])
                host = [det[key].cpu() for key in ("boxes", "scores", "num")]
            torch.cuda.synchronize()
            out[name] = round(steps * BATCH / (time.perf_counter() - t0), 1)
    return {"value": out["reference_order_native_modules"], "unit": "scenes/s", "steps": steps,
            "module_graph_with_fused_entries": out["this_builds_module_graph"],
            "what": "nn.Module graph in the reference's operation order over the compiled dropin_native modules (the reference's 17 entry "
                    "points only; library convolutions), batch of %d scenes per step; beside it the same graph with this build's "
                    "fused entries (query_and_group, folded MLPs, device-side proposal / final stage)" % BATCH}


def cpu_baseline(cfg, budget_s=30.0):
    """Same Python model code on CPU tensors, operator backend = the C oracle (OpenMP), convs = PyTorch CPU
    (kind "port": the reference has no CPU path for this pipeline).  BASELINE.md section 3 asks for a warm-up and the
    median of >= 5 runs: a batch of 8 takes ~18 s on the box, so the sample is batches of FOUR scenes (~4.5 s each; single
    scenes are unfair to the CPU: 2.0 s per scene at B = 1 against 1.05 s at B = 8, the convolutions thread over the batch):
    1 warm-up scene, then distinct batches until `budget_s` seconds are spent, at least 5, at most 8."""
    from oracle import ext_cpu, oracle as O
    E = importlib.import_module(PKG + ".eval_rcnn")
    synth = importlib.import_module(PKG + ".synth")
    model = E.build_model(cfg, "cpu")
    CB = 4
    pts = torch.from_numpy(synth.scenes(1 + 8 * CB, NPOINTS, seed0=0))
    times = []
    with ext_cpu.patch_package():
        E.infer_batch(model, cfg, pts[:1])          # warm-up (allocator, thread pools)
        spent = 0.0
        while len(times) < 8 and (len(times) < 5 or spent + times[-1] <= budget_s):
            k = 1 + CB * len(times)
            t0 = time.perf_counter()
            E.infer_batch(model, cfg, pts[k:k + CB])
            times.append(time.perf_counter() - t0)
            spent += times[-1]
    dt = float(np.median(times))
    threads = int(max(O.num_threads(), torch.get_num_threads()))
    return {"value": round(CB / dt, 4), "unit": "scenes/s", "cores": threads, "threads": threads, "host_cpus": os.cpu_count(),
            "kind": "port", "sample": "batches of %d synthetic scenes x %d points, full RPN+RCNN+postprocess; 1 warm-up scene, "
                                      "median of %d timed batches (%.1f s of CPU work, min %.2f / max %.2f s per batch)"
                                      % (CB, NPOINTS, len(times), spent, min(times), max(times))}


def roofline_fps(dev, reps=3):
    """The LONGEST launch of the step (on a side stream, one launch per group of 4 batches): furthest point sampling
    16384 -> 4096 over the 32 clouds of a geometry group, one workgroup per cloud.  A chain of M - 1 dependent picks; since
    round 4 a workgroup exchange decides SEVERAL of them (csrc/fps.hip fps_spec_kernel: ~5.5 picks per round on the uniform
    scene, ~4.4 on LiDAR-shaped ones).  Reported as time per pick beside the floor of the one-pick-per-exchange skeleton
    (0.69 us: scalar pivot load, box test, one LDS atomic, one barrier, one LDS read) that round 3's kernel ran at 0.99 us
    against, and, for completeness, on its algorithmic bytes (12 N + 4 M per cloud) against HBM -- a latency-bound kernel is
    far from any bandwidth roofline by construction."""
    pkg = importlib.import_module(PKG)
    if pkg.DROPIN_DIR not in sys.path:
        sys.path.insert(0, pkg.DROPIN_DIR)
    import pointnet2_cuda
    synth = importlib.import_module(PKG + ".synth")
    group = int(os.environ.get("PRCNN_GEO_GROUP", "4"))
    B, N, M = BATCH * group, NPOINTS, 4096
    xyz = torch.from_numpy(synth.scenes(BATCH, N, seed0=5000)).to(dev).repeat(group, 1, 1).contiguous()
    temp = torch.empty((B, N), device=dev)
    sel = torch.empty((B, M), dtype=torch.int32, device=dev)
    run = lambda: pointnet2_cuda.furthest_point_sampling_wrapper(B, N, M, xyz, temp.fill_(1e10), sel)
    run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in evs:
        temp.fill_(1e10)
        a.record(); pointnet2_cuda.furthest_point_sampling_wrapper(B, N, M, xyz, temp, sel); e.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(e) for a, e in evs]))
    nbytes = B * (12 * N + 4 * M)
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "latency", "kernel": "fps_spec_kernel<16>: speculative multi-pick (round 4), several exact picks per workgroup exchange "
                                          "(+ fps_order_kernel), %d clouds per launch" % B,
            "launch_ms": round(ms, 4), "us_per_pick": round(ms * 1e3 / (M - 1), 4),
            "sequential_exchange_floor_us": 0.69,      # what ONE pick per exchange cannot go below (round 3's kernel: 0.99 us per pick)
            "algorithmic_bytes_per_launch": nbytes,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "workgroups": B, "cus_held": B, "shape": {"clouds": B, "N": N, "M": M}}


def in_step(kernel_prefix, peak_time_field=None):
    """The same kernel's average duration INSIDE the pipelined step (other streams' kernels beside it), read from the committed rocprofv3
    step tables (profiles/r06_bench_step_kernel_stats_{uniform,lidar}.md, produced by `bash profiles/measure_r06.sh step`): the
    rooflines below time their kernel ALONE on the chip with HIP events; VERDICT r5 W5 / W6 asked for the in-step figure beside it.
    -> {"uniform": avg ms per launch, "lidar": ...} (None where the table or the row is missing)."""
    out = {}
    for kind in ("uniform", "lidar"):
        v = None
        try:
            with open(os.path.join(ROOT, "profiles", "r06_bench_step_kernel_stats_%s.md" % kind)) as f:
                for ln in f:
                    if ln.startswith("| `") and kernel_prefix in ln.split("`")[1]:
                        v = round(float(ln.rstrip().rstrip("|").split("|")[-1]) / 1e3, 4)
                        break
        except (OSError, ValueError, IndexError):
            pass
        out[kind] = v
    return out


def with_in_step(line, kernel_prefix, work_per_launch, peak, scale=1.0):
    """adds launch_ms_in_step / frac_in_step (work per launch / in-step duration / peak) to a roofline object"""
    if line is None:
        return line
    ms = in_step(kernel_prefix)
    line["launch_ms_in_step"] = ms
    line["frac_in_step"] = {k: (None if not v else round(work_per_launch * scale / (v * 1e-3) / peak, 4)) for k, v in ms.items()}
    line["in_step_source"] = "profiles/r06_bench_step_kernel_stats_{uniform,lidar}.md (rocprofv3 --kernel-trace of the pipelined bench)"
    return line


def rccl_world1_leg():
    """RCCL on this box (untimed, a child process so that a collective-library failure cannot take the headline with it):
    torch.distributed "nccl" as a world of one rank, the job's one exchange forced through it on rank 0's shard of the 3769-scene
    val split (472 scenes x 100 x 9 f32) -- tests/rccl_world1_child.py holds the equality asserts.  Evidence that the N > 1 code
    runs on HIP tensors over RCCL; NOT a scaling number (the curve needs an 8-GPU node: DESIGN section 8)."""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py"), "472"], env=env, cwd=ROOT,
                             capture_output=True, text=True, timeout=300)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"ok": False, "error": (out.stderr or out.stdout)[-400:]}
        return json.loads(lines[-1])
    except Exception as e:                                        # noqa: BLE001 -- a report field, never the headline
        return {"ok": False, "error": repr(e)[:400]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-driver", action="store_true", help="skip the whole-driver leg (loader processes + writer)")
    ap.add_argument("--no-lidar", action="store_true", help="skip the LiDAR-shaped-scene leg (config.lidar_like)")
    ap.add_argument("--scene", choices=("uniform", "lidar"), default="uniform",
                    help="scene generator of the HEADLINE loop: SURVEY 8d's uniform synthetic scene (default, the contract) or "
                         "synth.lidar_scene (for profiling that regime; the default run reports it under config.lidar_like)")
    ap.add_argument("--windows", type=int, default=5, help="closed timed windows of W + K steps each; `value` is the MEDIAN window, config.windows lists them all (VERDICT r4 W9: one K = 20 window in six came out 8 % low)")
    ap.add_argument("--points", type=int, default=16384, help="points per scene of the HEADLINE loop: 16384 = default.yaml (the contract); 32768 = tools/cfgs/double.yaml, "
                                                              "for profiling that configuration (the default run reports it under config.double_yaml_scenes_per_s)")
    ap.add_argument("--prewarm", type=int, default=24, help="untimed set-up steps before the W warm-up steps (allocator pool, code objects)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as ONE command: re-launch this script as N ranks (one process per GPU) under
        # torch.distributed.run on the loopback address; rank 0 of that job prints the JSON line.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU product path)")
    # PRCNN_BENCH_SHARE_GPU=1: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, gloo instead of RCCL, which refuses two ranks on one device).  Never set by the driver.
    share = os.environ.get("PRCNN_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if share else local_rank)
    torch.cuda.set_device(dev)
    comm_dev = torch.device("cpu") if share else dev
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    C = importlib.import_module(PKG + ".config")
    E = importlib.import_module(PKG + ".eval_rcnn")
    synth = importlib.import_module(PKG + ".synth")
    cfg = C.default_eval_cfg()
    if args.points != 16384:
        global NPOINTS
        NPOINTS = args.points
        C.merge_into({"RPN": {"NUM_POINTS": NPOINTS}}, cfg)
    model = E.build_model(cfg, dev, seed=0)
    M = cfg.TEST.RPN_POST_NMS_TOP_N

    # distinct synthetic scenes per rank and per step slot, resident in HBM before timing.  The slots outnumber the runner's
    # look-ahead (with fewer, the upcoming list aliases the current batch and geometry chains, keyed by tensor
    # identity, group differently from the real driver)
    n_slots = E.PipelinedRunner.default_depth() + 2
    make_scenes = synth.lidar_scenes if args.scene == "lidar" else synth.scenes
    batches = [torch.from_numpy(make_scenes(BATCH, NPOINTS, seed0=(rank * n_slots + s) * BATCH)).to(dev)
               for s in range(n_slots)]
    F = importlib.import_module(PKG + ".net.fast_infer")
    lagged = os.environ.get("PRCNN_TAIL_OVERLAP", "1") != "0"

    shared_runner = {}

    def timed_run(steps, warmup, batches=batches, fresh=False, keep_gc_off=False, key="runner", run_cfg=None):
        """W untimed + K timed steps of the pipelined runner; returns the elapsed time of the K steps and their detections"""
        # point-major engine + geometry chains on side streams; E.make_runner: the stages replayed as hipGraphs (captured once, at the
        # runner's first batch = during the set-up run below: graphs are part of the engine like the folded weights) unless PRCNN_GRAPHS=0.
        # `fresh`: an eager runner of its own (the caller changed engine switches; a captured graph would not see them)
        if fresh:
            runner = E.PipelinedRunner(model, cfg, dev)
        else:
            runner = shared_runner.setdefault(key, None) or E.make_runner(model, run_cfg or cfg, dev)
            shared_runner[key] = runner
        assert runner.depth + 2 <= len(batches), "batch slots must outnumber the look-ahead"
        n_slots = len(batches)
        total = warmup + steps
        # one pinned record per batch: boxes | scores | num (E.split_detections) -- the runner hands them over in one allocation
        host_blob = torch.empty((total, BATCH * (M * 8 + 1)), pin_memory=True)
        views = [E.split_detections(host_blob[i], BATCH, M) for i in range(total)]
        host_boxes = [v[0] for v in views]; host_scores = [v[1] for v in views]; host_num = [v[2] for v in views]

        import collections
        ready = collections.deque()

        def copy_out(det, i):
            # async D2H of batch i's detections into its pinned slot, on the stream that produced them
            with torch.cuda.stream(det["stream"]) if "stream" in det else contextlib.nullcontext():
                if det.get("blob") is not None:
                    host_blob[i].copy_(det["blob"], non_blocking=True)
                else:
                    host_boxes[i].copy_(det["boxes"], non_blocking=True)
                    host_scores[i].copy_(det["scores"], non_blocking=True)
                    host_num[i].copy_(det["num"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            # the host consumes results 3 batches late (as eval_rcnn.eval_scenes does): it never waits for the batch it
            # just submitted, but it cannot run arbitrarily far ahead of the device either -- unbounded run-ahead keeps
            # growing the set of live temporaries, i.e. hipMalloc calls inside the loop
            ready.append(ev)
            if len(ready) > HOST_LAG:
                ready.popleft().synchronize()

        def step(i, end):
            # every step does one RPN pass (batch i) and one RCNN + final pass; in the three-stream form the latter belongs to
            # batch i-1 (software pipeline, eval_rcnn.PipelinedRunner.submit) and the per-scene tails run beside the feature
            # stream.  Geometry runs ahead in chains over groups of batches, but never beyond `end`: the warm-up and the timed
            # region are each CLOSED -- no chain of a timed batch starts before the clock does, none is launched for a batch
            # that will not be processed, and the first chain's latency (cold start) is inside the timed region.
            nxt = [batches[(i + d) % n_slots] for d in range(1, runner.depth + 1) if i + d < end]
            if lagged:
                det = runner.submit(batches[i % n_slots], nxt)         # the detections of an EARLIER batch (in submit order), or None
                if det is not None:
                    copy_out(det, out_next[0])
                    out_next[0] += 1
            else:
                copy_out(runner.step(batches[i % n_slots], nxt), i)

        out_next = [0]                         # index of the batch whose detections come back next

        def drain(last):
            while lagged:                      # the runner hands back one batch per call until its pipeline is empty
                det = runner.flush()
                if det is None:
                    break
                copy_out(det, out_next[0])
                out_next[0] += 1
            assert not lagged or out_next[0] == last + 1, "detections of %d batches came back, %d were submitted" % (out_next[0], last + 1)

        # host housekeeping BEFORE the warm-up steps, not between them and the clock: a full collection takes tens of milliseconds, the GPU
        # falls back to its idle clocks meanwhile, and the ramp back up landed inside the timed region -- one K = 20 run in eight came out
        # 5-25 % low while the K = 100 loop of the same process did not move (second session of round 4)
        import gc
        tg = time.perf_counter()
        gc.collect()
        gc.disable()                           # a generational collection inside a 70 ms timed region costs several percent
        timed_run.gc_ms = round((time.perf_counter() - tg) * 1e3, 1)
        # ... and the device is WOKEN before the W warm-up steps (round 6): a collection of the heap the graph captures leave behind
        # takes 100+ ms, the idle GPU drops its clocks, and W = 5 steps (5 ms) do not bring them all the way back -- the first of five
        # windows came out 10-26 % low in half of the runs (`config.windows.gc_ms` shows the pause).  ~20 ms of dense
        # f32 work on the current stream, untimed like the warm-up steps it precedes, none of it the workload's.
        if warmup and steps > 1:
            wake = shared_runner.setdefault("wake", None)
            if wake is None:
                wake = shared_runner["wake"] = (torch.randn((4096, 4096), device=dev), torch.empty((4096, 4096), device=dev))
            for _ in range(24):
                torch.mm(wake[0], wake[0], out=wake[1])
        try:
            for i in range(warmup):
                step(i, warmup)
            if warmup:
                drain(warmup - 1)              # the pipeline is empty when timing starts ...
            assert not getattr(runner, "_chains", None), "a geometry chain of a timed batch was started during the warm-up"
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
        except BaseException:
            gc.enable()
            raise
        allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        timed_run.slot_phase = getattr(runner, "_next_slot", 0) % max(1, getattr(runner, "n_slots", 1))   # diagnostic: which group slot / side stream the window's first chain takes
        try:
            t0 = time.perf_counter()
            for i in range(warmup, total):
                step(i, total)
            drain(total - 1)                   # ... and drained inside the timed region: exactly K full batches
            torch.cuda.synchronize()
        finally:
            if not keep_gc_off:                # (the headline run: the caller's exchange step belongs to the timed region as well)
                gc.enable()
        timed_run.device_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0
        return t0, [(host_boxes[i], host_scores[i], host_num[i]) for i in range(warmup, total)]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # set-up, before the W warm-up steps: one closed run of the same loop so that the caching allocator owns the blocks the
    # steady state needs (every hipMalloc inside a step stalls the device) and HIP has loaded every code object
    if args.prewarm > 0:
        timed_run(args.prewarm, 0)
        # ... and one UNTIMED rehearsal of a whole window -- W warm-up + K steps, the detection tables packed and exchanged -- so that
        # the first timed window is not the first time this exact sequence runs (the first of five windows came out
        # 26 % low on the driver's box: 5477 against 7366-7396 scenes/s; the median hid it, config.windows showed it)
        _, dets0 = timed_run(args.steps, args.warmup)
        E.all_gather_detections(*E.pack_detections(list(range(rank * args.steps * BATCH, (rank + 1) * args.steps * BATCH)), dets0, M), comm_dev)
        barrier()
        torch.cuda.synchronize()
        del dets0
    # The headline: `--windows` CLOSED windows, each W warm-up steps + exactly K timed steps + the job's one exchange, bracketed by
    # barrier + synchronize on both sides, the pipeline empty at both ends -- and the MEDIAN window is what `value` reports (each
    # window's rate is in config.windows: a single 26-ms window is at the mercy of the clock ramp of an idle GPU).
    ids = list(range(rank * args.steps * BATCH, (rank + 1) * args.steps * BATCH))
    import gc
    window_s, allocs_w, phases_w, gc_w = [], [], [], []
    for _ in range(max(1, args.windows)):
        t0, dets = timed_run(args.steps, args.warmup, keep_gc_off=True)
        allocs_w.append(getattr(timed_run, "device_allocs", None))     # hipMalloc calls inside the timed region (each one stalls the device)
        gc_w.append(getattr(timed_run, "gc_ms", None))
        phases_w.append(getattr(timed_run, "slot_phase", None))
        # the one exchange of the job: padded detection tables of this rank's scenes
        table, counts = E.pack_detections(ids, dets, M)
        table, counts = E.all_gather_detections(table, counts, comm_dev)
        barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        window_s.append(el)
    order = sorted(range(len(window_s)), key=lambda i: window_s[i])
    mid = order[(len(order) - 1) // 2]                  # the median window (the faster of the middle two for an even count)
    elapsed, allocs_main = window_s[mid], allocs_w[mid]

    # ---- context for the headline (rank 0, untimed): how many grouped rows are distinct on this data, and the same step
    # with every nsample row evaluated (the reference's way)
    pu = importlib.import_module(PKG + ".pointnet2.pointnet2_utils")

    def distinct_rows(batch):
        """Fraction of the grouped rows that are distinct, per ball-query shape of one product step on `batch`."""
        real_pack, seen = pu.pointnet2.ball_pack_wrapper, []

        def spy(idx, *a):
            pk = real_pack(idx, *a)
            seen.append((tuple(idx.shape), pk.hdr))
            return pk
        pu.pointnet2.ball_pack_wrapper = spy
        try:
            E.infer_batch(model, cfg, batch, engine=F.FastPointRCNN(model, cfg))
            torch.cuda.synchronize()
        finally:
            pu.pointnet2.ball_pack_wrapper = real_pack
        return {"%dx%dx%d" % shp: round(int(hdr[1]) / float(shp[0] * shp[1] * shp[2]), 4) for shp, hdr in seen}

    def all_rows_rate(batch_set, k):
        saved = (F.USE_PACKED, F.USE_POOL_DEDUP)
        F.USE_PACKED, F.USE_POOL_DEDUP = False, False
        try:
            t1, _ = timed_run(k, 3, batch_set, fresh=True)
            return round(k * BATCH / (time.perf_counter() - t1), 1)
        finally:
            F.USE_PACKED, F.USE_POOL_DEDUP = saved

    def note(what):                      # PRCNN_BENCH_TRACE=1: which leg is running (stderr), for bisecting a failing run
        if os.environ.get("PRCNN_BENCH_TRACE") == "1":
            torch.cuda.synchronize()
            print("[bench] " + what, file=sys.stderr, flush=True)

    distinct, all_rows, steady, lidar = None, None, None, None
    if world == 1 and not args.no_roofline:      # (N = 1 only: these legs call the timed loop, which holds rank barriers)
        note('distinct rows')
        distinct = distinct_rows(batches[0])
        note('all rows')
        all_rows = all_rows_rate(batches, max(4, min(args.steps, 20)))
        # the same closed loop at K = 100: cold start + drain are 5 % of it instead of 20 % at the driver's K = 20
        note('K = 100')
        t1, _ = timed_run(100, args.warmup)
        steady = round(100 * BATCH / (time.perf_counter() - t1), 1)
    if world == 1 and not args.no_lidar:
        # ---- the same engine on LiDAR-SHAPED scenes (synth.lidar_scene: a ray-cast 64-beam sweep through the reference's
        # near / far sampler): density falls with range as on KITTI, most balls near the sensor are full, so the distinct-row
        # saving is realistic instead of at its best case (VERDICT r2 "what's weak" 2)
        note('lidar')
        lb = [torch.from_numpy(synth.lidar_scenes(BATCH, NPOINTS, seed0=70000 + s * BATCH)).to(dev) for s in range(n_slots)]
        timed_run(max(args.prewarm // 2, 4), 0, lb)
        l_windows = []
        for _ in range(max(1, args.windows)):
            t1, _ = timed_run(args.steps, args.warmup, lb)
            l_windows.append(args.steps * BATCH / (time.perf_counter() - t1))
        l_rate = sorted(l_windows)[len(l_windows) // 2]            # the median window, as the headline
        t1, _ = timed_run(100, args.warmup, lb)
        l_steady = 100 * BATCH / (time.perf_counter() - t1)
        note('lidar context')
        eng = F.FastPointRCNN(model, cfg)
        st = eng.rpn_stage(lb[0])
        rois, _ = eng.propose(st)
        pooled = eng.rcnn_geometry(st, rois) if hasattr(eng, "rcnn_geometry") else None
        per_roi = None
        if isinstance(pooled, dict) and pooled.get("pooled_cnt") is not None:
            per_roi = round(float(pooled["pooled_cnt"].float().mean()), 1)
        lidar = {"scenes_per_s": round(l_rate, 1), "steps": args.steps, "windows": [round(v, 1) for v in l_windows],
                 "scenes_per_s_k100": round(l_steady, 1),
                 "distinct_rows": distinct_rows(lb[0]), "mean_points_per_roi": per_roi,
                 "scenes_per_s_all_rows": all_rows_rate(lb, max(4, min(args.steps, 20))) if not args.no_roofline else None,
                 "scene": "synth.lidar_scene: 64 beams x 0.1728 deg azimuth steps over +-40.5 deg, ground + cars + facades + poles, "
                          "~28 k raw points in PC_AREA_SCOPE -> reference near/far sampler -> 16384"}

    double_leg = None
    if world == 1 and not args.no_lidar and not args.no_roofline:
        # ---- tools/cfgs/double.yaml (NUM_POINTS 32768, everything else default.yaml's): the same engine and runner, the same closed
        # windows, on 32768-point scenes (round 5: sampling on two workgroups per cloud, fused proposal path up to 65536 points)
        note('double.yaml')
        cfg2 = C.default_eval_cfg()
        C.merge_into({"RPN": {"NUM_POINTS": 32768}}, cfg2)
        db = [torch.from_numpy(synth.scenes(BATCH, 32768, seed0=90000 + s * BATCH)).to(dev) for s in range(n_slots)]
        timed_run(max(args.prewarm // 2, 4), 0, db, key="runner32k", run_cfg=cfg2)
        d_windows = []
        for _ in range(3):
            t1, _ = timed_run(args.steps, args.warmup, db, key="runner32k", run_cfg=cfg2)
            d_windows.append(args.steps * BATCH / (time.perf_counter() - t1))
        double_leg = {"scenes_per_s": round(sorted(d_windows)[1], 1), "windows": [round(v, 1) for v in d_windows], "steps": args.steps,
                      "points_per_scene": 32768, "what": "tools/cfgs/double.yaml:39 through make_runner(): median of 3 closed windows"}
        shared_runner.pop("runner32k", None)
        del db
        torch.cuda.empty_cache()

    scenes_total = world * args.steps * BATCH
    line = {
        "metric": "KITTI scenes/s end-to-end eval_rcnn (joint RPN+RCNN, synthetic 16384-pt scenes)",
        "value": round(scenes_total / elapsed, 3), "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: full PointRCNN RPN+RCNN inference, default.yaml shapes, "
                               "random-init weights, batch=8 synthetic KITTI scenes x 16384 pts per GPU per step",
                   "scene": "synth.scenes (SURVEY 8d: uniform clutter + ground + 10 dense cars)" if args.scene == "uniform" else "synth.lidar_scenes",
                   "scenes_per_step_per_gpu": BATCH, "points_per_scene": NPOINTS, "rois_per_scene": M,
                   "parallelism": "scene-sharded x%d, one final all_gather of detections" % world,
                   "detections_gathered": int(counts.sum()) if rank == 0 else None,
                   "distinct_rows": distinct, "scenes_per_s_all_rows": all_rows, "scenes_per_s_k100": steady,
                   "lidar_like": lidar, "double_yaml_scenes_per_s": double_leg,
                   # every closed window of this run (W + K steps each), in run order; `value` is the median one
                   "windows": {"n": len(window_s), "reported": "median", "scenes_per_s": [round(world * args.steps * BATCH / w, 1) for w in window_s],
                               "min": round(world * args.steps * BATCH / max(window_s), 1), "max": round(world * args.steps * BATCH / min(window_s), 1),
                               "first_group_slot": phases_w, "gc_ms": gc_w, "device_allocs": allocs_w},
                   "device_allocs_in_timed_region": allocs_main,
                   # every PRCNN_* switch this process saw (22 of them select kernels at import time): a line
                   # measured with a non-default engine says so
                   "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith("PRCNN_")},
                   "batch_slots": n_slots, "look_ahead": E.PipelinedRunner.default_depth(),
                   "hip_graphs": ({"captured": shared_runner["runner"].captures, "group_slots": shared_runner["runner"].n_slots}
                                  if getattr(shared_runner.get("runner"), "captures", 0) else None),
                   # the graphed runner launches every stage behind the geometry once per PAIR of consecutive steps (16 scenes, 1600 RoIs
                   # per launch): K steps = K batches of 8 scenes all the same, each batch's detections handed back separately
                   "steps_per_launch": getattr(shared_runner.get("runner"), "pair", 1)},
    }
    if rank == 0:
        if not args.no_roofline:
            note("rooflines")
            line["roofline"] = roofline_rpn_tail(dev, cfg, model) or roofline_sa_mlp_fused(dev)
            if "rpn_tail" in line["roofline"].get("kernel", ""):
                with_in_step(line["roofline"], "rpn_tail_lin_kernel", line["roofline"]["algorithmic_flops_per_launch"], MFMA_F32_PEAK_TFLOPS * 1e12)
            line["roofline_longest"] = roofline_fps(dev)
            line["roofline_longest"]["launch_ms_in_step"] = in_step("fps_spec_kernel<16>")
            line["roofline_mfma"] = roofline_sa_mlp_fused(dev)
            rp = line["roofline_product"] = roofline_roipool(dev, cfg, model)
            if rp:
                with_in_step(rp, "roipool3d_canonical_kernel", rp.get("algorithmic_bytes_per_launch", 0), HBM_PEAK_GBS * 1e9)
            line["roofline_reference_op"] = roofline_query_and_group(dev)
        if world == 1 and not args.no_driver:
            line["config"]["driver_scenes_per_s"] = driver_leg(cfg, model, dev)
        if world == 1 and not args.no_roofline:
            note("drop-in module path")
            line["config"]["dropin_module_scenes_per_s"] = dropin_module_leg(cfg, model, dev)
        if world == 1 and not args.no_roofline:
            note("RCCL at world size 1")
            line["config"]["rccl_world1"] = rccl_world1_leg()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
