"""How many index decisions move when the arithmetic is FMA-CONTRACTED?   (TEST INFRASTRUCTURE, like all of oracle/)

The reference builds its kernels with `nvcc -O2` (pointnet2_lib/pointnet2/setup.py:19-20), which fuses a*b+c into
FMAs -- e.g. the squared distance of ball_query_gpu.cu:33, sampling_gpu.cu:133, interpolate_gpu.cu:37 and the cross
products / rotations of iou3d_kernel.cu and roipool3d_kernel.cu.  That binary cannot run here, so its bits are
unobservable; the parity contract of this repo is the UN-contracted source semantics.  This script
bounds the distance between the two: it runs every index-deciding operator of the eval path ONCE with the contract-off
oracle (liboracle.so) and replays the SAME inputs on two FMA-contracted builds of the same source

    fma   liboracle_fma.so   whole file under gcc -ffp-contract=fast -mfma (the analogue of nvcc -O2; the distance
                             becomes fma(dz,dz, fma(dx,dx, dy*dy)), the form LLVM/NVVM and gcc both choose)
    fma2  liboracle_fma2.so  same, with the other association of the distance: fma(dz,dz, fma(dy,dy, dx*dx))

and counts what differs.  Operators are replayed on identical inputs (the contract-off run's), so a flipped FPS pick
does not cascade into the rows below it.

    python -m oracle.fma_table [--scenes 2] [--full]      -> markdown table (DESIGN.md section 3 holds a committed copy)

--full adds BASELINE configs[2] (the whole RPN+RCNN pipeline on CPU, every extension call intercepted)."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "3d_adapt_auto_driving_amd"

from oracle import oracle as O  # noqa: E402

VARIANTS = ("fma", "fma2")


class Tally:
    """per (operator, variant): units compared / units that differ (+ the largest value difference where it applies)"""

    def __init__(self):
        self.rows = {}

    def add(self, op, variant, total, differ, unit, maxdiff=None, note=""):
        r = self.rows.setdefault((op, variant), {"total": 0, "differ": 0, "unit": unit, "maxdiff": None, "calls": 0, "note": note})
        r["total"] += int(total)
        r["differ"] += int(differ)
        r["calls"] += 1
        if maxdiff is not None:
            r["maxdiff"] = maxdiff if r["maxdiff"] is None else max(r["maxdiff"], maxdiff)

    def markdown(self):
        ops = []
        for (op, _v) in self.rows:
            if op not in ops:
                ops.append(op)
        out = ["| operator | calls | units compared | differ under `fma` | differ under `fma2` | largest value difference |",
               "|---|---|---|---|---|---|"]
        for op in ops:
            a, b = self.rows.get((op, "fma")), self.rows.get((op, "fma2"))
            ref = a or b
            md = max([r["maxdiff"] for r in (a, b) if r and r["maxdiff"] is not None], default=None)

            def cell(r):
                if r is None:
                    return "-"
                return "%d (%.4g %%)" % (r["differ"], 100.0 * r["differ"] / max(1, r["total"]))
            out.append("| %s | %d | %d %s | %s | %s | %s |" % (op, ref["calls"], ref["total"], ref["unit"], cell(a), cell(b),
                                                            "-" if md is None else "%.3g" % md))
        return "\n".join(out)


def _fps_first_divergence(a, b):
    """FPS is sequential: after the first different pick the two runs are different samplings.  Report the number of
    scenes whose pick sequence differs anywhere (unit = scene) -- and separately positions."""
    neq = (a != b)
    return int(neq.any(axis=1).sum()), int(neq.sum())


def op_level(tally, scenes=2, npoints=16384, seed0=0):
    """BASELINE configs[1] (+ the other levels of the RPN backbone): FPS, ball query at the six (r, nsample) pairs of
    the config and the eight of the backbone, three_nn at the four FP levels -- on synthetic KITTI-shaped scenes."""
    synth = importlib.import_module(PKG + ".synth")
    xyz = synth.scenes(scenes, npoints, seed0=seed0)
    levels = [(4096, [(0.1, 16), (0.5, 32)]), (1024, [(0.5, 16), (1.0, 32)]), (256, [(1.0, 16), (2.0, 32)]), (64, [(2.0, 16), (4.0, 32)])]
    ratio = npoints / 16384.0
    cur = xyz
    l_xyz = [xyz]
    for li, (npoint, pairs) in enumerate(levels):
        npoint = max(8, int(npoint * ratio))
        sel = O.furthest_point_sample(cur, npoint)
        for v in VARIANTS:
            with O.variant(v):
                got = O.furthest_point_sample(cur, npoint)
            sc, pos = _fps_first_divergence(sel, got)
            tally.add("furthest_point_sample (scenes with any different pick)", v, sel.shape[0], sc, "scenes")
            tally.add("furthest_point_sample (pick positions)", v, sel.size, pos, "picks")
        new = np.take_along_axis(cur, sel.astype(np.int64)[..., None].repeat(3, -1), 1)
        qpairs = list(pairs)
        if li == 0:
            qpairs += [(r, ns) for r in (0.1, 0.2, 0.4) for ns in (32, 64)]       # BASELINE configs[1]
        for r, ns in qpairs:
            want = O.ball_query(r, ns, cur, new)
            for v in VARIANTS:
                with O.variant(v):
                    got = O.ball_query(r, ns, cur, new)
                tally.add("ball_query (centre rows)", v, want.shape[0] * want.shape[1], (want != got).any(-1).sum(), "rows")
                tally.add("ball_query (index slots)", v, want.size, (want != got).sum(), "slots")
        cur = new
        l_xyz.append(new)
    for k in range(4):
        d2, idx = O.three_nn(l_xyz[k], l_xyz[k + 1])
        for v in VARIANTS:
            with O.variant(v):
                gd2, gidx = O.three_nn(l_xyz[k], l_xyz[k + 1])
            tally.add("three_nn (neighbour indices)", v, idx.size, (idx != gidx).sum(), "indices")
            tally.add("three_nn (squared distances, value)", v, d2.size, (d2 != gd2).sum(), "values",
                      float(np.abs(d2.astype(np.float64) - gd2).max()))
    return tally


def box_level(tally, seed=0):
    """NMS (both kinds), BEV overlap / IoU, RoI pooling and the evaluator's rotated IoU on random KITTI-sized boxes."""
    rng = np.random.default_rng(seed)

    def boxes3d(rng, n, xz_scope):
        x = rng.uniform(*xz_scope[0], n); z = rng.uniform(*xz_scope[1], n)
        y = rng.uniform(1.2, 2.0, n)
        h = rng.uniform(1.3, 1.8, n); w = rng.uniform(1.4, 1.9, n); l = rng.uniform(3.2, 4.6, n)
        return np.stack([x, y, z, h, w, l, rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)

    def bev_of(b):     # kitti_utils.boxes3d_to_bev_torch: [x - l/2, z - w/2, x + l/2, z + w/2, ry]
        hl, hw = b[:, 5] / 2, b[:, 4] / 2
        return np.stack([b[:, 0] - hl, b[:, 2] - hw, b[:, 0] + hl, b[:, 2] + hw, b[:, 6]], 1).astype(np.float32)
    for trial in range(8):
        n = 2700 if trial == 0 else 600
        b3 = boxes3d(rng, n, xz_scope=((-20, 20), (5, 45)))
        bev = bev_of(b3)
        for name, fn, th in (("nms_normal (RPN, thresh 0.8: kept lists)", O.nms_normal, 0.8), ("nms rotated (final, thresh 0.1: kept lists)", O.nms, 0.1)):
            want = fn(bev, th)
            for v in VARIANTS:
                with O.variant(v):
                    got = fn(bev, th)
                same = len(want) == len(got) and np.array_equal(want, got)
                tally.add(name, v, 1, 0 if same else 1, "problems")
        a, b = bev[:256], bev[256:512]
        wo, wi = O.boxes_overlap_bev(a, b), O.boxes_iou_bev(a, b)
        for v in VARIANTS:
            with O.variant(v):
                go, gi = O.boxes_overlap_bev(a, b), O.boxes_iou_bev(a, b)
            tally.add("boxes_overlap_bev (value)", v, wo.size, (wo != go).sum(), "values", float(np.abs(wo - go).max()))
            tally.add("boxes_iou_bev (value)", v, wi.size, (wi != gi).sum(), "values", float(np.abs(wi - gi).max()))
    synth = importlib.import_module(PKG + ".synth")
    for s in range(4):
        pts = synth.scene(100 + s, 16384)[None]
        centres = pts[0, rng.choice(16384, 100, replace=False)]
        bx = np.concatenate([centres + [0, 1.0, 0], np.tile([3.5, 3.6, 5.9], (100, 1)), rng.uniform(-np.pi, np.pi, (100, 1))], 1)[None].astype(np.float32)
        feat = rng.standard_normal((1, 16384, 4)).astype(np.float32)
        wp, we = O.roipool3d(pts, bx, feat, 512)
        for v in VARIANTS:
            with O.variant(v):
                gp, ge = O.roipool3d(pts, bx, feat, 512)
            tally.add("roipool3d (boxes whose pooled point set differs)", v, 100, (wp != gp).any(axis=(2, 3)).sum(), "boxes")
    return tally


class _Replay:
    """Proxy around one of oracle.ext_cpu's operator classes: every call runs on the contract-off oracle (its result is
    what the pipeline continues with) and is then replayed on the same inputs under each FMA variant."""

    OUTS = {"ball_query_wrapper": (7,), "query_and_group_wrapper": (9,), "furthest_point_sampling_wrapper": (5,), "three_nn_wrapper": (5, 6),
            "forward": (3, 4), "nms_gpu": (1,), "nms_normal_gpu": (1,), "nms_device": (5, 6)}

    def __init__(self, backend, tally, prefix):
        self._b, self._t, self._p = backend, tally, prefix

    def __getattr__(self, name):
        import torch
        fn = getattr(self._b, name)
        if name not in self.OUTS:
            return fn

        def call(*args):
            ins = [a.clone() if torch.is_tensor(a) else a for a in args]
            ret = fn(*args)
            for v in VARIANTS:
                rep = [a.clone() if torch.is_tensor(a) else a for a in ins]
                with O.variant(v):
                    r2 = fn(*rep)
                self._count(name, v, args, rep, ret, r2)
            return ret
        return call

    def _count(self, name, v, args, rep, ret, r2):
        t, p = self._t, self._p
        if name == "ball_query_wrapper":
            a, b = args[7], rep[7]
            t.add(p + "ball_query (centre rows)", v, a.shape[0] * a.shape[1], int((a != b).any(-1).sum()), "rows")
        elif name == "query_and_group_wrapper":
            a, b = args[9], rep[9]
            t.add(p + "ball_query (centre rows)", v, a.shape[0] * a.shape[1], int((a != b).any(-1).sum()), "rows")
        elif name == "furthest_point_sampling_wrapper":
            a, b = args[5], rep[5]
            t.add(p + "furthest_point_sample (clouds with any different pick)", v, a.shape[0], int((a != b).any(-1).sum()), "clouds")
        elif name == "three_nn_wrapper":
            a, b = args[6], rep[6]
            t.add(p + "three_nn (neighbour indices)", v, a.numel(), int((a != b).sum()), "indices")
        elif name == "forward":
            a, b = args[3], rep[3]
            t.add(p + "roipool3d (boxes whose pooled point set differs)", v, a.shape[0] * a.shape[1], int((a != b).flatten(2).any(-1).sum()), "boxes")
        elif name in ("nms_gpu", "nms_normal_gpu"):
            same = ret == r2 and bool((args[1][:ret] == rep[1][:r2]).all())
            t.add(p + name + " (kept lists)", v, 1, 0 if same else 1, "problems")
        elif name == "nms_device":
            same = (args[6] == rep[6]) & ((args[5] == rep[5]).all(-1))
            t.add(p + "nms_device %s (kept lists)" % ("rotated" if args[3] else "normal"), v, same.numel(), int((~same).sum()), "problems")


def pipeline_level(tally, scenes=1, cfg_overrides=None, seed0=0):
    """BASELINE configs[2]: the whole RPN+RCNN pipeline (default.yaml shapes unless overridden) on CPU tensors with the
    oracle as operator backend; every index-deciding extension call is replayed under the FMA variants."""
    import torch
    from oracle import ext_cpu
    C, E, S = (importlib.import_module(PKG + "." + m) for m in ("config", "eval_rcnn", "synth"))
    cfg = C.default_eval_cfg()
    if cfg_overrides:
        C.merge_into(cfg_overrides, cfg)
    model = E.build_model(cfg, "cpu", seed=3)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():      # spread the heads so that proposals / detections are not all near-ties (as tests/test_gpu_e2e.py)
        for name, p in model.named_parameters():
            if ("reg_layer" in name or "cls_layer" in name) and p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        model.rcnn_net.cls_layer[-1].conv.weight.mul_(0.05)
        model.rcnn_net.cls_layer[-1].conv.bias.fill_(0.5)
    pts = torch.from_numpy(S.scenes(scenes, cfg.RPN.NUM_POINTS, seed0=seed0))
    pu = importlib.import_module(PKG + ".pointnet2.pointnet2_utils")
    iu = importlib.import_module(PKG + ".iou3d_utils")
    ru = importlib.import_module(PKG + ".roipool3d_utils")
    with ext_cpu.patch_package():
        saved = (pu.pointnet2, iu.iou3d_cuda, ru.roipool3d_cuda)
        pu.pointnet2 = _Replay(saved[0], tally, "pipeline: ")
        iu.iou3d_cuda = _Replay(saved[1], tally, "pipeline: ")
        ru.roipool3d_cuda = _Replay(saved[2], tally, "pipeline: ")
        try:
            E.infer_batch(model, cfg, pts)
        finally:
            pu.pointnet2, iu.iou3d_cuda, ru.roipool3d_cuda = saved
    return tally


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=2)
    ap.add_argument("--pipeline-scenes", type=int, default=1)
    ap.add_argument("--full", action="store_true", help="also run the whole pipeline (BASELINE configs[2]) on CPU")
    args = ap.parse_args()
    t = Tally()
    op_level(t, scenes=args.scenes)
    box_level(t)
    if args.full:
        pipeline_level(t, scenes=args.pipeline_scenes)
    print(t.markdown())


if __name__ == "__main__":
    main()
