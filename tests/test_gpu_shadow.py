"""-m gpu: BASELINE configs[2] at its stated batch of 8 -- every extension call of one full RPN+RCNN step is SHADOWED by the
CPU oracle on the very tensors the GPU kernel received.

The engine runs on the GPU exactly as in bench.py (point-major engine, packed MFMA kernels); a proxy around the extension
modules copies each call's arguments to the host, lets the HIP kernel run, then runs the oracle stand-in of the same entry
point (oracle/ext_cpu.py: scalar C restatements, the MLP kernels in THEIR summation order) and compares every output
tensor.  Because each comparison starts from the GPU's own inputs nothing cascades: a mismatch names the kernel, the call
and the element.  Index outputs, selections, copies and every MLP kernel this build owns must be BIT-EXACT; the one
tolerance (canonical RoI coordinates, 2e-5) is the f32 sincos of two libraries and is stated where it applies.

What is not shadowed here: NO layer -- since the end of round 2 no library GEMM is left on the engine's path, every MLP layer
runs on a kernel of this build (packed_layer / sa_packed / sa_wide / rpn_tail / rcnn_point_mlp / sa_xyz_mlp / rows_dot) and each of
those entries is in the shadow spec below.  Only the two fused tail entries are checked elsewhere (rpn_proposals /
rcnn_postprocess: bit-identical to their torch-op formulation, test_gpu_e2e.py, which is pinned to the reference fixtures
g7 / g8).  The step runs twice: on the uniform synthetic scene of SURVEY 8d (most balls hold one point: the sparse paths of
the packed kernels) and on LiDAR-shaped scenes (synth.lidar_scene: balls near the sensor are full -- early exit of the ball
query, full tiles in the MFMA kernels, RoIs that overflow 512 points)."""
import collections
import importlib

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import ext_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_cpu(a):
    if torch.is_tensor(a):
        return a if a.device.type == "meta" else a.detach().cpu().clone()      # (meta: a shape-only index tensor, never read)
    if hasattr(a, "rowinfo"):                      # BallPack -> the oracle's stand-in keeps the index tensor
        cp = lambda t: None if t is None else t.detach().cpu().clone()
        return ext_cpu._CpuPack(a.idx.detach().cpu().clone(), cp(a.limit), cp(getattr(a, "rep", None)), cp(getattr(a, "crep", None)))
    if isinstance(a, tuple):
        return tuple(to_cpu(x) for x in a)
    if isinstance(a, list):
        return [to_cpu(x) for x in a]
    return a


class Shadow:
    """proxy around one extension module; `spec[name]` = {arg position: 'exact' | tolerance} of the outputs to compare"""

    def __init__(self, gpu, cpu, spec, log):
        self._gpu, self._cpu, self._spec, self._log = gpu, cpu, spec, log

    def __getattr__(self, name):
        fn = getattr(self._gpu, name)
        if name not in self._spec:
            return fn

        def call(*args):
            host = [to_cpu(a) for a in args]
            ret = fn(*args)
            torch.cuda.synchronize()
            check = self._spec[name]
            if callable(check):
                check(self, name, args, host, ret)
            else:
                getattr(self._cpu, name)(*host)
                for pos, mode in check.items():
                    got, want = args[pos].detach().cpu(), host[pos]
                    if mode == "exact":
                        same = torch.equal(got, want)
                        if not same:
                            bad = (got != want) & ~(torch.isnan(got) & torch.isnan(want))
                            assert not bad.any(), "%s (call %d): output %d differs in %d of %d elements, max |d| = %g" % (
                                name, self._log[name], pos, int(bad.sum()), bad.numel(),
                                float((got.double() - want.double()).abs().max()))
                    else:
                        assert float((got - want).abs().max()) <= mode, (name, pos)
            self._log[name] += 1
            self._log["elements:" + name] += sum(int(args[p].numel()) for p in (check if not callable(check) else ()))
            return ret
        return call


def check_ball_pack(self, name, args, host, pack):
    """ball_pack returns the distinct-row list: compare its header with the definition (1 + last slot != slot 0; with a
    representative map: slot 0 + every slot whose point differs from slot 0's, is below the limit and is its own representative)"""
    idx = host[0].numpy()
    keep = idx != idx[..., :1]
    if len(host) > 3 and host[3] is not None:      # copies of pooled points (index >= the cloud's distinct count) are dropped too
        lim = np.maximum(host[3].numpy().reshape(-1, 1, 1), 1)
        keep &= idx < lim
        # the contract that makes this exact: every dropped copy's original is listed in the same row
        canon = np.where(idx >= lim, idx % lim, idx)
        for b_, c_ in zip(*np.nonzero((idx >= lim).any(-1))):
            row = idx[b_, c_]
            assert set(canon[b_, c_][row >= lim[b_, 0, 0]]) <= set(row[row < lim[b_, 0, 0]])
    hdr = pack.hdr.cpu().numpy()
    if len(host) > 4 and host[4] is not None:      # copies among the points themselves (SA centres sampled from copies)
        rep = host[4].numpy()
        assert (rep <= np.arange(rep.shape[1])[None]).all() and (np.take_along_axis(rep, rep, 1) == rep).all()
        mine = np.take_along_axis(rep[:, None, :].repeat(idx.shape[1], 1), idx, 2)       # representative of every slot's point
        own = mine == idx
        # the contract: a dropped slot's representative is listed in the same row (it lies in the same ball and has a lower index)
        srt = np.sort(idx, -1)
        pos = np.minimum((srt[..., None, :] < mine[..., :, None]).sum(-1), idx.shape[-1] - 1)
        assert (np.take_along_axis(srt, pos, -1) == mine).all()
        cnt = 1 + (keep & own)[..., 1:].sum(-1)
        self._log["rep_rows_dropped"] += int(((keep & ~own)[..., 1:]).sum())
    else:
        last = np.where(keep, np.arange(idx.shape[-1]), 0).max(-1)
        cnt = last + 1
    if len(host) > 5 and host[5] is not None:      # centres that copy an earlier centre get no rows at all
        crep = host[5].numpy()
        assert (crep <= np.arange(crep.shape[1])[None]).all() and (np.take_along_axis(crep, crep, 1) == crep).all()
        # the contract: a dropped centre has the coordinates of its representative (same ball, same pooled output)
        nx = host[2].numpy()
        assert np.array_equal(np.take_along_axis(nx, crep[..., None].repeat(3, -1).astype(np.int64), 1), nx)
        self._log["centres_skipped"] += int((crep != np.arange(crep.shape[1])[None]).sum())
        cnt = np.where(crep == np.arange(crep.shape[1])[None], cnt, 0)
    assert hdr[1] == cnt.sum(), (name, int(hdr[1]), int(cnt.sum()))
    if pack.tilecloud is None:                     # a list whose rows carry their cloud (prcnn_rcnn_roi_geometry_packs): no tile count, and
        info = pack.rowinfo[:int(hdr[1])].cpu().numpy().astype(np.int64)      # every cloud's rows in one block, as many as the definition says
        clouds = info >> 16
        assert np.array_equal(np.bincount(clouds, minlength=cnt.shape[0]), cnt.sum(-1)), name
        assert (np.diff(np.nonzero(np.diff(clouds))[0]).size == 0) or len(set(clouds[np.r_[True, np.diff(clouds) != 0]])) == len(np.unique(clouds)), name
        self._log["row_cloud_lists"] += 1
        return
    assert hdr[0] == sum((int(c.sum()) + 63) // 64 for c in cnt)


def check_sa_packed(self, name, args, host, ret):
    """the fused SA kernel over a row list: every centre that has rows == the oracle over all nsample rows, bit for bit; a centre
    that copies an earlier one (crep) has none -- its output row must still be what the caller put there (zero)"""
    self._cpu.sa_packed_mlp_wrapper(*host)
    got, want = args[9].detach().cpu(), host[9]
    pack = args[4]
    if getattr(pack, "crep", None) is None:
        assert torch.equal(got, want), name
        return
    crep = pack.crep.cpu()
    own = crep == torch.arange(crep.shape[1]).view(1, -1)
    c0 = host[10]
    width = host[7].shape[1]
    assert torch.equal(got[own], want[own]), name
    # (untouched = the caller's zero; the one exception is the cloud's LAST centre, which also owns the rows that pad the cloud's last
    # tile -- copies of its own first row, so it may hold the max over that one row: between 0 and the true value.  Nobody reads it.)
    skipped, truth = got[~own][:, c0:c0 + width], want[~own][:, c0:c0 + width]
    assert bool(((skipped >= 0) & (skipped <= truth)).all()), name
    # ... and what the skipped centres WOULD have produced is exactly their representative's row (why skipping is exact)
    repl = torch.gather(want, 1, crep.long().unsqueeze(-1).expand(-1, -1, want.shape[2]))
    assert torch.equal(repl, want), name


def check_roi_geometry(self, name, args, host, ret):
    """the fused geometry of the RoI clouds (FPS, limited ball query, representative map, twice) == the chain of oracle stand-ins,
    all six outputs bit for bit"""
    want = self._cpu.rcnn_roi_geometry_wrapper(*host)
    for k, (g, w) in enumerate(zip(ret, want)):
        assert torch.equal(g.cpu(), w), (name, k)


def check_roi_geometry_packs(self, name, args, host, ret):
    """... with both levels' row lists out of the same launch: the six geometry outputs as above, and each list's header against the
    definition of ball_pack (the rows themselves are checked through the MLP kernels that consume them: check_sa_packed)"""
    want = self._cpu.rcnn_roi_geometry_wrapper(*host[:8])
    no_idx = len(host) > 10 and host[10] is False           # the product's call: the index tensors are not written (shape-only stand-ins)
    for k, (g, w) in enumerate(zip(ret[:6], want)):
        if no_idx and k in (1, 4):
            assert tuple(g.shape) == tuple(w.shape) and g.device.type == "meta", (name, k)   # a shape on the "meta" device
            continue
        assert torch.equal(g.cpu(), w), (name, k)
    new1, idx1, rep1, new2, idx2, rep2 = want
    if no_idx:      # the oracle's stand-ins of the MLP kernels restate the levels from the index tensors: hand them the oracle's own
        ret[6].idx, ret[7].idx = idx1.to(ret[0].device), idx2.to(ret[0].device)
        self._log["roi_idx_not_written"] += 1
    check_ball_pack(self, name, None, [idx1, host[0], new1, host[1], None, rep1], ret[6])
    check_ball_pack(self, name, None, [idx2, new1, new2, None, rep1, rep2], ret[7])
    if len(ret) > 8 and hasattr(ret[8], "rowinfo"):      # the GroupAll level's list: one group per RoI (all its level-2 centres around the origin), copies marked by rep2
        b, m2 = rep2.shape
        ga = torch.arange(m2, dtype=torch.int32).view(1, 1, m2).expand(b, 1, m2).contiguous()
        assert torch.equal(ret[8].idx.cpu(), ga) and torch.equal(ret[8].rep.cpu(), rep2)
        check_ball_pack(self, name, None, [ga, new2, torch.zeros((b, 1, 3)), None, rep2, None], ret[8])
        self._log["group_all_list_fused"] += 1
    crows = next((x for x in ret[8:] if isinstance(x, tuple)), None)
    if crows is not None:   # the level-1 centres that are their own representatives, as rows: exactly those with rep1[c] == c, each once
        n = int(crows[1][1])
        got_rows = torch.sort(crows[0][:n].cpu().long())[0]
        own = (rep1 == torch.arange(rep1.shape[1]).view(1, -1)).nonzero()
        assert torch.equal(got_rows, own[:, 0] * rep1.shape[1] + own[:, 1]), name
        self._log["centre_rows_listed"] += 1


def check_dup_rep(self, name, args, host, ret):
    want = self._cpu.dup_rep_wrapper(*host)
    assert torch.equal(ret.cpu(), want), name


def check_forward_canonical(self, name, args, host, ret):
    """roipool3d_canonical = enlarge + the reference's RoI pooling + canonical transform + row layout: selection, features,
    mask, depth and the empty flags bit-exact vs the oracle's pooling; coordinates within 2e-5 (sinf / cosf of the heading
    come from two libraries; |coordinates| <= ~80 m)."""
    from oracle import oracle as O
    xyz, rois, feats, mask, depth, extra = (h if not torch.is_tensor(h) else h.numpy() for h in host[:6])
    pooled, empty = args[6].detach().cpu().numpy(), args[7].detach().cpu().numpy()
    pcnt = args[8].detach().cpu().numpy() if len(args) > 8 and args[8] is not None else None
    big = rois.copy()
    big[:, :, 3:6] += np.float32(extra * 2)       # kitti_utils.enlarge_box3d
    big[:, :, 1] += np.float32(extra)
    S = pooled.shape[2]
    ref_in = np.concatenate([mask[..., None], depth[..., None], feats], axis=2)
    want, wempty = O.roipool3d(xyz, big, ref_in, S)
    assert np.array_equal(empty, wempty)
    assert np.array_equal(pooled[..., 3:5], want[..., 3:5])
    if pcnt is None:
        assert np.array_equal(pooled[..., 8:], want[..., 5:])
    else:
        # distinct rows per box = min(#points inside, S), at least 1; feature columns are written up to the next multiple of 64
        inside = np.stack([O.pts_in_boxes3d(xyz[b_], big[b_]).sum(1) for b_ in range(xyz.shape[0])])
        assert np.array_equal(pcnt, np.maximum(np.minimum(inside, S), 1))
        live = np.arange(S)[None, None, :] < ((pcnt + 63) // 64 * 64)[..., None]
        assert np.array_equal(pooled[..., 8:][live], want[..., 5:][live])
        # rows beyond the distinct count ARE copies of row s % cnt (what the consumers rely on)
        s_idx = np.arange(S)[None, None, :] % pcnt[..., None]
        assert np.array_equal(want[..., 0:5], np.take_along_axis(want[..., 0:5], s_idx[..., None].repeat(5, -1), 2))
    assert (pooled[..., 5:8] == 0).all()
    rel = want[..., 0:3].astype(np.float64) - rois[:, :, None, 0:3]
    ca, sa = np.cos(rois[:, :, None, 6].astype(np.float64)), np.sin(rois[:, :, None, 6].astype(np.float64))
    rx = rel[..., 0] * ca - rel[..., 2] * sa
    rz = rel[..., 0] * sa + rel[..., 2] * ca
    canon = np.stack([rx, rel[..., 1], rz], -1)
    canon[wempty.astype(bool)] = canon[wempty.astype(bool)] * 0 + np.stack(
        [(-rois[..., 0] * ca[..., 0] + rois[..., 2] * sa[..., 0]), -rois[..., 1], (-rois[..., 0] * sa[..., 0] - rois[..., 2] * ca[..., 0])],
        -1)[wempty.astype(bool)][:, None, :]
    assert np.abs(pooled[..., 0:3] - canon).max() < 2e-5
    if len(args) > 10 and args[10] is not None:          # round 4: the same coordinates once more as dense clouds
        assert np.array_equal(args[10].detach().cpu().numpy(), pooled[..., 0:3])


POINTNET2 = {
    "furthest_point_sampling_wrapper": {4: "exact", 5: "exact"},
    "ball_query_wrapper": {7: "exact"},
    "ball_query_full_wrapper": {7: "exact"},       # round 4: every slot written by the kernel (no zero fill by the caller)
    "ball_query_limit_wrapper": {8: "exact"},
    "three_nn_wrapper": {5: "exact", 6: "exact"},
    "three_nn_weights_wrapper": {5: "exact", 6: "exact"},      # round 4: neighbours + inverse-distance weights from one kernel
    "three_interpolate_pm_wrapper": {3: "exact"},
    "three_interpolate_cat_pm_wrapper": {4: "exact"},            # interpolation + concat with the skip features (FP modules 3-1, PRCNN_NO_FP_LINEAR=1)
    "packed_layer_interp_wrapper": {4: "exact"},                 # FP modules 3-1, first layer with the interpolated coarse product in its epilogue
    "ball_pack_wrapper": check_ball_pack,
    "sa_xyz_mlp_wrapper": {9: "exact"},
    "sa_xyz_mlp_packed_wrapper": {9: "exact"},
    "sa_packed_mlp_wrapper": None,                 # filled below (centres without rows)
    "packed_gather_affine_wrapper": None,          # filled below: compared through the layers that consume it
    "packed_layer_wrapper": None,
    "packed_layer_segmax_wrapper": {6: "exact"},
    "rows_dot_wrapper": {3: "exact"},
    "rpn_tail_wrapper": {7: "exact", 8: "exact", 9: "exact"},     # finest FP module + both heads in one kernel (PRCNN_NO_FP_LINEAR=1)
    "rpn_tail_lin_wrapper": {7: "exact", 8: "exact", 9: "exact"}, # ... its first layer applied at the coarse level
    "rpn_tail_lin_boxes_wrapper": {14: "exact", 15: "exact", 16: "exact"},   # ... and the proposal layer's decode inside (round 5): features, scores, BOXES
    "rcnn_point_mlp_wrapper": None,                # filled below
}


def check_packed_rows(pos_out):
    """outputs laid out per PACKED row on the GPU and per (group, slot) on the CPU: compare through the row list"""
    def check(self, name, args, host, ret):
        pack = next(a for a in args if hasattr(a, "rowinfo"))
        cpu_pack = next(h for h in host if isinstance(h, ext_cpu._CpuPack))
        getattr(self._cpu, name)(*host)
        got, want = args[pos_out].detach().cpu(), host[pos_out]
        b, m, ns = cpu_pack.idx.shape
        tiles = int(pack.hdr[0])
        info = pack.rowinfo.cpu().numpy().view(np.uint32)[:tiles * 64].astype(np.int64)
        cloud = np.repeat(pack.tilecloud.cpu().numpy()[:tiles].astype(np.int64), 64)
        centre, point = info >> 16, info & 0xffff
        idx = cpu_pack.idx.numpy()
        # slot of `point` in its group's index row (first occurrence)
        rows_idx = idx[cloud, centre]                                        # (rows, ns)
        slot = (rows_idx == point[:, None]).argmax(1)
        assert (rows_idx[np.arange(len(slot)), slot] == point).all()
        src = torch.from_numpy((cloud * m + centre) * ns + slot)
        assert torch.equal(got[:tiles * 64], want[src]), name
    return check


POINTNET2["packed_gather_affine_wrapper"] = check_packed_rows(5)


def check_packed_layer(self, name, args, host, ret):
    """a layer is row-wise: the oracle evaluates the SAME rows (the GPU's packed list, or all rows of a per-point layer)"""
    a, wt, bias, relu, out = host[:5]
    rows = a.shape[0]
    if len(args) > 5 and args[5] is not None:                               # over a packed row list: hdr[0] tiles are live
        rows = int(args[5].hdr[0]) * 64
    want = torch.empty((rows, out.shape[1]))                                 # (a narrow last layer stores fewer than N columns)
    self._cpu.packed_layer_wrapper(a[:rows], wt, bias, relu, want)
    assert torch.equal(args[4].detach().cpu()[:rows], want), name


POINTNET2["packed_layer_wrapper"] = check_packed_layer


def check_rcnn_point_mlp(self, name, args, host, ret):
    """entrance chain: with a tile list only the tiles holding distinct pooled rows are computed -- compare those"""
    self._cpu.rcnn_point_mlp_wrapper(*host[:13])
    rows = host[0].shape[0]
    live = torch.ones(rows, dtype=torch.bool)
    if len(args) > 13 and args[13] is not None:
        tilemap, hdr = args[13]
        n = int(hdr[0])
        live[:] = False
        tiles = tilemap[:n].cpu().long()
        assert len(torch.unique(tiles)) == n
        live.view(-1, 64)[tiles] = True
        self._log["live_rows_fraction_x1000"] = int(1000 * n * 64 / rows)
    for pos in (10, 11, 12):
        if args[pos] is None:                                                  # fused form: only p (position 12) is produced
            continue
        got, want = args[pos].detach().cpu(), host[pos]
        assert torch.equal(got[live], want[live]), (name, pos)


POINTNET2["rcnn_point_mlp_wrapper"] = check_rcnn_point_mlp


def check_rcnn_point_mlp_rows(self, name, args, host, ret):
    """the entrance chain over the list of distinct pooled rows: the listed rows == the oracle's rows, bit for bit; the list holds the
    first max(count, 1) rows of every RoI exactly once (the counts are what the RoI pooling handed over: checked there)"""
    self._cpu.rcnn_point_mlp_rows_wrapper(*host[:11], None)
    rowmap, hdr = args[11]
    n = int(hdr[1])
    listed = rowmap[:n].cpu().long()
    assert len(torch.unique(listed)) == n
    got, want = args[10].detach().cpu(), host[10]
    assert torch.equal(got[listed], want[listed]), name
    self._log["live_rows_fraction_x1000"] = int(1000 * n / host[0].shape[0])


POINTNET2["rcnn_point_mlp_rows_wrapper"] = check_rcnn_point_mlp_rows


def check_rows_gemm128_rows(self, name, args, host, ret):
    """a 128-wide layer over a list of rows: the listed rows == the oracle's, bit for bit; the others untouched (the poisoned torch.empty
    of this run: NaN)"""
    a, wt, bias, relu = host[:4]
    want = self._cpu.rows_gemm128_wrapper(a, wt, bias, relu)
    rowmap, hdr = args[5]
    n = int(hdr[1])
    listed = rowmap[:n].cpu().long()
    assert len(torch.unique(listed)) == n
    got = args[4].detach().cpu()
    assert torch.equal(got[listed], want[listed]), name
    rest = torch.ones(got.shape[0], dtype=torch.bool); rest[listed] = False
    assert torch.isnan(got[rest]).all(), name
    self._log["rows_gemm_listed_x1000"] = int(1000 * n / got.shape[0])


POINTNET2["rows_gemm128_rows_wrapper"] = check_rows_gemm128_rows


def check_packed_segmax(self, name, args, host, ret):
    """last layer + pool: the oracle evaluates the layer on the GPU's packed rows and pools them by centre"""
    a, wt, bias, _, b, m, _, out_col = host[:8]
    pack = args[3]
    tiles = int(pack.hdr[0])
    y = torch.empty((tiles * 64, wt.shape[1]))
    self._cpu.packed_layer_wrapper(a[:tiles * 64], wt, bias, True, y)
    info = pack.rowinfo.cpu().numpy().view(np.uint32)[:tiles * 64].astype(np.int64)
    cloud = np.repeat(pack.tilecloud.cpu().numpy()[:tiles].astype(np.int64), 64)
    centre = torch.from_numpy(cloud * m + (info >> 16))
    want = torch.zeros((b * m, wt.shape[1]))
    want.scatter_reduce_(0, centre.view(-1, 1).expand(-1, wt.shape[1]), y, reduce="amax", include_self=True)
    got = args[6].detach().cpu().view(b * m, -1)[:, out_col:out_col + wt.shape[1]]
    assert torch.equal(got, want), name
    assert len(torch.unique(centre)) == b * m                              # every centre owns at least one row


POINTNET2["packed_layer_segmax_wrapper"] = check_packed_segmax


def check_fps_new_xyz(self, name, args, host, ret):
    want_idx, want_xyz = self._cpu.fps_new_xyz_wrapper(*host)
    assert torch.equal(ret[0].cpu(), want_idx) and torch.equal(ret[1].cpu(), want_xyz), name


POINTNET2["fps_new_xyz_wrapper"] = check_fps_new_xyz


def check_point_aux(self, name, args, host, ret):
    """seg / depth / depth_norm of every point in one launch: depth and depth_norm bit for bit (sqrt, divide, subtract are correctly rounded
    on both sides); the foreground flag wherever the sigmoid is not within 2 ulp of the threshold (expf of two libraries)"""
    scores, xyz, thresh = host[0], host[1], host[2]
    seg, depth, dn = (torch.empty_like(scores) for _ in range(3))
    self._cpu.point_aux_wrapper(scores, xyz, thresh, seg, depth, dn)
    assert torch.equal(args[4].cpu(), depth) and torch.equal(args[5].cpu(), dn), name
    sg = torch.sigmoid(scores.double())
    decided = (sg - thresh).abs() > 1e-6
    assert torch.equal(args[3].cpu()[decided], seg[decided]), name


POINTNET2["point_aux_wrapper"] = check_point_aux
POINTNET2["dup_rep_wrapper"] = check_dup_rep
POINTNET2["rcnn_roi_geometry_wrapper"] = check_roi_geometry
POINTNET2["rcnn_roi_geometry_packs_wrapper"] = check_roi_geometry_packs
POINTNET2["sa_packed_mlp_wrapper"] = check_sa_packed
POINTNET2["sa_wide_fused_wrapper"] = {9: "exact"}          # one scale of a wide level in one kernel: output slice vs the oracle chain
POINTNET2["sa_wide_fused3_wrapper"] = {10: "exact"}        # ... with the per-point layer inside (the RCNN's GroupAll level)


def batched(check):
    """a batched wrapper takes ONE list of problems: every problem is checked like a call of the single-problem wrapper"""
    def run(self, name, args, host, ret):
        assert 1 <= len(args[0]) <= 4
        for prob, hprob in zip(args[0], host[0]):
            check(self, name.replace("_batch", ""), list(prob), list(hprob), None)
    return run


POINTNET2["packed_layer_batch_wrapper"] = batched(check_packed_layer)
def check_sa_packed_slice(self, name, args, host, ret):
    """one problem of sa_packed_mlp_batch_wrapper (both scales of RPN SA2 in one launch, round 5): the problems share the output
    tensor, each owns a column slice -- the oracle over all nsample rows fills its slice of a host copy, compared bit for bit"""
    self._cpu.sa_packed_mlp_batch_wrapper([tuple(host)])   # host[12]: the real widths under the padding (the padded chain is the definition; a 64-wide P is padded back)
    c0, width = host[10], host[7].shape[1]
    got, want = args[9].detach().cpu()[..., c0:c0 + width], host[9][..., c0:c0 + width]
    assert getattr(args[4], "crep", None) is None
    assert torch.equal(got, want), name
    assert float(want.abs().max()) > 0


POINTNET2["sa_packed_mlp_batch_wrapper"] = batched(check_sa_packed_slice)
POINTNET2["packed_gather_affine_batch_wrapper"] = batched(POINTNET2["packed_gather_affine_wrapper"])
POINTNET2["packed_layer_segmax_batch_wrapper"] = batched(check_packed_segmax)


def spread_heads(model):
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ("reg_layer" in name or "cls_layer" in name) and p.dim() > 1:
                p.copy_((torch.randn(p.shape, generator=g) * 0.3).to(p.device))
        model.rcnn_net.cls_layer[-1].conv.weight.mul_(0.05)
        model.rcnn_net.cls_layer[-1].conv.bias.fill_(0.5)


@pytest.mark.parametrize("wide_fused,scene_kind", [(True, "uniform"), (False, "uniform"), (True, "lidar")])
def test_batch8_step_every_kernel_call_equals_the_oracle(wide_fused, scene_kind, monkeypatch):
    """wide_fused = False: the RCNN's GroupAll level layer by layer (gather + affine, layer, layer + pool) like the RPN's wide levels,
    instead of the one-kernel form of csrc/sa_wide.hip.  scene_kind = "lidar": LiDAR-shaped scenes (dense near the sensor)."""
    C, E, F, S = pkg("config"), pkg("eval_rcnn"), pkg("net.fast_infer"), pkg("synth")
    monkeypatch.setattr(F, "USE_WIDE_FUSED", wide_fused)
    pu, ru = pkg("pointnet2.pointnet2_utils"), pkg("roipool3d_utils")
    cfg = C.default_eval_cfg()
    model = E.build_model(cfg, DEV, seed=3)
    spread_heads(model)
    eng = F.FastPointRCNN(model, cfg)
    B = 8
    make = S.lidar_scenes if scene_kind == "lidar" else S.scenes
    pts = torch.from_numpy(make(B, cfg.RPN.NUM_POINTS, seed0=77)).to(DEV)
    plain = E.infer_batch(model, cfg, pts, engine=eng)                        # un-instrumented run
    log = collections.Counter()
    saved = (pu.pointnet2, ru.roipool3d_cuda)
    pu.pointnet2 = Shadow(saved[0], ext_cpu.pointnet2_cpu, POINTNET2, log)
    ru.roipool3d_cuda = Shadow(saved[1], None, {"forward_canonical": check_forward_canonical}, log)
    real_empty = torch.empty

    def poisoned(*a, **k):                                                    # unwritten tiles / rows surface as NaN
        t = real_empty(*a, **k)
        if t.is_cuda and t.is_floating_point():
            t.fill_(float("nan"))
        return t
    torch.empty = poisoned
    try:
        det = E.infer_batch(model, cfg, pts, engine=eng)
    finally:
        torch.empty = real_empty
        pu.pointnet2, ru.roipool3d_cuda = saved
    # the instrumented run IS the product run: identical detections, all 8 scenes finite and populated
    for k in ("rois", "rcnn_cls", "rcnn_reg", "boxes", "scores", "num"):
        assert torch.equal(det[k], plain[k]), k
        assert torch.isfinite(det[k].float()).all(), k
    assert (det["num"] > 0).all()
    # coverage: every kernel family of the step was exercised at the batch-8 shapes
    fg = F.USE_ROI_GEOMETRY          # the RoI clouds' FPS / ball query / representative maps of both sampled levels in one launch
    fp = fg and F.USE_ROI_PACKS      # ... and their two row lists out of that launch
    want_calls = {"furthest_point_sampling_wrapper": 0, "fps_new_xyz_wrapper": 4 if fg else 6, "dup_rep_wrapper": 0 if fg else 2, "point_aux_wrapper": 1,
                  "ball_query_full_wrapper": 8, "ball_query_wrapper": 0 if fg else 1, "ball_query_limit_wrapper": 0 if fg else 1, "rcnn_roi_geometry_wrapper": 1 if (fg and not fp) else 0, "rcnn_roi_geometry_packs_wrapper": 1 if fp else 0, "three_nn_wrapper": 0, "three_nn_weights_wrapper": 4, "ball_pack_wrapper": (8 if log["group_all_list_fused"] else 9) if fp else 11,
                  "sa_xyz_mlp_packed_wrapper": 2, "sa_packed_mlp_wrapper": 2 if (F.USE_SCALE_BATCH and F.USE_SA2_BATCH) else 4, "sa_packed_mlp_batch_wrapper": 1 if (F.USE_SCALE_BATCH and F.USE_SA2_BATCH) else 0,
                  "three_interpolate_cat_pm_wrapper": 0 if F.USE_FP_LINEAR else 3, "packed_layer_interp_wrapper": 3 if F.USE_FP_LINEAR else 0,
                  "rpn_tail_wrapper": 0 if F.USE_FP_LINEAR else 1, "rpn_tail_lin_wrapper": 1 if F.USE_FP_LINEAR and not F.USE_TAIL_DECODE else 0,
                  "rpn_tail_lin_boxes_wrapper": 1 if F.USE_FP_LINEAR and F.USE_TAIL_DECODE else 0, "rcnn_point_mlp_wrapper": 0 if F.USE_POOLED_ROWS else 1, "rcnn_point_mlp_rows_wrapper": 1 if F.USE_POOLED_ROWS else 0, "forward_canonical": 1,
                  "rows_gemm128_rows_wrapper": 1 if (fp and F.USE_CENTRE_ROWS) else 0}
    want_calls.update({"packed_layer_segmax_batch_wrapper": 2, "packed_gather_affine_batch_wrapper": 2, "packed_layer_batch_wrapper": 6 if (F.USE_SCALE_BATCH and F.USE_SA2_BATCH and not F.USE_SA_NARROW) else 5})   # RPN SA3, SA4; the two branches of the RCNN head (round 4); RPN SA2's per-point parts (round 5; one plain product for both scales with the narrow kernel)
    if wide_fused:       # the RCNN's GroupAll level (every row distinct: 800 units of work) in one kernel
        f3 = 1 if F.USE_WIDE_FUSED3 else 0   # ... and its per-point layer inside that kernel (csrc/sa_wide3.hip)
        want_calls.update({"sa_wide_fused3_wrapper": f3, "sa_wide_fused_wrapper": 1 - f3, "packed_layer_segmax_wrapper": 0, "packed_gather_affine_wrapper": 0})
    else:
        want_calls.update({"sa_wide_fused3_wrapper": 0, "sa_wide_fused_wrapper": 0, "packed_layer_segmax_wrapper": 1, "packed_gather_affine_wrapper": 1})
    for name, n in want_calls.items():
        assert log[name] == n, (name, log[name], n)
    assert log["packed_layer_wrapper"] >= 3 and log["rows_dot_wrapper"] == 1
    assert log["roi_idx_not_written"] == (1 if fp else 0) and log["row_cloud_lists"] == ((3 if wide_fused and F.USE_WIDE_FUSED3 else 2) if fp else 0)
    assert log["rep_rows_dropped"] > 1000            # the deeper RCNN levels really dropped rows of copied centres
    assert log["centres_skipped"] > 1000             # ... and skipped the centres that copy an earlier one
    print("shadowed calls:", {k: v for k, v in log.items() if not k.startswith("elements:")})
