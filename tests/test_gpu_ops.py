"""-m gpu parity tests: every HIP operator, called through the drop-in extension modules (i.e.
through the C ABI of libprcnn_hip.so), against the CPU oracle on the same seeded inputs.
Index outputs must be bit-exact; float outputs of pure copies/selects bit-exact as well;
geometric float outputs within the tolerance stated next to the assert."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from helpers import scene, scenes, bev_boxes, boxes3d

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def fps_gpu(ext, xyz, m):
    b, n, _ = xyz.shape
    t = T(xyz)
    temp = torch.full((b, n), 1e10, device=DEV)
    idx = torch.empty((b, m), dtype=torch.int32, device=DEV)
    ext.pointnet2.furthest_point_sampling_wrapper(b, n, m, t, temp, idx)
    return idx.cpu().numpy(), temp.cpu().numpy()


@pytest.mark.parametrize("n,m", [(128, 32), (512, 128), (1000, 100), (1024, 256), (4096, 1024), (16384, 4096)])
def test_fps_matches_oracle(ext, oracle, n, m):
    xyz = scenes(2, n, seed0=n)
    got, gtemp = fps_gpu(ext, xyz, m)
    want, wtemp = oracle.furthest_point_sample(xyz, m, return_temp=True)
    assert np.array_equal(got, want)
    assert np.array_equal(gtemp, wtemp)


def test_fps_tie_rule_lattice_and_duplicates(ext, oracle):
    # integer lattice: many exactly equal distances -> the block-size dependent tie rule decides
    g = np.stack(np.meshgrid(np.arange(16), np.arange(8), np.arange(16), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(5)
    lat = g[rng.permutation(len(g))].astype(np.float32)[None]            # (1, 2048, 3)
    dup = np.repeat(scene(3, 256), 4, axis=0)[None]                     # every point 4 times
    dup = dup[:, rng.permutation(dup.shape[1])]
    for xyz, m in ((lat, 512), (dup, 300), (lat[:, :700], 128)):
        got, _ = fps_gpu(ext, xyz, m)
        assert np.array_equal(got, oracle.furthest_point_sample(xyz, m))


@pytest.mark.parametrize("b,n,m", [(2, 32768, 4096), (1, 16385, 4096), (9, 20000, 300), (3, 32768, 8192), (17, 24576, 256)])
def test_fps_two_workgroups_per_cloud(ext, oracle, b, n, m):
    """16384 < n <= 32768 (tools/cfgs/double.yaml: NUM_POINTS 32768): fps_spec2_kernel -- the cloud's halves in the registers of TWO
    workgroups that run the speculative rounds in lockstep over a table in global memory (round 5) -- gives the oracle's picks and
    running minima bit for bit: uniform scenes, a cloud count that does not fill the last set of 16 blocks, a ragged second half
    (n = 16385: the second workgroup holds ONE point), LiDAR-shaped density."""
    S = pkg("synth")
    xyz = scenes(b, n, seed0=n + m) if b != 3 else np.stack([S.lidar_scene(40 + i, n) for i in range(b)], 0)
    got, gtemp = fps_gpu(ext, xyz, m)
    want, wtemp = oracle.furthest_point_sample(xyz, m, return_temp=True)
    assert np.array_equal(got, want)
    assert np.array_equal(gtemp, wtemp)


def test_fps_two_workgroups_tie_rule_lattice_and_duplicates(ext, oracle):
    """the same kernel on clouds full of EXACT ties: a 32 x 16 x 48 integer lattice (24576 points) and a cloud whose every point occurs
    four times -- the (bitrev(k mod bs), k div bs) tie rule decides thousands of picks, across the two workgroups' tables"""
    g = np.stack(np.meshgrid(np.arange(32), np.arange(16), np.arange(48), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(6)
    lat = g[rng.permutation(len(g))].astype(np.float32)[None]            # (1, 24576, 3)
    dup = np.repeat(scene(4, 8192), 4, axis=0)[None]                    # every point 4 times: 32768
    dup = dup[:, rng.permutation(dup.shape[1])]
    for xyz, m in ((lat, 2048), (dup, 1000), (np.concatenate([lat, lat[:, ::-1]], 0), 600)):
        got, _ = fps_gpu(ext, xyz, m)
        assert np.array_equal(got, oracle.furthest_point_sample(xyz, m))


@pytest.mark.parametrize("capacity,what", [("64", "two launches of 16 + 4 clouds"), ("0", "fps_generic_kernel")])
def test_fps_two_workgroups_respects_the_co_resident_capacity(tmp_path, oracle, capacity, what):
    """the two-workgroup kernel spins on its partner, so the host side launches at most HALF of what the device holds at
    once (PRCNN_FPS2_CAPACITY overrides the occupancy query: 64 slots -> 16 clouds per launch, a batch of 20 goes as two launches over
    offset pointers) and falls back to fps_generic_kernel where the kernel cannot be co-resident at all (capacity 0 = a part that
    refuses 84 KB of LDS); picks and running minima stay the oracle's.  A child process: the capacity is read once per device."""
    import subprocess, sys
    xyz = scenes(20, 17000, seed0=3)
    np.save(tmp_path / "xyz.npy", xyz)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from conftest import pkg\nimport importlib\n"
            "p = pkg(); sys.path.insert(0, p.DROPIN_DIR); import pointnet2_cuda\n"
            "xyz = np.load(%r); t = torch.from_numpy(xyz).cuda()\n"
            "temp = torch.full(xyz.shape[:2], 1e10, device='cuda'); idx = torch.empty((xyz.shape[0], 300), dtype=torch.int32, device='cuda')\n"
            "pointnet2_cuda.furthest_point_sampling_wrapper(xyz.shape[0], xyz.shape[1], 300, t, temp, idx); torch.cuda.synchronize()\n"
            "np.savez(%r, idx=idx.cpu().numpy(), temp=temp.cpu().numpy())\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               str(tmp_path / "xyz.npy"), str(tmp_path / "out.npz")))
    env = dict(os.environ, PRCNN_FPS2_CAPACITY=capacity)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (what, r.stdout[-1500:], r.stderr[-1500:])
    got = np.load(tmp_path / "out.npz")
    want, wtemp = oracle.furthest_point_sample(xyz, 300, return_temp=True)
    assert np.array_equal(got["idx"], want), what
    assert np.array_equal(got["temp"], wtemp), what


def test_fps_many_small_clouds(ext, oracle):
    xyz = np.random.default_rng(1).uniform(-1, 1, (300, 512, 3)).astype(np.float32)
    got, _ = fps_gpu(ext, xyz, 128)
    assert np.array_equal(got, oracle.furthest_point_sample(xyz, 128))


def centres(oracle, xyz, m):
    idx = oracle.furthest_point_sample(xyz, m).astype(np.int64)
    return np.take_along_axis(xyz, idx[..., None].repeat(3, -1), 1)


@pytest.mark.parametrize("n,m,r,ns", [(16384, 4096, 0.1, 16), (16384, 4096, 0.5, 32), (16384, 4096, 0.4, 64),
                                      (4096, 1024, 1.0, 32), (1024, 256, 2.0, 32), (256, 64, 4.0, 32),
                                      (1000, 77, 0.7, 20)])
def test_ball_query_matches_oracle(ext, oracle, n, m, r, ns):
    xyz = scenes(2, n, seed0=7)
    new_xyz = centres(oracle, xyz, m)
    new_xyz[0, 0] = [500, 500, 500]  # an empty ball
    idx = torch.full((2, m, ns), -7, dtype=torch.int32, device=DEV)
    ext.pointnet2.ball_query_wrapper(2, n, m, r, ns, T(new_xyz), T(xyz), idx)
    want = np.full((2, m, ns), -7, np.int32)
    oracle.ball_query_into(r, ns, xyz, new_xyz, want)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert (want[0, 0] == -7).all()  # empty ball: row untouched


@pytest.mark.parametrize("n,m,r,ns", [(16384, 4096, 0.1, 16), (16384, 4096, 0.2, 32), (16384, 4096, 0.5, 32),
                                      (8192, 1000, 1.0, 64), (4096, 1024, 3.0, 16), (16384, 300, 0.05, 8)])
def test_ball_query_grid_equals_brute_force(ext, oracle, n, m, r, ns):
    """The bucket-sorted grid with a wave per centre (mode 0, automatic for n >= 2048: csrc/ball_dense.hip), the brute-force
    scan (mode 1) and round 1's linked-list grid (mode 2) return the same bits, also on a cloud with duplicated points,
    far-away centres and coordinates on cell boundaries."""
    import importlib
    lib = importlib.import_module("3d_adapt_auto_driving_amd._lib")
    xyz = scenes(2, n, seed0=n + m)
    xyz[1, n // 2:] = xyz[1, :n // 2]                                  # duplicates
    xyz[0, :64, 0] = np.round(xyz[0, :64, 0] / np.float32(r * 1.001)) * np.float32(r * 1.001)   # on cell edges
    new_xyz = centres(oracle, xyz, m)
    new_xyz[0, 1] = [1e6, 0, -1e6]
    res = []
    for mode in (0, 1, 2):
        lib.call("prcnn_set_ball_query_mode", mode)
        idx = torch.full((2, m, ns), -3, dtype=torch.int32, device=DEV)
        ext.pointnet2.ball_query_wrapper(2, n, m, r, ns, T(new_xyz), T(xyz), idx)
        res.append(idx.cpu().numpy())
    lib.call("prcnn_set_ball_query_mode", 0)
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[2], res[1])
    want = np.full((2, m, ns), -3, np.int32)
    oracle.ball_query_into(r, ns, xyz, new_xyz, want)
    assert np.array_equal(res[0], want)


def test_ball_query_rcnn_shape(ext, oracle):
    rng = np.random.default_rng(2)
    xyz = rng.uniform(-2.5, 2.5, (200, 512, 3)).astype(np.float32)
    new_xyz = centres(oracle, xyz, 128)
    idx = torch.zeros((200, 128, 64), dtype=torch.int32, device=DEV)
    ext.pointnet2.ball_query_wrapper(200, 512, 128, 0.2, 64, T(new_xyz), T(xyz), idx)
    assert np.array_equal(idx.cpu().numpy(), oracle.ball_query(0.2, 64, xyz, new_xyz))


@pytest.mark.parametrize("n,m", [(512, 128), (128, 32), (100, 100), (1024, 37), (1, 1)])
def test_fps_with_coordinates_output(ext, oracle, n, m):
    """prcnn_fps_new_xyz (no distance scratch from the caller, coordinates of the selection written by the kernel) == the
    reference sequence furthest_point_sample + gather on wrapped RoI clouds (many exact ties: copies of the same point)."""
    rng = np.random.default_rng(n * 7 + m)
    b = 50
    xyz = rng.uniform(-1, 1, (b, n, 3)).astype(np.float32)
    for i in range(0, b, 3):                                              # a third of the clouds: few distinct points, wrapped
        c = int(rng.integers(1, max(2, n // 4)))
        xyz[i] = xyz[i, np.arange(n) % c]
    idx, new_xyz = ext.pointnet2.fps_new_xyz_wrapper(T(xyz), m)
    want = oracle.furthest_point_sample(xyz, m)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(new_xyz.cpu().numpy(), np.take_along_axis(xyz, want[:, :, None].astype(np.int64), axis=1))


@pytest.mark.parametrize("b,n,m", [(3, 16384, 4096), (2, 4096, 1024), (2, 16384, 100), (3, 2000, 300), (2, 1500, 64), (1, 32768, 8192), (2, 20000, 50)])
def test_fps_with_coordinates_output_every_shape(ext, oracle, b, n, m):
    """prcnn_fps_new_xyz beyond the small-cloud range: the speculative kernel writes the coordinates itself, every other shape
    (few samples, n > 16384: double.yaml's 32768 points) runs over an internal distance scratch + a gather (round 5)."""
    xyz = np.random.default_rng(n + m).uniform(-20, 20, (b, n, 3)).astype(np.float32)
    idx, new_xyz = ext.pointnet2.fps_new_xyz_wrapper(T(xyz), m)
    want = oracle.furthest_point_sample(xyz, m)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(new_xyz.cpu().numpy(), np.take_along_axis(xyz, want[:, :, None].astype(np.int64), axis=1))


def test_ball_query_with_scan_limit_on_wrapped_clouds(ext, oracle):
    """Clouds filled the way RoI pooling fills a box holding fewer than 512 points (row k >= count is a copy of row
    k % count, roipool3d_kernel.cu:152-159; an empty box is all one point).  prcnn_ball_query_limit scans the first count
    rows only: (1) it equals the reference ball query run on the truncated cloud, bit for bit; (2) per ball it names exactly
    the same SET of distinct points (rows mod count) as the reference query over all 512 rows."""
    rng = np.random.default_rng(7)
    b, n, m, r, ns = 60, 512, 128, 0.2, 64
    cnt = rng.integers(0, 200, b).astype(np.int32)
    cnt[0], cnt[1], cnt[2], cnt[3] = 0, 1, 511, 512
    xyz = np.empty((b, n, 3), np.float32)
    for i in range(b):
        c = max(int(cnt[i]), 1)
        base = rng.uniform(-0.6, 0.6, (c, 3)).astype(np.float32)
        xyz[i] = base[np.arange(n) % c]
    new_xyz = centres(oracle, xyz, m)
    got = torch.full((b, m, ns), -5, dtype=torch.int32, device=DEV)              # every slot is written by the kernel
    ext.pointnet2.ball_query_limit_wrapper(b, n, m, r, ns, T(new_xyz), T(xyz), T(cnt), got)
    got = got.cpu().numpy()
    full = oracle.ball_query(r, ns, xyz, new_xyz)
    for i in range(b):
        c = max(int(cnt[i]), 1)
        want = oracle.ball_query(r, ns, xyz[i:i + 1, :c].copy(), new_xyz[i:i + 1])
        assert np.array_equal(got[i], want[0]), i
        for ctr in range(0, m, 7):
            assert set(got[i, ctr] % c) == set(full[i, ctr] % c), (i, ctr)


@pytest.mark.parametrize("c", [0, 1, 16, 128])
def test_query_and_group_fused(ext, oracle, c):
    n, m, r, ns = 4096, 1024, 0.8, 32
    xyz = scenes(2, n, seed0=11)
    new_xyz = centres(oracle, xyz, m)
    new_xyz[1, 5] = [-500, 0, 0]
    feats = np.random.default_rng(c).standard_normal((2, c, n)).astype(np.float32) if c else None
    idx = torch.empty((2, m, ns), dtype=torch.int32, device=DEV)
    out = torch.empty((2, 3 + c, m, ns), device=DEV)
    ext.pointnet2.query_and_group_wrapper(2, n, m, c, r, ns, T(new_xyz), T(xyz), T(feats) if c else None, idx, out)
    want, widx = oracle.query_and_group(r, ns, xyz, new_xyz, feats)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("n,m,r,ns", [(16384, 4096, 0.1, 16), (16384, 4096, 0.5, 32), (16384, 4096, 0.4, 64), (16384, 4096, 2.0, 64),
                                      (4096, 1024, 0.5, 16), (4096, 1024, 1.0, 32), (2048, 512, 1.0, 32)])
def test_ball_query_on_lidar_shaped_clouds(ext, oracle, n, m, r, ns):
    """LiDAR-shaped clouds (synth.lidar_scene: hundreds of points per cell near the sensor, none far away): the regime in
    which the nsample smallest indices of a FULL ball decide, the wave's sorted insertion runs thousands of times and the
    chunk pruning of csrc/ball_dense.hip skips most of every bucket range.  Centres = the cloud's own FPS picks (as in the
    backbone), one cloud with exact duplicates, one centre far outside.  Bit-exact vs the oracle's index-order scan."""
    import importlib
    S = importlib.import_module("3d_adapt_auto_driving_amd.synth")
    xyz = np.stack([S.lidar_scene(11, 16384)[:n], S.lidar_scene(12, 16384)[:n]], 0)
    xyz[1, n // 2:] = xyz[1, :n // 2]
    new_xyz = centres(oracle, xyz, m)
    new_xyz[0, 3] = [700, 2, -700]
    idx = torch.full((2, m, ns), -5, dtype=torch.int32, device=DEV)
    ext.pointnet2.ball_query_wrapper(2, n, m, r, ns, T(new_xyz), T(xyz), idx)
    want = np.full((2, m, ns), -5, np.int32)
    oracle.ball_query_into(r, ns, xyz, new_xyz, want)
    assert np.array_equal(idx.cpu().numpy(), want)
    if r >= 0.4:
        assert (want[0, :, -1] != want[0, :, 0]).mean() > 0.1       # a good share of the balls are full (14 % at r = 0.4 / 64)


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("c", [0, 128])
@pytest.mark.parametrize("r,ns", [(0.1, 32), (0.1, 64), (0.2, 32), (0.2, 64), (0.4, 32), (0.4, 64)])
def test_query_and_group_benchmarked_shapes(ext, oracle, r, ns, c, dense):
    """BASELINE configs[1] exactly as bench.py / profiles/op_microbench.py run it: N = 16384, M = 4096 FPS centres, the six
    (r, nsample) pairs, C in {0, 128} -- the shape that selects group_cat_lds_kernel<1> and the hashed-grid ball query.
    `dense` shrinks the scene 33x so that the balls are FULL (first-nsample-by-index and the early exit decide), the
    plain scene leaves most of them nearly empty (back-fill decides)."""
    n, m = 16384, 4096
    xyz = scenes(2, n, seed0=1000)
    if dense:
        xyz = (xyz * np.float32(0.03)).astype(np.float32)
    new_xyz = centres(oracle, xyz, m)
    feats = np.random.default_rng(7).standard_normal((2, c, n)).astype(np.float32) if c else None
    idx = torch.empty((2, m, ns), dtype=torch.int32, device=DEV)
    out = torch.full((2, 3 + c, m, ns), float("nan"), device=DEV)
    ext.pointnet2.query_and_group_wrapper(2, n, m, c, r, ns, T(new_xyz), T(xyz), T(feats) if c else None, idx, out)
    want, widx = oracle.query_and_group(r, ns, xyz, new_xyz, feats)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.array_equal(out.cpu().numpy(), want)
    if dense:
        assert (widx[..., -1] != widx[..., 0]).mean() > 0.5       # most balls really are full


def test_group_and_gather(ext, oracle):
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((3, 19, 777)).astype(np.float32)
    idx = rng.integers(0, 777, (3, 50, 9)).astype(np.int32)
    out = torch.empty((3, 19, 50, 9), device=DEV)
    ext.pointnet2.group_points_wrapper(3, 19, 777, 50, 9, T(pts), T(idx), out)
    assert np.array_equal(out.cpu().numpy(), oracle.group_points(pts, idx))
    gidx = rng.integers(0, 777, (3, 123)).astype(np.int32)
    gout = torch.empty((3, 19, 123), device=DEV)
    ext.pointnet2.gather_points_wrapper(3, 19, 777, 123, T(pts), T(gidx), gout)
    assert np.array_equal(gout.cpu().numpy(), oracle.gather_points(pts, gidx))


def test_grad_kernels(ext, oracle):
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 200, (2, 40, 8)).astype(np.int32)
    go = rng.standard_normal((2, 5, 40, 8)).astype(np.float32)
    g = torch.zeros((2, 5, 200), device=DEV)
    ext.pointnet2.group_points_grad_wrapper(2, 5, 200, 40, 8, T(go), T(idx), g)
    np.testing.assert_allclose(g.cpu().numpy(), oracle.group_points_grad(go, idx, 200), rtol=1e-5, atol=1e-5)
    gidx = rng.integers(0, 200, (2, 60)).astype(np.int32)
    go2 = rng.standard_normal((2, 5, 60)).astype(np.float32)
    g2 = torch.zeros((2, 5, 200), device=DEV)
    ext.pointnet2.gather_points_grad_wrapper(2, 5, 200, 60, T(go2), T(gidx), g2)
    np.testing.assert_allclose(g2.cpu().numpy(), oracle.gather_points_grad(go2, gidx, 200), rtol=1e-5, atol=1e-5)
    i3 = rng.integers(0, 50, (2, 70, 3)).astype(np.int32)
    w3 = rng.uniform(0, 1, (2, 70, 3)).astype(np.float32)
    go3 = rng.standard_normal((2, 5, 70)).astype(np.float32)
    g3 = torch.zeros((2, 5, 50), device=DEV)
    ext.pointnet2.three_interpolate_grad_wrapper(2, 5, 70, 50, T(go3), T(i3), T(w3), g3)
    np.testing.assert_allclose(g3.cpu().numpy(), oracle.three_interpolate_grad(go3, i3, w3, 50), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,m", [(256, 64), (1024, 256), (4096, 1024), (16384, 4096), (333, 5)])
def test_three_nn_and_interpolate(ext, oracle, n, m):
    unknown = scenes(2, n, seed0=21)
    known = centres(oracle, unknown, m)
    known[0, 1] = known[0, 0]  # exact tie: lowest index must win
    d2 = torch.empty((2, n, 3), device=DEV)
    idx = torch.empty((2, n, 3), dtype=torch.int32, device=DEV)
    ext.pointnet2.three_nn_wrapper(2, n, m, T(unknown), T(known), d2, idx)
    wd2, widx = oracle.three_nn(unknown, known)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.array_equal(d2.cpu().numpy(), wd2)
    rng = np.random.default_rng(n)
    feats = rng.standard_normal((2, 24, m)).astype(np.float32)
    w = rng.uniform(0, 1, (2, n, 3)).astype(np.float32)
    out = torch.empty((2, 24, n), device=DEV)
    ext.pointnet2.three_interpolate_wrapper(2, 24, m, n, T(feats), idx, T(w), out)
    assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(feats, widx, w))  # same rounding sequence


@pytest.mark.parametrize("n,m", [(16384, 4096), (4096, 1024)])
def test_three_nn_on_lidar_shaped_clouds(ext, oracle, n, m):
    """The FP levels' three_nn on LiDAR-shaped clouds (known = the cloud's FPS picks, as in the backbone): the known points
    crowd into an eighth of their bounding rectangle, the queries are served in cell order (csrc/three_nn_grid.hip, round 3)
    and written back at their own position.  One cloud with exact duplicates (distance ties -> lowest index).  Bit-exact."""
    import importlib
    S = importlib.import_module("3d_adapt_auto_driving_amd.synth")
    unknown = np.stack([S.lidar_scene(21, 16384)[:n], S.lidar_scene(22, 16384)[:n]], 0)
    unknown[1, n // 2:] = unknown[1, :n // 2]
    known = centres(oracle, unknown, m)
    d2 = torch.full((2, n, 3), float("nan"), device=DEV)
    idx = torch.full((2, n, 3), -1, dtype=torch.int32, device=DEV)
    ext.pointnet2.three_nn_wrapper(2, n, m, T(unknown), T(known), d2, idx)
    wd2, widx = oracle.three_nn(unknown, known)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.array_equal(d2.cpu().numpy(), wd2)


def test_three_nn_grid_edge_cases(ext, oracle):
    """Grid path of three_nn (m >= 1024): queries far outside the known points' extent, all known points
    identical (degenerate extent), heavy duplication (distance ties -> lowest indices), a tight cluster plus
    far outliers (many rings), unrelated clouds.  dist2 and idx bit-exact vs the brute-force oracle."""
    rng = np.random.default_rng(77)
    n, m = 2048, 1024
    cases = []
    kn = rng.uniform([-40, -1, 0], [40, 3, 70], (m, 3)).astype(np.float32)
    un = rng.uniform([-200, -5, -150], [200, 5, 300], (n, 3)).astype(np.float32)        # mostly outside
    cases.append((un, kn))
    cases.append((rng.standard_normal((n, 3)).astype(np.float32), np.tile(np.float32([[1.5, 0.5, 2.5]]), (m, 1))))
    base = rng.uniform(-10, 10, (16, 3)).astype(np.float32)
    cases.append((rng.uniform(-12, 12, (n, 3)).astype(np.float32), base[rng.integers(0, 16, m)]))   # 16 distinct sites
    kn = (rng.standard_normal((m, 3)) * 0.05).astype(np.float32)
    kn[:5] = [[500, 0, 500], [-500, 0, 500], [500, 0, -500], [-500, 0, -500], [0, 0, 800]]
    un = np.concatenate([(rng.standard_normal((n // 2, 3)) * 0.05), rng.uniform(-600, 600, (n // 2, 3))]).astype(np.float32)
    cases.append((un, kn))
    un = scenes(1, n, seed0=5)[0]
    cases.append((un, un[rng.permutation(n)[:m]].copy()))                                 # known = subset of unknown
    for un, kn in cases:
        unknown, known = np.stack([un, un[::-1].copy()]), np.stack([kn, kn[::-1].copy()])
        d2 = torch.empty((2, n, 3), device=DEV)
        idx = torch.empty((2, n, 3), dtype=torch.int32, device=DEV)
        ext.pointnet2.three_nn_wrapper(2, n, m, T(unknown), T(known), d2, idx)
        wd2, widx = oracle.three_nn(unknown, known)
        assert np.array_equal(idx.cpu().numpy(), widx)
        assert np.array_equal(d2.cpu().numpy(), wd2)


def test_roipool3d(ext, oracle):
    rng = np.random.default_rng(8)
    xyz = scenes(2, 16384, seed0=31)
    boxes = np.stack([boxes3d(rng, 100), boxes3d(rng, 100)], 0)
    boxes[0, 3, :3] = [300, 300, 300]            # empty box
    boxes[1, :10, 0:3] = xyz[1, :10] + [0, 1.0, 0]  # boxes that certainly contain points
    boxes[:, :, 3:6] += 2.0; boxes[:, :, 1] += 1.0  # enlarged as roipool3d_utils.py:19 does
    feat = rng.standard_normal((2, 16384, 130)).astype(np.float32)
    pooled = torch.zeros((2, 100, 512, 133), device=DEV)
    empty = torch.zeros((2, 100), dtype=torch.int32, device=DEV)
    ext.roipool3d.forward(T(xyz), T(boxes), T(feat), pooled, empty)
    wp, we = oracle.roipool3d(xyz, boxes, feat, 512)
    assert np.array_equal(empty.cpu().numpy(), we)
    assert np.array_equal(pooled.cpu().numpy(), wp)
    assert we[0, 3] == 1 and we.sum() < 150


def test_roipool3d_full_and_partial_boxes(ext, oracle):
    rng = np.random.default_rng(9)
    xyz = rng.uniform([-3, 0, 8], [3, 2, 14], (1, 5000, 3)).astype(np.float32)
    boxes = np.array([[[0, 2, 11, 2, 5.5, 5.5, 0.3],      # > 512 points inside
                       [2.5, 2, 13.5, 2, 0.6, 0.6, 1.0],   # a few points: wrap-around fill
                       [50, 2, 50, 2, 1, 1, 0]]], np.float32)
    feat = rng.standard_normal((1, 5000, 7)).astype(np.float32)
    pooled = torch.zeros((1, 3, 512, 10), device=DEV)
    empty = torch.zeros((1, 3), dtype=torch.int32, device=DEV)
    ext.roipool3d.forward_slow(T(xyz), T(boxes), T(feat), pooled, empty)
    wp, we = oracle.roipool3d(xyz, boxes, feat, 512)
    assert np.array_equal(empty.cpu().numpy(), we) and list(we[0]) == [0, 0, 1]
    assert np.array_equal(pooled.cpu().numpy(), wp)


@pytest.mark.parametrize("n,thresh", [(6300, 0.8), (2700, 0.8), (900, 0.5), (65, 0.3), (1, 0.5)])
def test_nms_normal(ext, oracle, n, thresh):
    boxes = bev_boxes(np.random.default_rng(n), n, spread=25.0, rotated=False)
    keep = torch.zeros(n, dtype=torch.int64)
    k = ext.iou3d.nms_normal_gpu(T(boxes), keep, thresh)
    want = oracle.nms_normal(boxes, thresh)
    assert k == len(want) and np.array_equal(keep[:k].numpy(), want)


@pytest.mark.parametrize("n,thresh", [(100, 0.1), (300, 0.3), (64, 0.01), (700, 0.5), (129, 0.2), (2000, 0.3), (4097, 0.05)])
def test_nms_rotated(ext, oracle, n, thresh):
    boxes = bev_boxes(np.random.default_rng(n + 1), n, spread=12.0)
    keep = torch.zeros(n, dtype=torch.int64)
    k = ext.iou3d.nms_gpu(T(boxes), keep, thresh)
    want = oracle.nms(boxes, thresh)
    assert k == len(want) and np.array_equal(keep[:k].numpy(), want)


@pytest.mark.parametrize("rotated", [False, True])
@pytest.mark.parametrize("n,thresh,spread", [(8449, 0.5, 40.0), (9000, 0.8, 25.0), (9000, 0.3, 60.0), (8512 + 64 * 64 + 1, 0.4, 70.0), (20000, 0.5, 80.0)])
def test_nms_beyond_the_near_window(ext, oracle, n, thresh, spread, rotated):
    """More than 8448 boxes (W > 132 mask words per row; iou3d.hip's header names 9000, the reference's RPN_PRE_NMS_TOP_N): a kept
    row's words further than 131 blocks behind it are ORed in by the resolve kernel's far pass (they were dropped)."""
    boxes = bev_boxes(np.random.default_rng(n + int(rotated)), n, spread=spread, rotated=rotated)
    keep = torch.zeros(n, dtype=torch.int64)
    k = (ext.iou3d.nms_gpu if rotated else ext.iou3d.nms_normal_gpu)(T(boxes), keep, thresh)
    want = (oracle.nms if rotated else oracle.nms_normal)(boxes, thresh)
    assert 0.05 * n < len(want) <= n - 50                           # both outcomes represented
    assert k == len(want) and np.array_equal(keep[:k].numpy(), want)


@pytest.mark.parametrize("rotated", [False, True])
@pytest.mark.parametrize("gap", [8448 - 300, 8448, 8448 + 64 * 70, 30000])
def test_nms_suppressed_only_by_a_far_row(ext, oracle, rotated, gap):
    """300 disjoint boxes, ``gap`` disjoint fillers elsewhere, then copies of the first 300: each copy overlaps exactly one earlier box,
    300 + gap rows in front of it -- it falls through the far words alone (or, for the smallest gap, through the near window)."""
    rng = np.random.default_rng(gap)
    gx, gz = np.meshgrid(np.arange(20) * 8.0, np.arange(15) * 8.0)
    head = np.stack([gx.ravel() - 2, gz.ravel() - 1, gx.ravel() + 2, gz.ravel() + 1,
                     rng.uniform(-3, 3, 300) if rotated else np.zeros(300)], 1)
    side = int(np.ceil(np.sqrt(gap)))
    fx, fz = np.meshgrid(1000.0 + np.arange(side) * 8.0, np.arange(side) * 8.0)
    fill = np.stack([fx.ravel() - 2, fz.ravel() - 1, fx.ravel() + 2, fz.ravel() + 1, np.zeros(side * side)], 1)[:gap]
    tail = head.copy(); tail[:, :4] += 0.05
    tail = tail[rng.permutation(300)][:257]
    boxes = np.concatenate([head, fill, tail], 0).astype(np.float32)
    n = boxes.shape[0]
    keep = torch.zeros(n, dtype=torch.int64)
    k = (ext.iou3d.nms_gpu if rotated else ext.iou3d.nms_normal_gpu)(T(boxes), keep, 0.5)
    want = (oracle.nms if rotated else oracle.nms_normal)(boxes, 0.5)
    assert len(want) == 300 + gap
    assert k == len(want) and np.array_equal(keep[:k].numpy(), want)


@pytest.mark.parametrize("K,thresh,spread", [(70, 0.8, 15.0), (70, 0.3, 6.0), (256, 0.2, 4.0), (300, 0.3, 6.0)])
def test_nms_device_batched_prefix(ext, oracle, K, thresh, spread):
    """The quota kernel (max_keep <= 256: a block's columns are tested against the KEPT boxes when the block is staged) and
    the general kernel (kept rows knock out all later columns) == the first K entries of the oracle's greedy list; dense
    boxes and low thresholds so that most drops come from boxes kept many blocks earlier."""
    rng = np.random.default_rng(12)
    P, nmax = 6, 3000
    counts = np.array([3000, 2999, 64, 0, 1500, 1], np.int32)
    boxes = np.stack([bev_boxes(rng, nmax, spread=spread, rotated=False) for _ in range(P)], 0)
    keep = torch.empty((P, K), dtype=torch.int32, device=DEV)
    num = torch.empty((P,), dtype=torch.int32, device=DEV)
    ext.iou3d.nms_device(T(boxes), T(counts), thresh, False, K, keep, num)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for p in range(P):
        want = oracle.nms_normal(boxes[p, :counts[p]], thresh)[:K]
        assert num[p] == len(want)
        assert np.array_equal(keep[p, :num[p]], want)
        assert (keep[p, num[p]:] == -1).all()


@pytest.mark.parametrize("rotated", [True, False])
@pytest.mark.parametrize("nmax,K,thresh,spread", [(100, 100, 0.1, 6.0), (128, 128, 0.1, 3.0), (128, 40, 0.3, 2.0), (65, 65, 0.01, 4.0), (7, 7, 0.1, 1.0)])
def test_nms_device_small_problems_dense_form(ext, oracle, nmax, K, thresh, spread, rotated):
    """n <= 128 (the final rotated NMS of a scene: <= 100 boxes, threshold 0.1): the all-pairs mask + one-wave resolve of
    csrc/iou3d.hip (round 3) against the oracle's greedy loop, ragged counts (incl. 0 and 1), a keep quota below the answer,
    boxes packed so tightly that most are suppressed by a box kept long before them."""
    rng = np.random.default_rng(nmax + K)
    P = 9
    counts = np.array([nmax, nmax - 1, 64, 65, 0, 1, 2, nmax // 2, nmax], np.int32)
    boxes = np.stack([bev_boxes(rng, nmax, spread=spread, rotated=rotated) for _ in range(P)], 0)
    boxes[8, 1::2] = boxes[8, 0::2][:boxes[8, 1::2].shape[0]]                  # exact duplicates: every second box identical to its predecessor
    keep = torch.full((P, K), -9, dtype=torch.int32, device=DEV)
    num = torch.full((P,), -9, dtype=torch.int32, device=DEV)
    ext.iou3d.nms_device(T(boxes), T(counts), thresh, rotated, K, keep, num)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for p in range(P):
        want = (oracle.nms if rotated else oracle.nms_normal)(boxes[p, :counts[p]], thresh)[:K]
        assert num[p] == len(want), (p, num[p], len(want))
        assert np.array_equal(keep[p, :num[p]], want)
        assert (keep[p, num[p]:] == -1).all()


def test_overlap_and_iou_bev(ext, oracle):
    rng = np.random.default_rng(13)
    a, b = bev_boxes(rng, 256, spread=8.0), bev_boxes(rng, 200, spread=8.0)
    b[:5] = a[:5]  # identical boxes (degenerate polygon case)
    ov = torch.zeros((256, 200), device=DEV); iou = torch.zeros((256, 200), device=DEV)
    ext.iou3d.boxes_overlap_bev_gpu(T(a), T(b), ov)
    ext.iou3d.boxes_iou_bev_gpu(T(a), T(b), iou)
    # same rounding sequence as the oracle; libm differences (f64 sin/cos/atan2 rounded to f32) may
    # move a value by an ulp-scale amount -> 1e-5 absolute on areas of O(10)
    np.testing.assert_allclose(ov.cpu().numpy(), oracle.boxes_overlap_bev(a, b), rtol=0, atol=1e-5)
    np.testing.assert_allclose(iou.cpu().numpy(), oracle.boxes_iou_bev(a, b), rtol=0, atol=1e-6)
    assert (oracle.boxes_overlap_bev(a, b) > 0).mean() > 0.02


def test_rotate_iou_eval(oracle):
    import importlib
    lib = importlib.import_module("3d_adapt_auto_driving_amd._lib")
    rng = np.random.default_rng(14)
    def cbox(n):
        return np.stack([rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), rng.uniform(1.4, 2, n),
                         rng.uniform(3, 5, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
    a, q = cbox(150), cbox(90)
    for crit in (-1, 0, 1, 2):
        out = torch.zeros((150, 90), device=DEV)
        ta, tq = T(a), T(q)
        lib.call("prcnn_rotate_iou_eval", 150, 90, ta.data_ptr(), tq.data_ptr(), out.data_ptr(), crit,
                 lib.current_stream(out))
        np.testing.assert_allclose(out.cpu().numpy(), oracle.rotate_iou_eval(a, q, crit), rtol=0, atol=1e-5)


def test_rotate_iou_kernel_vs_reference_python_fixture():
    """rotate_iou.hip against g9 = output of the reference's OWN evaluate/rotate_iou.py (run in the build container by
    tests/golden/numba_shim.py), not against the builder's oracle.  The kernel follows the same numba typing
    (f64 area sum and final ratio), so the only freedom is OCML's f64 cos/sin vs glibc's inside the (float) f64
    trig contract: tolerance 1e-6 absolute on ratios <= 1 and 1e-5 on areas (criterion 2, values up to ~20 m^2),
    and at least 99 % of the pairs bit for bit."""
    import importlib
    lib = importlib.import_module("3d_adapt_auto_driving_amd._lib")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_rotate_iou_ref.npz"))
    a, q, und = g["boxes"], g["query_boxes"], g["undefined"]
    ta, tq = T(a), T(q)
    for crit in (-1, 0, 1, 2):
        out = torch.full((256, 256), float("nan"), device=DEV)
        lib.call("prcnn_rotate_iou_eval", 256, 256, ta.data_ptr(), tq.data_ptr(), out.data_ptr(), crit,
                 lib.current_stream(out))
        got, want = out.cpu().numpy(), g["iou_c%d" % crit]
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got[~und], want[~und], rtol=0, atol=1e-5 if crit == 2 else 1e-6)
        same = (got.view(np.uint32) == want.view(np.uint32))[~und].mean()
        assert same >= 0.99, (crit, same)


@pytest.mark.parametrize("ns1,ns2", [(64, 64), (16, 48)])
def test_rcnn_roi_geometry_equals_the_separate_entry_points(ext, ns1, ns2):
    """prcnn_rcnn_roi_geometry (one wave per RoI: FPS 512 -> 128, limited ball query, representative map, FPS 128 -> 32, ball query,
    representative map) == the six separate entry points and == the oracle chain, bit for bit.  RoI clouds as RoI pooling makes
    them: `count` distinct points (1, a handful, ~60, 127-129, 300, 511, 512), the rest wrap-around copies (k % count); duplicates
    inside the distinct part; a cloud whose points all coincide."""
    from oracle import ext_cpu
    rng = np.random.default_rng(ns1 * 100 + ns2)
    counts = [1, 2, 7, 33, 60, 64, 65, 100, 127, 128, 129, 200, 300, 511, 512, 512, 50, 90]
    b = len(counts)
    xyz = np.zeros((b, 512, 3), np.float32)
    for i, c in enumerate(counts):
        base = (rng.standard_normal((c, 3)) * [1.2, 0.5, 0.6]).astype(np.float32)
        if i == 5:
            base[10:20] = base[0:10]                        # duplicates among the "distinct" points
        if i == 16:
            base[:] = base[0]                               # all points coincide
        xyz[i] = base[np.arange(512) % c]
    limit = torch.tensor(counts, dtype=torch.int32, device=DEV)
    P = ext.pointnet2
    X = T(xyz)
    got = P.rcnn_roi_geometry_wrapper(X, limit, 128, 0.2, ns1, 32, 0.4, ns2)
    sel1, new1 = P.fps_new_xyz_wrapper(X, 128)
    idx1 = torch.full((b, 128, ns1), -1, dtype=torch.int32, device=DEV)
    P.ball_query_limit_wrapper(b, 512, 128, 0.2, ns1, new1, X, limit, idx1)
    rep1 = P.dup_rep_wrapper(sel1, 512, limit, None)
    sel2, new2 = P.fps_new_xyz_wrapper(new1, 32)
    idx2 = torch.zeros((b, 32, ns2), dtype=torch.int32, device=DEV)
    P.ball_query_wrapper(b, 128, 32, 0.4, ns2, new2, new1, idx2)
    rep2 = P.dup_rep_wrapper(sel2, 128, None, rep1)
    for k, (g, w) in enumerate(zip(got, (new1, idx1, rep1, new2, idx2, rep2))):
        assert torch.equal(g, w), k
    cpu = ext_cpu.pointnet2_cpu.rcnn_roi_geometry_wrapper(X.cpu(), limit.cpu(), 128, 0.2, ns1, 32, 0.4, ns2)
    for k, (g, w) in enumerate(zip(got, cpu)):
        assert torch.equal(g.cpu(), w), k
    assert int((got[2] != torch.arange(128, device=DEV).view(1, -1)).sum()) > 500        # many centres are copies of earlier ones


def test_rcnn_roi_geometry_ties_and_every_count(ext):
    """The same chain on clouds that make the tie rule decide: coordinates on a coarse lattice (many equal running minima between
    DIFFERENT points, so the smallest key among the copies of each point matters), every count of distinct points from 1 to 512 in steps
    that cross the 64 / 128 / 256 register-slot boundaries of the kernel, the distinct part itself holding repeats."""
    rng = np.random.default_rng(77)
    counts = list(range(1, 20)) + [31, 32, 33, 63, 64, 65, 66, 96, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 258, 384, 510, 511, 512]
    counts += [int(c) for c in rng.integers(1, 513, size=40)]
    b = len(counts)
    xyz = np.zeros((b, 512, 3), np.float32)
    for i, c in enumerate(counts):
        step = [0.25, 0.1, 0.05][i % 3]
        base = (rng.integers(-6, 7, size=(c, 3)) * step).astype(np.float32)
        if i % 4 == 0:
            base = (base + rng.standard_normal((c, 3)).astype(np.float32) * 0.3).astype(np.float32)
        xyz[i] = base[np.arange(512) % c]
    limit = torch.tensor(counts, dtype=torch.int32, device=DEV)
    P = ext.pointnet2
    X = T(xyz)
    got = P.rcnn_roi_geometry_wrapper(X, limit, 128, 0.2, 64, 32, 0.4, 64)
    sel1, new1 = P.fps_new_xyz_wrapper(X, 128)
    idx1 = torch.full((b, 128, 64), -1, dtype=torch.int32, device=DEV)
    P.ball_query_limit_wrapper(b, 512, 128, 0.2, 64, new1, X, limit, idx1)
    rep1 = P.dup_rep_wrapper(sel1, 512, limit, None)
    sel2, new2 = P.fps_new_xyz_wrapper(new1, 32)
    idx2 = torch.zeros((b, 32, 64), dtype=torch.int32, device=DEV)
    P.ball_query_wrapper(b, 128, 32, 0.4, 64, new2, new1, idx2)
    rep2 = P.dup_rep_wrapper(sel2, 128, None, rep1)
    for k, (g, w) in enumerate(zip(got, (new1, idx1, rep1, new2, idx2, rep2))):
        bad = (g != w).reshape(b, -1).any(1).nonzero().flatten().tolist()
        assert not bad, (k, [counts[i] for i in bad][:10])


def _pack_rows_by_cloud(pk, b):
    """a BallPack's tiles grouped by cloud (the order of the clouds' tiles in the list is the counter's): per cloud the rowinfo words and
    relative coordinates of its tiles, in tile order"""
    ntiles, nrows = int(pk.hdr[0]), int(pk.hdr[1])
    tc = pk.tilecloud[:ntiles].cpu().numpy()
    info = pk.rowinfo[:ntiles * 64].cpu().numpy().reshape(ntiles, 64)
    dxyz = pk.rowdxyz[:ntiles * 64].cpu().numpy().reshape(ntiles, 64, 4)
    out = []
    for c in range(b):
        t = np.nonzero(tc == c)[0]
        assert len(t) == 0 or (np.diff(t) == 1).all(), "a cloud's tiles are consecutive"
        out.append((info[t].copy(), dxyz[t].copy()))
    return ntiles, nrows, out


@pytest.mark.parametrize("ns1,ns2", [(64, 64), (16, 48)])
def test_rcnn_roi_geometry_packs_equal_ball_pack(ext, ns1, ns2):
    """prcnn_rcnn_roi_geometry_packs: the six geometry outputs of prcnn_rcnn_roi_geometry and, out of the same launch, the two row lists
    exactly as prcnn_ball_pack_ex writes them from idx1 (limit, crep = rep1) and idx2 (rep = rep1, crep = rep2): per cloud the same rows
    in the same order in the same tiles, relative coordinates bit for bit, the padding rows of a cloud's last tile included."""
    rng = np.random.default_rng(5 + ns1)
    counts = [1, 2, 7, 33, 60, 64, 65, 100, 127, 128, 129, 200, 300, 511, 512, 512, 50, 90] + [int(c) for c in rng.integers(1, 513, size=30)]
    b = len(counts)
    xyz = np.zeros((b, 512, 3), np.float32)
    for i, c in enumerate(counts):
        base = (rng.standard_normal((c, 3)) * [1.2, 0.5, 0.6]).astype(np.float32)
        if i % 5 == 0:
            base = (rng.integers(-6, 7, size=(c, 3)) * 0.1).astype(np.float32)      # lattice: ties, coinciding points
        if i == 16:
            base[:] = base[0]
        xyz[i] = base[np.arange(512) % c]
    limit = torch.tensor(counts, dtype=torch.int32, device=DEV)
    P = ext.pointnet2
    X = T(xyz)
    want = P.rcnn_roi_geometry_wrapper(X, limit, 128, 0.2, ns1, 32, 0.4, ns2)
    for zeroed, with_idx in ((False, True), (True, True), (True, False)):
        hdrs = (torch.zeros(4, dtype=torch.int32, device=DEV), torch.zeros(4, dtype=torch.int32, device=DEV)) if zeroed else (None, None)
        got = P.rcnn_roi_geometry_packs_wrapper(X, limit, 128, 0.2, ns1, 32, 0.4, ns2, *hdrs, with_idx)
        for k, (g, w) in enumerate(zip(got[:6], want)):
            if not with_idx and k in (1, 4):                 # not written: a shape without storage
                #  on the "meta" device: the shape is there, a consumer that reads VALUES gets an error instead of garbage
                assert tuple(g.shape) == tuple(w.shape) and g.device.type == "meta" and g.dtype == torch.int32
                with pytest.raises(RuntimeError):
                    P.ball_pack_wrapper(g, X, got[0])
                with pytest.raises((RuntimeError, NotImplementedError)):
                    g.cpu().numpy()
                continue
            assert torch.equal(g, w), k
        new1, idx1, rep1, new2, idx2, rep2 = want
        for lvl, (pk, ref) in enumerate(((got[6], P.ball_pack_wrapper(idx1, X, new1, limit, None, rep1)),
                                         (got[7], P.ball_pack_wrapper(idx2, new1, new2, None, rep1, rep2)))):
            a, r = _pack_rows_by_cloud(pk, b), _pack_rows_by_cloud(ref, b)
            assert a[0] == r[0] and a[1] == r[1], (lvl, a[:2], r[:2])
            for c in range(b):
                assert np.array_equal(a[2][c][0], r[2][c][0]), (lvl, counts[c])
                assert np.array_equal(a[2][c][1].view(np.uint32), r[2][c][1].view(np.uint32)), (lvl, counts[c])
    # ... and in the form whose rows carry their cloud: every cloud's rows in one block (no padding), the same rows in the same order
    got = P.rcnn_roi_geometry_packs_wrapper(X, limit, 128, 0.2, ns1, 32, 0.4, ns2, None, None, False, True)
    for lvl, (pk, ref) in enumerate(((got[6], P.ball_pack_wrapper(idx1, X, new1, limit, None, rep1)),
                                     (got[7], P.ball_pack_wrapper(idx2, new1, new2, None, rep1, rep2)))):
        assert pk.tilecloud is None and int(pk.hdr[1]) == int(ref.hdr[1])
        n = int(pk.hdr[1])
        info = pk.rowinfo[:n].cpu().numpy().astype(np.int64)
        dx = pk.rowdxyz[:n].cpu().numpy()
        cl = info >> 16
        r = _pack_rows_by_cloud(ref, b)[2]
        for c in range(b):
            sel = np.nonzero(cl == c)[0]
            assert len(sel) > 0 and (np.diff(sel) == 1).all(), (lvl, counts[c])
            want_info = r[c][0].reshape(-1)[:len(sel)].astype(np.int64)
            assert np.array_equal((info[sel] >> 9) & 0x7f, want_info >> 16) and np.array_equal(info[sel] & 0x1ff, want_info & 0xffff), (lvl, counts[c])
            assert np.array_equal(dx[sel].view(np.uint32), r[c][1].reshape(-1, 4)[:len(sel)].view(np.uint32)), (lvl, counts[c])
            # what the per-cloud list holds beyond these rows is the padding of the cloud's last tile
            assert len(sel) > 64 * (r[c][0].shape[0] - 1), (lvl, counts[c])


def test_point_major_kernels(ext, oracle):
    """group_cat_pm / maxpool_pm / three_interpolate_pm against the oracle's channel-major results
    rearranged to the point-major row layout [features | pad | dx dy dz | 0] (pure data movement and
    the same rounding sequence: bit-exact)."""
    rng = np.random.default_rng(33)
    for c in (0, 1, 6, 128):
        n, m, ns = 2048, 300, 16
        xyz = scenes(2, n, seed0=50 + c)
        new_xyz = centres(oracle, xyz, m)
        feats_cm = rng.standard_normal((2, c, n)).astype(np.float32) if c else None
        want, idx = oracle.query_and_group(0.9, ns, xyz, new_xyz, feats_cm)       # (2, 3+c, m, ns)
        c4 = (c + 3) // 4 * 4
        out = torch.full((2, m * ns, c4 + 4), 7.0, device=DEV)
        feats_pm = T(np.ascontiguousarray(feats_cm.transpose(0, 2, 1))) if c else None
        ext.pointnet2.group_cat_pm_wrapper(2, n, m, c, ns, T(new_xyz), T(xyz), feats_pm, T(idx), out)
        got = out.cpu().numpy().reshape(2, m, ns, c4 + 4)
        assert np.array_equal(got[..., :c], want[:, 3:].transpose(0, 2, 3, 1))
        assert np.array_equal(got[..., c4:c4 + 3], want[:, :3].transpose(0, 2, 3, 1))
        assert (got[..., c:c4] == 0).all() and (got[..., c4 + 3] == 0).all()
    x = rng.standard_normal((150 * 32, 64)).astype(np.float32)
    out = torch.zeros((150, 100), device=DEV)
    ext.pointnet2.maxpool_pm_wrapper(T(x), 32, out, 36)
    assert np.array_equal(out.cpu().numpy()[:, 36:], x.reshape(150, 32, 64).max(1)) and (out[:, :36] == 0).all()
    known = rng.standard_normal((2, 24, 200)).astype(np.float32)                   # channel-major for the oracle
    i3 = rng.integers(0, 200, (2, 777, 3)).astype(np.int32)
    w3 = rng.uniform(0, 1, (2, 777, 3)).astype(np.float32)
    buf = torch.zeros((2, 777, 40), device=DEV)
    ext.pointnet2.three_interpolate_pm_wrapper(T(np.ascontiguousarray(known.transpose(0, 2, 1))), T(i3), T(w3), buf, 8)
    assert np.array_equal(buf.cpu().numpy()[:, :, 8:32], oracle.three_interpolate(known, i3, w3).transpose(0, 2, 1))
    # the FP module's whole input in one launch: [interpolated | skip features]; ragged row count, several widths
    for c_skip in (4, 96, 256):
        skip = rng.standard_normal((2, 777, c_skip)).astype(np.float32)
        cat = torch.full((2, 777, 24 + c_skip), 7.0, device=DEV)
        ext.pointnet2.three_interpolate_cat_pm_wrapper(T(np.ascontiguousarray(known.transpose(0, 2, 1))), T(i3), T(w3), T(skip), cat)
        got = cat.cpu().numpy()
        assert np.array_equal(got[:, :, :24], oracle.three_interpolate(known, i3, w3).transpose(0, 2, 1))
        assert np.array_equal(got[:, :, 24:], skip)


def test_gather_affine_relu_pm(ext):
    """Layer-1 shortcut kernel vs its definition relu(P[idx] + wxyz . (xyz[idx] - centre)) evaluated by
    torch on the same device (same f32 operation order per element; fma contraction is off)."""
    from oracle import ext_cpu
    rng = np.random.default_rng(35)
    b, n, m, ns, cout = 3, 500, 40, 16, 64
    xyz = T(rng.standard_normal((b, n, 3)).astype(np.float32))
    new_xyz = xyz[:, :m].contiguous()
    P = T(rng.standard_normal((b, n, cout)).astype(np.float32))
    w = T(rng.standard_normal((3, cout)).astype(np.float32))
    idx = T(rng.integers(0, n, (b, m, ns)).astype(np.int32))
    out = torch.empty((b, m * ns, cout), device=DEV)
    ext.pointnet2.gather_affine_relu_pm_wrapper(new_xyz, xyz, P, w, idx, out)
    want = torch.empty((b, m * ns, cout))
    ext_cpu.pointnet2_cpu.gather_affine_relu_pm_wrapper(new_xyz.cpu(), xyz.cpu(), P.cpu(), w.cpu(), idx.cpu(), want)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=0, atol=2e-6)
    assert (out == 0).float().mean() > 0.2                      # the ReLU really clips


@pytest.mark.parametrize("c3", [128, 256])
def test_sa_mlp_fused_mfma_kernel(ext, c3):
    """Hand-written MFMA kernel (gather -> 3 layers -> max) vs the same computation with library f32
    GEMMs on the same device: identical gathers, GEMM summation order differs -> 2e-5 relative."""
    from oracle import ext_cpu
    rng = np.random.default_rng(36 + c3)
    b, n, m, ns = 5, 512, 37, 64
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = xyz[:, :m].contiguous()
    P = T(rng.standard_normal((b, n, 128)).astype(np.float32))
    wx = T((rng.standard_normal((3, 128)) * 0.5).astype(np.float32))
    idx = T(rng.integers(0, n, (b, m, ns)).astype(np.int32))
    w2 = T((rng.standard_normal((128, 128)) / 11).astype(np.float32)); b2 = T(rng.standard_normal(128).astype(np.float32) * 0.1)
    w3 = T((rng.standard_normal((128, c3)) / 11).astype(np.float32)); b3 = T(rng.standard_normal(c3).astype(np.float32) * 0.1)
    out = torch.full((b, m, c3 + 8), -1.0, device=DEV)
    ext.pointnet2.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, out, 8)
    # reference: same math through torch on the GPU (library GEMMs)
    y = torch.empty((b, m * ns, 128), device=DEV)
    ext.pointnet2.gather_affine_relu_pm_wrapper(new_xyz, xyz, P, wx, idx, y)
    y = torch.addmm(b2, y.view(-1, 128), w2).clamp_(min=0)
    y = torch.addmm(b3, y, w3).clamp_(min=0)
    want = y.view(b * m, ns, c3).amax(dim=1).view(b, m, c3)
    got = out[:, :, 8:]
    assert (out[:, :, :8] == -1.0).all()
    err = (got - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    assert (want > 0).float().mean() > 0.5
    # and against a float64 evaluation on the host (transpose-detecting: weights are not symmetric)
    yc = torch.empty((b, m * ns, 128))
    ext_cpu.pointnet2_cpu.gather_affine_relu_pm_wrapper(new_xyz.cpu(), xyz.cpu(), P.cpu(), wx.cpu(), idx.cpu(), yc)
    y64 = (yc.double().view(-1, 128) @ w2.cpu().double() + b2.cpu().double()).clamp_(min=0)
    y64 = (y64 @ w3.cpu().double() + b3.cpu().double()).clamp_(min=0).view(b * m, ns, c3).amax(dim=1)
    assert (got.cpu().double().view(-1, c3) - y64).abs().max().item() < 5e-5


@pytest.mark.parametrize("c3", [128, 256])
def test_sa_mlp_fused_serves_every_tile_at_scale(ext, c3):
    """Ticket scheduling at a size where workgroups retire and are replaced (20 000 tiles, 2 500 workgroups): every
    tile must be served exactly once -- the output starts as NaN and two launches are bit-identical."""
    rng = np.random.default_rng(c3)
    b, n, m, ns = 100, 256, 200, 64
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = xyz[:, :m].contiguous()
    P = T(rng.standard_normal((b, n, 128)).astype(np.float32))
    wx = T((rng.standard_normal((3, 128)) * 0.5).astype(np.float32))
    idx = T(rng.integers(0, n, (b, m, ns)).astype(np.int32))
    w2 = T((rng.standard_normal((128, 128)) / 11).astype(np.float32)); b2 = T(rng.standard_normal(128).astype(np.float32) * 0.1)
    w3 = T((rng.standard_normal((128, c3)) / 11).astype(np.float32)); b3 = T(rng.standard_normal(c3).astype(np.float32) * 0.1)
    outs = []
    for _ in range(2):
        out = torch.full((b, m, c3), float("nan"), device=DEV)
        ext.pointnet2.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, out, 0)
        outs.append(out)
    assert torch.isfinite(outs[0]).all(), "%d values never written" % int((~torch.isfinite(outs[0])).sum())
    assert torch.equal(outs[0], outs[1])
    y = torch.empty((b, m * ns, 128), device=DEV)
    ext.pointnet2.gather_affine_relu_pm_wrapper(new_xyz, xyz, P, wx, idx, y)
    y = torch.addmm(b2, y.view(-1, 128), w2).clamp_(min=0)
    y = torch.addmm(b3, y, w3).clamp_(min=0)
    want = y.view(b * m, ns, c3).amax(dim=1).view(b, m, c3)
    assert (outs[0] - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_mlp_epilogue_kernels(ext):
    rng = np.random.default_rng(34)
    x = rng.standard_normal((3, 20, 50, 16)).astype(np.float32)
    b = rng.standard_normal(20).astype(np.float32)
    t = T(x)
    ext.pointnet2.bias_relu_inplace_wrapper(t, T(b))
    assert np.array_equal(t.cpu().numpy(), np.maximum(x + b[None, :, None, None], 0))
    for ns in (16, 32, 64, 20):
        y = rng.standard_normal((3, 20, 50, ns)).astype(np.float32)
        out = torch.empty((3, 20, 50), device=DEV)
        ext.pointnet2.maxpool_bias_relu_wrapper(T(y), T(b), out)
        assert np.array_equal(out.cpu().numpy(), np.maximum(y + b[None, :, None, None], 0).max(-1))


def test_edge_cases_empty_ragged_and_degenerate(ext, oracle):
    """Empty and degenerate sizes through the C ABI: zero centres / zero boxes are no-ops, fewer than 3
    known points leave +inf / index 0 in three_nn (interpolate_gpu.cu:31-32), more samples than
    points repeats FPS picks, a radius covering everything returns indices 0..ns-1, a single-point
    cloud works, and RoI pooling over an empty cloud flags every box."""
    E = ext.pointnet2
    xyz = scenes(1, 777, seed0=3)
    t = T(xyz)
    # m = 0 / b = 0: nothing is touched
    idx = torch.full((1, 0, 8), 5, dtype=torch.int32, device=DEV)
    E.ball_query_wrapper(1, 777, 0, 0.5, 8, torch.empty((1, 0, 3), device=DEV), t, idx)
    E.furthest_point_sampling_wrapper(0, 777, 4, torch.empty((0, 777, 3), device=DEV),
                                      torch.empty((0, 777), device=DEV), torch.empty((0, 4), dtype=torch.int32, device=DEV))
    # FPS: m == 1, m > n (repeats once everything is taken), n == 1
    for n, m in ((777, 1), (5, 9), (1, 3)):
        pts = xyz[:, :n].copy()
        got, _ = fps_gpu(ext, pts, m)
        assert np.array_equal(got, oracle.furthest_point_sample(pts, m))
    # three_nn with 1 and 2 known points
    for m in (1, 2):
        d2 = torch.empty((1, 777, 3), device=DEV); i3 = torch.empty((1, 777, 3), dtype=torch.int32, device=DEV)
        E.three_nn_wrapper(1, 777, m, t, T(xyz[:, :m].copy()), d2, i3)
        wd, wi = oracle.three_nn(xyz, xyz[:, :m].copy())
        assert np.array_equal(i3.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
        assert np.isinf(wd[0, :, m:]).all()
    # huge radius: the first nsample indices; nsample larger than the cloud: back-fill with index 0
    for n, ns in ((777, 16), (5000, 16), (9, 32)):
        cloud = scenes(1, max(n, 16), seed0=n)[:, :n].copy()
        new_xyz = cloud[:, :7].copy()
        idx = torch.zeros((1, 7, ns), dtype=torch.int32, device=DEV)
        E.ball_query_wrapper(1, n, 7, 1e4, ns, T(new_xyz), T(cloud), idx)
        want = np.tile(np.where(np.arange(ns) < n, np.arange(ns), 0).astype(np.int32), (1, 7, 1))
        assert np.array_equal(idx.cpu().numpy(), want) and np.array_equal(want, oracle.ball_query(1e4, ns, cloud, new_xyz))
    # nsample = 1 grouping, single channel
    f = np.random.default_rng(0).standard_normal((1, 1, 777)).astype(np.float32)
    gi = np.random.default_rng(1).integers(0, 777, (1, 33, 1)).astype(np.int32)
    out = torch.empty((1, 1, 33, 1), device=DEV)
    E.group_points_wrapper(1, 1, 777, 33, 1, T(f), T(gi), out)
    assert np.array_equal(out.cpu().numpy(), oracle.group_points(f, gi))
    # RoI pooling: zero boxes is a no-op; an empty cloud flags all boxes and leaves rows untouched
    boxes = np.array([[[0, 2, 10, 2, 3, 5, 0.1]] * 3], np.float32)
    pooled = torch.full((1, 3, 16, 5), 9.0, device=DEV); flag = torch.zeros((1, 3), dtype=torch.int32, device=DEV)
    ext.roipool3d.forward(torch.empty((1, 0, 3), device=DEV), T(boxes), torch.empty((1, 0, 2), device=DEV), pooled, flag)
    assert flag.cpu().tolist() == [[1, 1, 1]] and (pooled == 9.0).all()
    ext.roipool3d.forward(t, torch.empty((1, 0, 7), device=DEV), torch.empty((1, 777, 2), device=DEV),
                          torch.empty((1, 0, 16, 5), device=DEV), torch.empty((1, 0), dtype=torch.int32, device=DEV))
    # NMS on zero boxes
    assert ext.iou3d.nms_gpu(torch.empty((0, 5), device=DEV), torch.zeros(0, dtype=torch.int64), 0.1) == 0
    assert ext.iou3d.nms_normal_gpu(torch.empty((0, 5), device=DEV), torch.zeros(0, dtype=torch.int64), 0.1) == 0
    # all boxes identical: only the first survives, for both IoU kinds
    same = np.tile(np.array([[0, 0, 4, 2, 0.3]], np.float32), (130, 1))
    keep = torch.zeros(130, dtype=torch.int64)
    assert ext.iou3d.nms_gpu(T(same), keep, 0.5) == 1 and keep[0] == 0
    assert ext.iou3d.nms_normal_gpu(T(same), keep, 0.5) == 1 and keep[0] == 0
    torch.cuda.synchronize()


def test_maximum_sizes_properties(ext, oracle):
    """BASELINE-size and larger inputs checked through size-independent properties (the oracle would
    take minutes): sortedness / radius / back-fill of ball query rows, FPS picks are distinct and the
    min-distance sequence is non-increasing, grouping equals a torch gather, dense-cloud sizes
    (N = 131072: generic FPS kernel, grid ball query with capped buckets) run."""
    n, m, ns, r = 131072, 2048, 32, 0.3
    xyz = scenes(1, n, seed0=9)
    t = T(xyz)
    sel, temp = fps_gpu(ext, xyz, m)
    assert len(np.unique(sel[0])) == m and sel[0, 0] == 0
    picked = xyz[0, sel[0].astype(np.int64)]
    dmin = [np.min(np.sum((picked[:j] - picked[j]) ** 2, 1)) for j in range(1, 300)]
    assert all(dmin[j] >= dmin[j + 1] - 1e-3 for j in range(len(dmin) - 1))         # FPS: radii shrink
    new_xyz = np.ascontiguousarray(picked[None])
    idx = torch.zeros((1, m, ns), dtype=torch.int32, device=DEV)
    ext.pointnet2.ball_query_wrapper(1, n, m, r, ns, T(new_xyz), t, idx)
    ix = idx.cpu().numpy()[0].astype(np.int64)
    d2 = np.sum((xyz[0][ix] - new_xyz[0][:, None, :]) ** 2, -1)
    assert (d2 < r * r + 1e-5).all()                                                  # every listed point is inside
    first_repeat = (ix == ix[:, :1])
    for row, rep in zip(ix[:64], first_repeat[:64]):
        cnt = ns if not rep[1:].any() else 1 + int(np.argmax(rep[1:]))
        assert (np.diff(row[:cnt]) > 0).all() and (row[cnt:] == row[0]).all()         # index order + back-fill
    # exact check of a few rows against numpy brute force
    for p in (0, 1, 777, m - 1):
        dd = xyz[0] - new_xyz[0, p]
        hits = np.nonzero((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2] < np.float32(r) * np.float32(r))[0][:ns]
        want = np.full(ns, hits[0]); want[:len(hits)] = hits
        assert np.array_equal(ix[p], want)
    feats = torch.randn((1, 8, n), device=DEV)
    out = torch.empty((1, 8, m, ns), device=DEV)
    ext.pointnet2.group_points_wrapper(1, 8, n, m, ns, feats, idx, out)
    assert torch.equal(out, torch.gather(feats, 2, idx.view(1, 1, -1).long().expand(-1, 8, -1)).view(1, 8, m, ns))


def test_rotate_iou_python_api_and_d3_overlap(oracle):
    """Reference-named evaluator entry points: rotate_iou_gpu_eval / bev_box_overlap / d3_box_overlap
    (evaluate/rotate_iou.py:294, eval2.py:131,165), the 3D one against the literal double loop."""
    import importlib
    R = importlib.import_module("3d_adapt_auto_driving_amd.rotate_iou")
    rng = np.random.default_rng(15)
    def box7(n):   # [x, y, z, l, h, w, ry] camera frame
        return np.stack([rng.uniform(-6, 6, n), rng.uniform(1, 2, n), rng.uniform(4, 16, n), rng.uniform(3, 5, n),
                         rng.uniform(1.3, 1.8, n), rng.uniform(1.4, 2, n), rng.uniform(-np.pi, np.pi, n)], 1)
    a, q = box7(70), box7(40)                                              # float64, as the evaluator passes them
    bev = R.bev_box_overlap(a[:, [0, 2, 3, 5, 6]], q[:, [0, 2, 3, 5, 6]])
    assert bev.dtype == np.float64
    np.testing.assert_allclose(bev, oracle.rotate_iou_eval(a[:, [0, 2, 3, 5, 6]], q[:, [0, 2, 3, 5, 6]], -1), atol=1e-5)
    for crit in (-1, 0, 1):
        got = R.d3_box_overlap(a, q, crit)
        rinc = oracle.rotate_iou_eval(a[:, [0, 2, 3, 5, 6]], q[:, [0, 2, 3, 5, 6]], 2).astype(np.float64)
        want = rinc.copy()
        for i in range(70):
            for j in range(40):
                if rinc[i, j] > 0:
                    iw = min(a[i, 1], q[j, 1]) - max(a[i, 1] - a[i, 4], q[j, 1] - q[j, 4])
                    if iw > 0:
                        a1, a2 = a[i, 3] * a[i, 4] * a[i, 5], q[j, 3] * q[j, 4] * q[j, 5]
                        inc = iw * rinc[i, j]
                        ua = {-1: a1 + a2 - inc, 0: a1, 1: a2}[crit]
                        want[i, j] = inc / ua
                    else:
                        want[i, j] = 0.0
        np.testing.assert_allclose(got, want, atol=1e-5)
        assert (want > 0).sum() > 20
    assert R.rotate_iou_gpu_eval(a[:0, :5], q[:, :5]).shape == (0, 40)


def test_bad_arguments_raise(ext):
    lib = __import__("importlib").import_module("3d_adapt_auto_driving_amd._lib")
    x = torch.zeros((1, 8, 3), device=DEV)
    with pytest.raises(RuntimeError):
        ext.pointnet2.ball_query_wrapper(1, 8, 8, 0.1, 4, x.cpu(), x, torch.zeros((1, 8, 4), dtype=torch.int32, device=DEV))
    with pytest.raises(lib.PrcnnError):
        lib.call("prcnn_ball_query", 1, 8, 8, 0.1, 4, None, None, None, None)


@pytest.mark.parametrize("c1,c2,c3,ns", [(16, 16, 32, 16), (32, 32, 64, 32)])
def test_sa_xyz_mlp_fused_level(ext, oracle, c1, c2, c3, ns):
    """Coordinates-only SA scale in one VALU kernel (csrc/sa_xyz_mlp.hip) vs the same function composed from
    torch ops in f32 (gather, centre subtraction, three linear+ReLU layers, max over nsample).  Tolerance 1e-5
    relative: only the summation order / FMA rounding differs.  Row count not a multiple of the block (tail
    lanes), output written into a column slice of a wider buffer."""
    rng = np.random.default_rng(c1)
    B, N, M = 3, 4096, 333
    xyz = scenes(B, N, seed0=61)
    new = centres(oracle, xyz, M)
    idx = oracle.ball_query(0.5 if ns == 32 else 0.1, ns, xyz, new)
    w1 = (rng.standard_normal((4, c1)) * 0.8).astype(np.float32); w1[3] = 0
    w2 = (rng.standard_normal((c1, c2)) * 0.3).astype(np.float32)
    w3 = (rng.standard_normal((c2, c3)) * 0.3).astype(np.float32)
    b1, b2, b3 = (rng.standard_normal(c).astype(np.float32) * 0.1 for c in (c1, c2, c3))
    out = torch.full((B, M, c3 + 8), -7.0, device=DEV)
    ext.pointnet2.sa_xyz_mlp_wrapper(T(new), T(xyz), T(idx), T(w1), T(b1), T(w2), T(b2), T(w3), T(b3), out, 5)
    txyz, tnew, tidx = T(xyz), T(new), T(idx).long()
    g = torch.gather(txyz.unsqueeze(1).expand(-1, M, -1, -1), 2, tidx.unsqueeze(-1).expand(-1, -1, -1, 3)) - tnew.unsqueeze(2)
    y = torch.relu(g @ T(w1)[:3] + T(b1))
    y = torch.relu(y @ T(w2) + T(b2))
    y = torch.relu(y @ T(w3) + T(b3)).max(dim=2).values
    got = out[:, :, 5:5 + c3]
    assert torch.allclose(got, y, rtol=1e-5, atol=1e-5), float((got - y).abs().max())
    assert bool((out[:, :, :5] == -7.0).all()) and bool((out[:, :, 5 + c3:] == -7.0).all())   # slice only
    assert ext.pointnet2.sa_xyz_mlp_supported(c1, c2, c3, ns) and not ext.pointnet2.sa_xyz_mlp_supported(c1, c2, c3, 64)


def test_rcnn_point_mlp_kernels(ext):
    """csrc/rcnn_point_mlp.hip (xyz_up x2 + concat + merge_down + SA1 per-point part as two MFMA kernels) vs the same
    chain with library f32 GEMMs; NaN-initialised outputs (every tile served), bit-identical reruns, 3000 tiles."""
    rng = np.random.default_rng(71)
    R, ld = 64 * 3000, 136
    rows = rng.standard_normal((R, ld)).astype(np.float32)
    rows[:, 5:8] = 0
    W = lambda *sh: T((rng.standard_normal(sh) / np.sqrt(sh[0])).astype(np.float32))
    wu1 = W(8, 128); wu1[5:] = 0
    wu2, wm, wp = W(128, 128), W(256, 128), W(128, 128)
    bu1, bu2, bm, bp = (T(rng.standard_normal(128).astype(np.float32) * 0.1) for _ in range(4))
    trow = T(rows)
    outs = []
    for _ in range(2):
        xfeat, merged, p = (torch.full((R, 128), float("nan"), device=DEV) for _ in range(3))
        ext.pointnet2.rcnn_point_mlp_wrapper(trow, 8, wu1, bu1, wu2, bu2, wm, bm, wp, bp, xfeat, merged, p)
        outs.append((xfeat, p, merged))
    assert all(torch.isfinite(o).all() for o in outs[0]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    x = torch.relu(torch.relu(trow[:, :8] @ wu1 + bu1) @ wu2 + bu2)
    want = torch.relu(torch.cat((x, trow[:, 8:]), dim=1) @ wm + bm) @ wp + bp
    assert (outs[0][0] - x).abs().max().item() <= 2e-5 * max(1.0, x.abs().max().item())
    assert (outs[0][1] - want).abs().max().item() <= 3e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(Exception):
        ext.pointnet2.rcnn_point_mlp_wrapper(trow[:100].contiguous(), 8, wu1, bu1, wu2, bu2, wm, bm, wp, bp, xfeat, merged, p)   # rows % 64
    # the one-kernel form (xfeat = merged = None: a tile never leaves LDS) gives the SAME BITS as the three launches, over all
    # tiles and over a live-tile list; tiles that are not listed stay untouched
    p1 = torch.full((R, 128), float("nan"), device=DEV)
    ext.pointnet2.rcnn_point_mlp_wrapper(trow, 8, wu1, bu1, wu2, bu2, wm, bm, wp, bp, None, None, p1)
    assert torch.equal(p1, outs[0][1])
    cnt = torch.from_numpy(rng.integers(0, 200, R // 512).astype(np.int32)).to(DEV)
    tiles = ext.pointnet2.pooled_tiles_wrapper(cnt, 512)
    p2 = torch.full((R, 128), float("nan"), device=DEV)
    ext.pointnet2.rcnn_point_mlp_wrapper(trow, 8, wu1, bu1, wu2, bu2, wm, bm, wp, bp, None, None, p2, tiles)
    live = torch.zeros(R // 64, dtype=torch.bool, device=DEV)
    live[tiles[0][:int(tiles[1][0])].long()] = True
    rows_live = live.repeat_interleave(64)
    assert torch.equal(p2[rows_live], outs[0][1][rows_live]) and torch.isnan(p2[~rows_live]).all() and 0 < int(live.sum()) < R // 64
    # ... and over the LIST of distinct rows (prcnn_pooled_rows / prcnn_rcnn_point_mlp_rows): exactly the first max(cnt, 1) rows of every
    # cloud are computed, the same bits, whichever tile of the list a row falls into; with a header zeroed by the caller as well
    for hdr in (None, torch.zeros(4, dtype=torch.int32, device=DEV)):
        rowlist = ext.pointnet2.pooled_rows_wrapper(cnt, 512, *(() if hdr is None else (hdr,)))
        n = int(rowlist[1][1])
        want_rows = torch.cat([torch.arange(max(int(c), 1)) + 512 * i for i, c in enumerate(cnt.cpu())])
        assert n == len(want_rows) and torch.equal(torch.sort(rowlist[0][:n].cpu().long())[0], want_rows)
        p3 = torch.full((R, 128), float("nan"), device=DEV)
        ext.pointnet2.rcnn_point_mlp_rows_wrapper(trow, 8, wu1, bu1, wu2, bu2, wm, bm, wp, bp, p3, rowlist)
        listed = torch.zeros(R, dtype=torch.bool, device=DEV)
        listed[want_rows.to(DEV)] = True
        assert torch.equal(p3[listed], outs[0][1][listed]) and torch.isnan(p3[~listed]).all()


@pytest.mark.parametrize("K,relu", [(128, True), (128, False), (256, True), (256, False)])
def test_rows_gemm128_layer(ext, K, relu):
    """The tiled MFMA layer kernel as a general 128-wide layer: contiguous and row-strided inputs vs torch f32."""
    rng = np.random.default_rng(K + int(relu))
    R = 64 * 333
    wt = T((rng.standard_normal((K, 128)) / np.sqrt(K)).astype(np.float32))
    b = T(rng.standard_normal(128).astype(np.float32) * 0.1)
    wide = T(rng.standard_normal((R, K + 8)).astype(np.float32))
    for a in (wide[:, :K].contiguous(), wide[:, 4:4 + K] if False else wide[:, 8:8 + K]):   # contiguous; strided (ld = K+8, col 8)
        out = torch.full((R, 128), float("nan"), device=DEV)
        ext.pointnet2.rows_gemm128_wrapper(a, wt, b, relu, out)
        want = a @ wt + b
        want = torch.relu(want) if relu else want
        assert torch.isfinite(out).all()
        assert (out - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_randomised_operator_sweep(ext, oracle):
    """~15 s of tests/fuzz_gpu_ops.py (random shapes across every dispatch threshold, clustered / duplicated / lattice
    clouds): FPS, ball query + grouping, three_nn + interpolation, NMS + IoU, RoI pooling bit-exact vs the oracle.
    The long form (python tests/fuzz_gpu_ops.py --seconds 240) ran 6608 cases clean on the round-1 kernels."""
    import time
    import fuzz_gpu_ops as Z
    rng = np.random.default_rng(20260928)
    cases = [Z.fuzz_fps, Z.fuzz_ball_group, Z.fuzz_three_nn, Z.fuzz_nms, Z.fuzz_roipool]
    t0, i = time.time(), 0
    while time.time() - t0 < 15.0 or i < 25:
        cases[i % len(cases)](rng)
        i += 1
    assert i >= 25
