"""-m gpu: the set-abstraction MLP over DISTINCT grouped rows (csrc/sa_packed.hip) and the bit-exact MLP oracle.

  * the MFMA kernels are a FIXED-order fma chain; oracle/mlp_oracle.c restates that order in scalar C, so the fused
    kernels (packed and unpacked, both output widths) are compared BIT FOR BIT with the CPU -- not within a tolerance;
  * the packed kernel skips the back-filled copies of a ball query's first hit (ball_query_gpu.cu:35-39) and must give
    the SAME BITS as the kernel that evaluates all nsample rows, for ball-query-shaped index rows and for arbitrary ones."""
import numpy as np
import pytest
import torch

from oracle import ext_cpu
from helpers import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def ball_like_idx(rng, b, m, n, ns, mean_cnt):
    """index rows shaped like a ball query's output: cnt distinct increasing indices, then copies of the first"""
    idx = np.empty((b, m, ns), np.int32)
    cnt = np.clip(rng.geometric(1.0 / mean_cnt, (b, m)), 1, ns)
    cnt[0, 0], cnt[-1, -1] = ns, 1
    if m > 2:
        cnt[0, 1] = 1
    for i in range(b):
        for c in range(m):
            k = np.sort(rng.choice(n, cnt[i, c], replace=False))
            idx[i, c, :cnt[i, c]] = k
            idx[i, c, cnt[i, c]:] = k[0]
    return idx, cnt


def mlp_params(rng, c3):
    w2 = T((rng.standard_normal((128, 128)) / 11).astype(np.float32)); b2 = T(rng.standard_normal(128).astype(np.float32) * 0.1)
    w3 = T((rng.standard_normal((128, c3)) / 11).astype(np.float32)); b3 = T(rng.standard_normal(c3).astype(np.float32) * 0.1)
    return w2, b2, w3, b3


def oracle_fused(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, width, col):
    out = torch.full((idx.shape[0], idx.shape[1], width), -1.0)
    ext_cpu.pointnet2_cpu.sa_mlp_fused_wrapper(new_xyz.cpu(), xyz.cpu(), P.cpu(), wx.cpu(), idx.cpu(), w2.cpu(), b2.cpu(),
                                               w3.cpu(), b3.cpu(), out, col)
    return out


@pytest.mark.parametrize("c3", [128, 256])
def test_fused_mfma_kernel_is_bit_exact_vs_the_fma_chain_oracle(ext, c3):
    """v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain (k = s, then s + 64, for s = 0..63): the whole fused kernel --
    builder, two MFMA layers, bias / ReLU epilogues, max -- reproduced bit for bit by oracle/mlp_oracle.c."""
    rng = np.random.default_rng(500 + c3)
    b, n, m, ns = 6, 512, 41, 64
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = xyz[:, :m].contiguous()
    P = T(rng.standard_normal((b, n, 128)).astype(np.float32))
    wx = T((rng.standard_normal((3, 128)) * 0.5).astype(np.float32))
    idx = T(rng.integers(0, n, (b, m, ns)).astype(np.int32))
    w2, b2, w3, b3 = mlp_params(rng, c3)
    out = torch.full((b, m, c3 + 8), -1.0, device=DEV)
    ext.pointnet2.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, out, 8)
    want = oracle_fused(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, c3 + 8, 8)
    assert torch.equal(out.cpu(), want), float((out.cpu() - want).abs().max())
    # the other order of the instruction's two k values is NOT what the hardware does (the check above is not vacuous)
    from oracle import oracle as O
    O.lib().orc_set_mfma_korder(1)
    try:
        other = oracle_fused(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, c3 + 8, 8)
    finally:
        O.lib().orc_set_mfma_korder(0)
    assert not torch.equal(other, want)


@pytest.mark.parametrize("c3", [128, 256])
@pytest.mark.parametrize("mean_cnt", [1.5, 9, 40])
def test_packed_kernel_bit_identical_to_unpacked_and_to_oracle(ext, c3, mean_cnt):
    rng = np.random.default_rng(int(c3 + 10 * mean_cnt))
    b, n, m, ns = 7, 512, 53, 64
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = T(rng.uniform(-2, 2, (b, m, 3)).astype(np.float32))
    P = T(rng.standard_normal((b, n, 128)).astype(np.float32))
    wx = T((rng.standard_normal((3, 128)) * 0.5).astype(np.float32))
    idx_np, cnt = ball_like_idx(rng, b, m, n, ns, mean_cnt)
    idx = T(idx_np)
    w2, b2, w3, b3 = mlp_params(rng, c3)
    full = torch.full((b, m, c3 + 4), -1.0, device=DEV)
    ext.pointnet2.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, full, 4)
    pk = ext.pointnet2.ball_pack_wrapper(idx)
    hdr = pk.hdr.cpu().numpy()
    assert hdr[1] == cnt.sum()                                               # distinct rows
    assert hdr[0] == sum((int(cnt[i].sum()) + 63) // 64 for i in range(b))   # tiles: per cloud, rounded up
    got = torch.full((b, m, c3 + 4), float("nan"), device=DEV)
    got[:, :, :4] = -1.0
    ext.pointnet2.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, got, 4)
    assert torch.equal(got, full), float((got - full).abs().max())
    want = oracle_fused(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, c3 + 4, 4)
    assert torch.equal(got.cpu(), want)
    # the row list: every centre's distinct points, in slot order, tiles of one cloud each
    tiles = int(hdr[0])
    info = pk.rowinfo.cpu().numpy().view(np.uint32)[:tiles * 64].reshape(tiles, 64)
    tc = pk.tilecloud.cpu().numpy()[:tiles]
    for i in range(b):
        rows = info[tc == i].reshape(-1)[:int(cnt[i].sum())]
        want_rows = np.concatenate([(np.uint32(c) << 16) | idx_np[i, c, :cnt[i, c]].astype(np.uint32) for c in range(m)])
        assert np.array_equal(rows, want_rows)


def test_packed_kernel_on_arbitrary_index_rows(ext):
    """cnt = 1 + (last slot that differs from slot 0): correct for ANY index tensor, not only ball-query output --
    random rows with repeats anywhere, rows of one repeated index, and strictly distinct rows."""
    rng = np.random.default_rng(77)
    b, n, m, ns = 3, 128, 64, 64
    xyz = T(rng.uniform(-1, 1, (b, n, 3)).astype(np.float32))
    new_xyz = xyz[:, :m].contiguous()
    P = T(rng.standard_normal((b, n, 128)).astype(np.float32))
    wx = T(rng.standard_normal((3, 128)).astype(np.float32))
    idx_np = rng.integers(0, 6, (b, m, ns)).astype(np.int32)               # heavy repetition, no structure
    idx_np[0, 0] = 5
    idx_np[1, 1] = np.arange(ns)
    idx_np[2, 2, 10:] = idx_np[2, 2, 0]
    idx = T(idx_np)
    w2, b2, w3, b3 = mlp_params(rng, 128)
    full = torch.empty((b, m, 128), device=DEV)
    ext.pointnet2.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, full, 0)
    got = torch.full((b, m, 128), float("nan"), device=DEV)
    ext.pointnet2.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, ext.pointnet2.ball_pack_wrapper(idx), w2, b2, w3, b3, got, 0)
    assert torch.equal(got, full)


@pytest.mark.parametrize("c3", [128, 256])
def test_packed_kernel_at_the_rcnn_batch8_shape(ext, oracle, c3):
    """BASELINE configs[2] shape of the two RCNN levels (800 clouds): real ball queries on KITTI-shaped RoI clouds, NaN-
    poisoned output, two launches bit-identical, packed == unpacked, and a sample of clouds == the CPU oracle."""
    rng = np.random.default_rng(c3)
    b, n, m, ns, r = (800, 512, 128, 64, 0.2) if c3 == 128 else (800, 128, 32, 64, 0.4)
    # RoI-like clouds: a few hundred distinct points in a car-sized box, wrapped around to n (roipool3d duplicates)
    base = rng.uniform([-2.5, -1.0, -1.5], [2.5, 1.0, 1.5], (b, n, 3)).astype(np.float32)
    uniq = rng.integers(20, n, b)
    for i in range(b):
        base[i] = base[i, np.arange(n) % uniq[i]]
    xyz = T(base)
    sel = torch.empty((b, m), dtype=torch.int32, device=DEV)
    ext.pointnet2.furthest_point_sampling_wrapper(b, n, m, xyz, torch.full((b, n), 1e10, device=DEV), sel)
    new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = torch.zeros((b, m, ns), dtype=torch.int32, device=DEV)
    ext.pointnet2.ball_query_wrapper(b, n, m, r, ns, new_xyz, xyz, idx)
    P = T(rng.standard_normal((b, n, 128)).astype(np.float32))
    wx = T((rng.standard_normal((3, 128)) * 0.5).astype(np.float32))
    w2, b2, w3, b3 = mlp_params(rng, c3)
    full = torch.empty((b, m, c3), device=DEV)
    ext.pointnet2.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, full, 0)
    outs = []
    for _ in range(2):
        got = torch.full((b, m, c3), float("nan"), device=DEV)
        ext.pointnet2.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, ext.pointnet2.ball_pack_wrapper(idx), w2, b2, w3, b3, got, 0)
        outs.append(got)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], full)
    pick = torch.tensor([0, 1, 399, 798, 799])
    want = oracle_fused(new_xyz[pick], xyz[pick], P[pick], wx, idx[pick], w2, b2, w3, b3, c3, 0)
    assert torch.equal(outs[0][pick].cpu(), want)
