"""-m gpu: the set-abstraction MLP over DISTINCT grouped rows (csrc/sa_packed.hip) and the bit-exact MLP oracle.

  * the MFMA kernels are a FIXED-order fma chain; oracle/mlp_oracle.c restates that order in scalar C, so the fused
    kernels (packed and unpacked, both output widths) are compared BIT FOR BIT with the CPU -- not within a tolerance;
  * the packed kernel skips the back-filled copies of a ball query's first hit (ball_query_gpu.cu:35-39) and must give
    the SAME BITS as the kernel that evaluates all nsample rows, for ball-query-shaped index rows and for arbitrary ones."""
import numpy as np
import pytest
import torch

from oracle import ext_cpu
from conftest import pkg
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


@pytest.mark.parametrize("mean_cnt", [1.5, 9, 40])
@pytest.mark.parametrize("col", [0, 5])
def test_narrow_scales_skip_the_padding_and_keep_its_bits(ext, mean_cnt, col):
    """RPN SA2's scales (64-64-128, 64-96-128; cfgs/default.yaml SA_CONFIG.MLPS[1]) arrive zero-padded to 128-128-128; told the real
    widths, the batched launch feeds only the real k to the MFMAs (csrc/sa_packed.hip sa_pk128_narrow) -- the same bits as the padded
    kernel one problem at a time, and as the oracle's padded chain over all nsample rows"""
    rng = np.random.default_rng(int(77 + 10 * mean_cnt + col))
    b, n, m = 5, 700, 61
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = T(rng.uniform(-2, 2, (b, m, 3)).astype(np.float32))
    width = col + 256
    probs, idxs = [], []
    for c2, ns, c0 in ((64, 16, col), (96, 32, col + 128)):
        def padded(a, shape):
            o = np.zeros(shape, np.float32)
            o[tuple(slice(0, d) for d in a.shape)] = a
            return T(o)
        P = padded(rng.standard_normal((b, n, 64)).astype(np.float32), (b, n, 128))
        wx = padded((rng.standard_normal((3, 64)) * 0.5).astype(np.float32), (3, 128))
        w2 = padded((rng.standard_normal((64, c2)) / 8).astype(np.float32), (128, 128))
        b2 = padded(rng.standard_normal(c2).astype(np.float32) * 0.1, (128,))
        w3 = padded((rng.standard_normal((c2, 128)) / 8).astype(np.float32), (128, 128))
        b3 = T(rng.standard_normal(128).astype(np.float32) * 0.1)
        idx = T(ball_like_idx(rng, b, m, n, ns, min(mean_cnt, ns / 2))[0])
        pk = ext.pointnet2.ball_pack_wrapper(idx, xyz, new_xyz)
        probs.append([new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, None, c0, True, (64, c2)])
        idxs.append(idx)
    outs = {}
    for name in ("narrow", "padded_batch", "single"):
        out = torch.zeros((b, m, width), device=DEV)
        for q in probs:
            q[9] = out
        if name == "single":
            for q in probs:
                ext.pointnet2.sa_packed_mlp_wrapper(*q[:12])
        else:
            ext.pointnet2.sa_packed_mlp_batch_wrapper([tuple(q) if name == "narrow" else tuple(q[:12]) for q in probs])
        outs[name] = out.cpu()
    assert torch.equal(outs["padded_batch"], outs["single"])
    assert torch.equal(outs["narrow"], outs["single"]), float((outs["narrow"] - outs["single"]).abs().max())
    # the engine's form: both scales' per-point parts side by side in ONE (b, n, 128) tensor, each problem reading its 64 columns
    pcat = torch.cat([q[2][:, :, :64] for q in probs], 2).contiguous()
    out = torch.zeros((b, m, width), device=DEV)
    ext.pointnet2.sa_packed_mlp_batch_wrapper([tuple(q[:2]) + (pcat[:, :, 64 * k:64 * k + 64],) + tuple(q[3:9]) + (out,) + tuple(q[10:]) for k, q in enumerate(probs)])
    assert torch.equal(out.cpu(), outs["single"])
    with pytest.raises(RuntimeError):          # a 64-wide P without the widths would be read as 128 columns
        ext.pointnet2.sa_packed_mlp_batch_wrapper([tuple(q[:2]) + (pcat[:, :, 64 * k:64 * k + 64],) + tuple(q[3:9]) + (out,) + tuple(q[10:12]) for k, q in enumerate(probs)])
    for q, idx in zip(probs, idxs):
        want = oracle_fused(new_xyz, xyz, q[2], q[3], idx, q[5], q[6], q[7], q[8], width, q[10])
        assert torch.equal(outs["narrow"][..., q[10]:q[10] + 128], want[..., q[10]:q[10] + 128])
    assert float(outs["narrow"].abs().max()) > 0


@pytest.mark.parametrize("c3", [128, 256])
@pytest.mark.parametrize("mean_cnt", [1.5, 9, 40])
@pytest.mark.parametrize("col", [4, 5])
def test_packed_kernel_bit_identical_to_unpacked_and_to_oracle(ext, c3, mean_cnt, col):
    """col = 4: output rows are 16-byte aligned -- tiles that hold many centres pool through LDS with 16-byte stores
    (csrc/segmax.hpp), the others in registers; col = 5: unaligned rows, every tile pools in registers.  Same bits."""
    rng = np.random.default_rng(int(c3 + 10 * mean_cnt))
    b, n, m, ns = 7, 512, 53, 64
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = T(rng.uniform(-2, 2, (b, m, 3)).astype(np.float32))
    P = T(rng.standard_normal((b, n, 128)).astype(np.float32))
    wx = T((rng.standard_normal((3, 128)) * 0.5).astype(np.float32))
    idx_np, cnt = ball_like_idx(rng, b, m, n, ns, mean_cnt)
    idx = T(idx_np)
    w2, b2, w3, b3 = mlp_params(rng, c3)
    full = torch.full((b, m, c3 + col), -1.0, device=DEV)
    ext.pointnet2.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, full, col)
    pk = ext.pointnet2.ball_pack_wrapper(idx, xyz, new_xyz)
    hdr = pk.hdr.cpu().numpy()
    assert hdr[1] == cnt.sum()                                               # distinct rows
    assert hdr[0] == sum((int(cnt[i].sum()) + 63) // 64 for i in range(b))   # tiles: per cloud, rounded up
    got = torch.full((b, m, c3 + col), float("nan"), device=DEV)
    got[:, :, :col] = -1.0
    ext.pointnet2.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, got, col)
    assert torch.equal(got, full), float((got - full).abs().max())
    want = oracle_fused(new_xyz, xyz, P, wx, idx, w2, b2, w3, b3, c3 + col, col)
    assert torch.equal(got.cpu(), want)
    # the row list: every centre's distinct points, in slot order, tiles of one cloud each
    tiles = int(hdr[0])
    info = pk.rowinfo.cpu().numpy().view(np.uint32)[:tiles * 64].reshape(tiles, 64)
    tc = pk.tilecloud.cpu().numpy()[:tiles]
    for i in range(b):
        rows = info[tc == i].reshape(-1)[:int(cnt[i].sum())]
        want_rows = np.concatenate([(np.uint32(c) << 16) | idx_np[i, c, :cnt[i, c]].astype(np.uint32) for c in range(m)])
        assert np.array_equal(rows, want_rows)
        d = pk.rowdxyz.cpu().numpy()[:tiles * 64].reshape(tiles, 64, 4)[tc == i].reshape(-1, 4)[:int(cnt[i].sum())]
        xc, cc = xyz[i].cpu().numpy(), new_xyz[i].cpu().numpy()
        assert np.array_equal(d[:, :3], xc[want_rows & 0xffff] - cc[want_rows >> 16])


def test_grouped_pack_equals_the_pack_of_every_batch(ext):
    """prcnn_ball_pack_groups: the row lists of several batches in one launch == prcnn_ball_pack on each batch's slice (tile
    counts, row counts, rows, relative coordinates; the order of a list's tiles may differ -- tiles are allocated by an atomic
    counter -- so tiles are compared cloud by cloud)."""
    rng = np.random.default_rng(123)
    lists, group, n, m, ns = 3, 4, 256, 37, 16
    b = lists * group
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = T(rng.uniform(-2, 2, (b, m, 3)).astype(np.float32))
    idx_np, cnt = ball_like_idx(rng, b, m, n, ns, 3.0)
    idx = T(idx_np)
    packs = ext.pointnet2.ball_pack_groups_wrapper(idx, xyz, new_xyz, group)
    assert len(packs) == lists
    for l, pk in enumerate(packs):
        lo, hi = l * group, (l + 1) * group
        one = ext.pointnet2.ball_pack_wrapper(idx[lo:hi].contiguous(), xyz[lo:hi].contiguous(), new_xyz[lo:hi].contiguous())
        assert pk.max_tiles == one.max_tiles
        ha, hb = pk.hdr.cpu().numpy(), one.hdr.cpu().numpy()
        assert ha[0] == hb[0] and ha[1] == hb[1] == cnt[lo:hi].sum()
        tiles = int(ha[0])
        def by_cloud(p):
            tc = p.tilecloud.cpu().numpy()[:tiles]
            info = p.rowinfo.cpu().numpy().view(np.uint32)[:tiles * 64].reshape(tiles, 64)
            d = p.rowdxyz.cpu().numpy()[:tiles * 64].reshape(tiles, 64, 4)
            return [(info[tc == c], d[tc == c]) for c in range(group)]
        for (ia, da), (ib, db) in zip(by_cloud(pk), by_cloud(one)):
            assert np.array_equal(ia, ib) and np.array_equal(da, db)


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
    ext.pointnet2.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, ext.pointnet2.ball_pack_wrapper(idx, xyz, new_xyz), w2, b2, w3, b3, got, 0)
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
        ext.pointnet2.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, ext.pointnet2.ball_pack_wrapper(idx, xyz, new_xyz), w2, b2, w3, b3, got, 0)
        outs.append(got)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], full)
    pick = torch.tensor([0, 1, 399, 798, 799])
    want = oracle_fused(new_xyz[pick], xyz[pick], P[pick], wx, idx[pick], w2, b2, w3, b3, c3, 0)
    assert torch.equal(outs[0][pick].cpu(), want)


@pytest.mark.parametrize("cin,c1,c2,c3", [(256, 128, 196, 256), (512, 256, 384, 512), (256, 256, 256, 512)])
def test_wide_packed_level_layer_by_layer_is_bit_exact(ext, cin, c1, c2, c3):
    """RPN SA3 / SA4 and the RCNN GroupAll level (config.py:58-61, 118-120): per-point layer, gather + affine, MFMA layer,
    MFMA layer + segmented max over the DISTINCT rows == the CPU oracle evaluating ALL nsample rows in the kernels'
    summation order, bit for bit; widths zero-padded to 128s exactly as net/fast_infer.py does."""
    rng = np.random.default_rng(cin + c2)
    b, n, m, ns = 3, 256, 40, 32
    E = ext.pointnet2
    c1p, c2p = (c1 + 127) // 128 * 128, (c2 + 127) // 128 * 128
    xyz = T(rng.uniform(-2, 2, (b, n, 3)).astype(np.float32))
    new_xyz = T(rng.uniform(-2, 2, (b, m, 3)).astype(np.float32))
    feats = T(rng.standard_normal((b, n, cin)).astype(np.float32))
    idx_np, cnt = ball_like_idx(rng, b, m, n, ns, 5)
    idx = T(idx_np)

    def padded(shape, real):
        w = np.zeros(shape, np.float32)
        w[tuple(slice(0, r) for r in real)] = rng.standard_normal(real) / np.sqrt(real[0])
        return T(w)
    wf, wx, b1 = padded((cin, c1p), (cin, c1)), padded((3, c1p), (3, c1)), padded((c1p,), (c1,))
    w2, b2 = padded((c1p, c2p), (c1, c2)), padded((c2p,), (c2,))
    w3, b3 = padded((c2p, c3), (c2, c3)), padded((c3,), (c3,))

    def level(X, dev_pack):
        dev = xyz.device if dev_pack else "cpu"
        mv = (lambda t: t) if dev_pack else (lambda t: t.cpu())
        P = torch.empty((b * n, c1p), device=dev)
        X.packed_layer_wrapper(mv(feats).view(b * n, cin), mv(wf), mv(b1), False, P)
        pk = X.ball_pack_wrapper(mv(idx), mv(xyz), mv(new_xyz))
        rows = pk.max_tiles * 64
        a1 = torch.zeros((rows, c1p), device=dev)
        X.packed_gather_affine_wrapper(mv(new_xyz), mv(xyz), P.view(b, n, c1p), mv(wx), pk, a1)
        y2 = torch.zeros((rows, c2p), device=dev)
        X.packed_layer_wrapper(a1, mv(w2), mv(b2), True, y2, pk)
        out = torch.full((b, m, c3 + 4), float("nan"), device=dev)
        out[:, :, :4] = -1
        X.packed_layer_segmax_wrapper(y2, mv(w3), mv(b3), pk, b, m, out, 4)
        return P, out
    Pg, og = level(E, True)
    Pc, oc = level(ext_cpu.pointnet2_cpu, False)
    assert torch.equal(Pg.cpu(), Pc)
    assert torch.isfinite(og).all()
    assert torch.equal(og.cpu(), oc), float((og.cpu() - oc).abs().max())
    # the three launches after the per-point part as ONE kernel (csrc/sa_wide.hip): the same bits, inside the same output slice
    assert E.sa_wide_fused_supported(c1p, c2p, c3)
    for zeroed in (False, True):
        of = torch.full((b, m, c3 + 4), float("nan"), device=DEV)
        of[:, :, :4] = -1
        if zeroed:
            of[:, :, 4:] = 0
        E.sa_wide_fused_wrapper(new_xyz, xyz, Pg.view(b, n, c1p), wx, E.ball_pack_wrapper(idx, xyz, new_xyz), w2, b2, w3, b3, of, 4, zeroed)
        assert torch.equal(of, og), (zeroed, float((of - og).abs().max()))
    # ... and with the per-point layer inside as well (csrc/sa_wide3.hip; the engine uses it where a level groups every point once):
    # the same bits again, for any row list
    if E.sa_wide_fused3_supported(cin, c1p, c2p, c3):
        wcat = torch.cat([wf.reshape(-1), w2.reshape(-1), w3.reshape(-1)]).contiguous()
        for zeroed in (False, True):
            of = torch.full((b, m, c3 + 4), float("nan"), device=DEV)
            of[:, :, :4] = -1
            if zeroed:
                of[:, :, 4:] = 0
            E.sa_wide_fused3_wrapper(new_xyz, xyz, feats, wcat, b1, wx, E.ball_pack_wrapper(idx, xyz, new_xyz), b2, b3, (cin, c1p, c2p, c3), of, 4, zeroed)
            assert torch.equal(of, og), ("fused3", zeroed, float((of - og).abs().max()))
        oc3 = torch.full((b, m, c3 + 4), float("nan"))
        oc3[:, :, :4] = -1
        ext_cpu.pointnet2_cpu.sa_wide_fused3_wrapper(new_xyz.cpu(), xyz.cpu(), feats.cpu(), wcat.cpu(), b1.cpu(), wx.cpu(),
                                                     ext_cpu.pointnet2_cpu.ball_pack_wrapper(idx.cpu(), xyz.cpu(), new_xyz.cpu()), b2.cpu(), b3.cpu(),
                                                     (cin, c1p, c2p, c3), oc3, 4)
        assert torch.equal(oc3, oc)
    else:
        assert (cin, c1, c2, c3) != (256, 256, 256, 512), "the RCNN GroupAll shape must be served"
    # and within f32 rounding of plain library arithmetic (the padding changes nothing)
    ix = idx.long().view(b, m * ns)
    base = torch.gather(Pg.view(b, n, c1p), 1, ix.unsqueeze(-1).expand(-1, -1, c1p))
    d = torch.gather(xyz, 1, ix.unsqueeze(-1).expand(-1, -1, 3)).view(b, m, ns, 3) - new_xyz.unsqueeze(2)
    a = (base + d.view(b, m * ns, 3) @ wx).clamp_(min=0)
    y = torch.addmm(b3, torch.addmm(b2, a.view(-1, c1p), w2).clamp_(min=0), w3).clamp_(min=0).view(b * m, ns, c3).amax(1)
    assert (og[:, :, 4:].reshape(b * m, c3) - y).abs().max().item() < 5e-5 * max(1.0, y.abs().max().item())


def test_packed_layer_host_row_count_ragged(ext):
    """The layer kernel as a plain row-major GEMM layer: row counts that are not multiples of 64, strided input, ReLU off/on."""
    rng = np.random.default_rng(9)
    # (K >= 256 and fewer than 256 tiles of 64 rows x 128 columns -> the 32-row-tile kernel; otherwise the 64-row kernels)
    # (K = 128 and 512 or more tiles x column blocks -> persistent workgroups that stream over the row tiles, ragged last tile read
    #  as zeros through the buffer's bounds check: 40000 rows x 128 columns = 625 tiles, 20037 rows x 256 columns = 2 x 314)
    for R, K, N, relu in ((1, 128, 128, True), (63, 256, 128, False), (200, 128, 384, True), (8192, 512, 256, False),
                          (33, 384, 256, True), (2048, 1536, 512, True), (800, 512, 256, False), (40000, 128, 128, True),
                          (20037, 128, 256, False),
                          # K >= 256 and 256 or more (tile, column block) items: the persistent pipeline (round 4) -- fewer workgroups than
                          # items (each walks several, the next item's first panel fetched behind the last panel of the running one),
                          # ragged last tile, 2 / 3 / 4 panels
                          (20037, 256, 256, True), (33001, 384, 128, False), (16400, 512, 384, True)):
        a_full = T(rng.standard_normal((R, K + 8)).astype(np.float32))
        a = a_full[:, 4:4 + K] if False else a_full[:, :K]           # row stride K + 8, 16-byte aligned
        w = T((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))
        bias = T(rng.standard_normal(N).astype(np.float32))
        out = torch.full((R + 1, N), float("nan"), device=DEV)
        ext.pointnet2.packed_layer_wrapper(a, w, bias, relu, out[:R])
        assert torch.isnan(out[R]).all()                               # nothing written past the last row
        want = torch.empty((R, N))
        ext_cpu.pointnet2_cpu.packed_layer_wrapper(a.cpu(), w.cpu(), bias.cpu(), relu, want)
        assert torch.equal(out[:R].cpu(), want)


def test_packed_layer_interp_persistent_forms(ext):
    """prcnn_packed_layer_interp (FP layer 1 over the skip features + the interpolated coarse product in the epilogue) at sizes that take
    the persistent kernels of round 4 -- K = 128: weights resident, rows streamed (625 tiles x 2 column blocks); K >= 256: the panel
    pipeline that runs on into the next tile (628 / 771 items for 512 workgroups) -- against the oracle's restatement, bit for bit;
    ragged last tiles, ReLU on and off."""
    rng = np.random.default_rng(77)
    for b, n, m, K, N, relu in ((5, 8000, 2000, 128, 256, True), (3, 6679, 1500, 256, 256, False), (2, 8200, 1000, 384, 384, True)):
        rows = b * n
        a = T(rng.standard_normal((rows, K)).astype(np.float32))
        w = T((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))
        bias = T(rng.standard_normal(N).astype(np.float32))
        G = T(rng.standard_normal((b, m, N)).astype(np.float32))
        idx = T(rng.integers(0, m, (b, n, 3)).astype(np.int32))
        wt = rng.uniform(0.05, 1.0, (b, n, 3)).astype(np.float32)
        wt /= wt.sum(-1, keepdims=True)
        weight = T(wt.astype(np.float32))
        out = torch.full((rows + 1, N), float("nan"), device=DEV)
        ext.pointnet2.packed_layer_interp_wrapper(a, w, bias, relu, out[:rows], G, idx, weight)
        assert torch.isnan(out[rows]).all()
        want = torch.empty((rows, N))
        ext_cpu.pointnet2_cpu.packed_layer_interp_wrapper(a.cpu(), w.cpu(), bias.cpu(), relu, want, G.cpu(), idx.cpu(), weight.cpu())
        assert torch.equal(out[:rows].cpu(), want), (b, n, m, K, N, float((out[:rows].cpu() - want).abs().max()))


@pytest.mark.parametrize("c1,c2,c3,ns", [(16, 16, 32, 16), (32, 32, 64, 32)])
def test_xyz_level_over_packed_rows_bit_identical(ext, c1, c2, c3, ns):
    """RPN SA1 (coordinates only) over the distinct rows == the all-rows VALU kernel == the oracle, bit for bit; output slice
    inside a wider (MSG-concatenated) buffer, other columns untouched."""
    rng = np.random.default_rng(c3)
    b, n, m = 3, 2048, 333
    xyz = T(rng.uniform(-3, 3, (b, n, 3)).astype(np.float32))
    new_xyz = T(rng.uniform(-3, 3, (b, m, 3)).astype(np.float32))
    idx_np, cnt = ball_like_idx(rng, b, m, n, ns, 2.5)
    idx = T(idx_np)
    w1 = T((rng.standard_normal((3, c1)) * 0.7).astype(np.float32)); b1 = T(rng.standard_normal(c1).astype(np.float32) * 0.2)
    w2 = T((rng.standard_normal((c1, c2)) / np.sqrt(c1)).astype(np.float32)); b2 = T(rng.standard_normal(c2).astype(np.float32) * 0.2)
    w3 = T((rng.standard_normal((c2, c3)) / np.sqrt(c2)).astype(np.float32)); b3 = T(rng.standard_normal(c3).astype(np.float32) * 0.2)
    full = torch.full((b, m, c3 + 8), -1.0, device=DEV)
    ext.pointnet2.sa_xyz_mlp_wrapper(new_xyz, xyz, idx, w1, b1, w2, b2, w3, b3, full, 4)
    got = torch.full((b, m, c3 + 8), -1.0, device=DEV)
    pk = ext.pointnet2.ball_pack_wrapper(idx, xyz, new_xyz)
    ext.pointnet2.sa_xyz_mlp_packed_wrapper(new_xyz, xyz, pk, w1, b1, w2, b2, w3, b3, got, 4)
    assert torch.equal(got, full)
    want = torch.full((b, m, c3 + 8), -1.0)
    ext_cpu.pointnet2_cpu.sa_xyz_mlp_wrapper(new_xyz.cpu(), xyz.cpu(), idx.cpu(), w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu(), w3.cpu(),
                                             b3.cpu(), want, 4)
    assert torch.equal(got.cpu(), want)
    assert (got[..., 4:4 + c3] > 0).float().mean() > 0.2


def _rpn_tail_case(rng, b, n, m, n_reg):
    known = T(rng.standard_normal((b, m, 256)).astype(np.float32))
    idx = T(rng.integers(0, m, (b, n, 3)).astype(np.int32))
    wgt = rng.random((b, n, 3)).astype(np.float32) + 0.05
    wgt = T((wgt / wgt.sum(axis=2, keepdims=True)).astype(np.float32))
    wcat = (rng.standard_normal((768, 128)) / 11).astype(np.float32)
    wcat[:256] *= 0.7
    wcat[640:, n_reg:] = 0                                           # the narrow last layer arrives zero-padded to 128 columns
    bcat = (rng.standard_normal((5, 128)) * 0.1).astype(np.float32)
    bcat[4, n_reg:] = 0
    wc2 = T((rng.standard_normal(128) / 11).astype(np.float32)); bc2 = T(rng.standard_normal(1).astype(np.float32))
    return known, idx, wgt, T(wcat), T(bcat), wc2, bc2


@pytest.mark.parametrize("b,n,m,n_reg", [(1, 1, 3, 76), (2, 1000, 300, 76), (3, 64, 17, 128), (1, 4100, 700, 4)])
def test_rpn_tail_kernel_equals_the_chain_of_oracle_functions(ext, b, n, m, n_reg):
    """csrc/rpn_tail.hip (interpolation -> FP module 256-128-128 -> cls 128-128-1 and reg 128-128-n_reg, one kernel) vs the
    oracle's three_interpolate -> five MFMA-ordered layers -> GEMV, BIT FOR BIT: ragged row counts (not multiples of 64, fewer
    than one tile, several tiles per workgroup), regression widths 4 / 76 / 128, nothing written outside the outputs."""
    rng = np.random.default_rng(b * 1000 + n)
    known, idx, wgt, wcat, bcat, wc2, bc2 = _rpn_tail_case(rng, b, n, m, n_reg)
    guard = torch.full((b * n + 8, n_reg), float("nan"), device=DEV)
    reg = guard[:b * n].view(b, n, n_reg)
    cls = torch.full((b, n, 1), float("nan"), device=DEV)
    feats = torch.full((b, n, 128), float("nan"), device=DEV)
    ext.pointnet2.rpn_tail_wrapper(known, idx, wgt, wcat, bcat, wc2, bc2, feats, cls, reg)
    assert torch.isnan(guard[b * n:]).all()
    wf, wc, wr = torch.empty((b, n, 128)), torch.empty((b, n, 1)), torch.empty((b, n, n_reg))
    ext_cpu.pointnet2_cpu.rpn_tail_wrapper(known.cpu(), idx.cpu(), wgt.cpu(), wcat.cpu(), bcat.cpu(), wc2.cpu(), bc2.cpu(), wf, wc, wr)
    assert torch.equal(feats.cpu(), wf) and torch.equal(cls.cpu(), wc) and torch.equal(reg.cpu(), wr)
    assert float(wf.abs().max()) > 0 and float(wr.abs().max()) > 0


@pytest.mark.parametrize("b,n,m,n_reg", [(1, 1, 3, 76), (2, 1000, 300, 76), (3, 64, 17, 128), (1, 4100, 700, 4), (8, 16384, 4096, 76)])
def test_rpn_tail_lin_kernel_equals_its_oracle_restatement(ext, b, n, m, n_reg):
    """csrc/rpn_tail.hip rpn_tail_lin_kernel (FP layer 1 applied at the coarse level: the 128-wide product is interpolated, + bias,
    ReLU, then FP layer 2 and both heads in one kernel) vs oracle/ext_cpu.py rpn_tail_lin_wrapper, BIT FOR BIT: ragged sizes, several
    tiles per workgroup, regression widths 4 / 76 / 128, the benchmarked B = 8 shape; nothing written outside the outputs.  And the
    new association agrees with the reference's order (interpolate the 256-wide features, then the layer) to ~1e-6."""
    rng = np.random.default_rng(b * 1000 + n + 1)
    known, idx, wgt, wcat, bcat, wc2, bc2 = _rpn_tail_case(rng, b, n, m, n_reg)
    P = ext.pointnet2
    G = P.packed_layer_wrapper(known.view(b * m, 256), wcat[:256].contiguous(), torch.zeros(128, device=DEV), False,
                               torch.empty((b * m, 128), device=DEV)).view(b, m, 128)
    wlin = wcat[256:].contiguous()
    guard = torch.full((b * n + 8, n_reg), float("nan"), device=DEV)
    reg = guard[:b * n].view(b, n, n_reg)
    cls = torch.full((b, n, 1), float("nan"), device=DEV)
    feats = torch.full((b, n, 128), float("nan"), device=DEV)
    P.rpn_tail_lin_wrapper(G, idx, wgt, wlin, bcat, wc2, bc2, feats, cls, reg)
    assert torch.isnan(guard[b * n:]).all()
    wf, wc, wr = torch.empty((b, n, 128)), torch.empty((b, n, 1)), torch.empty((b, n, n_reg))
    ext_cpu.pointnet2_cpu.rpn_tail_lin_wrapper(G.cpu(), idx.cpu(), wgt.cpu(), wlin.cpu(), bcat.cpu(), wc2.cpu(), bc2.cpu(), wf, wc, wr)
    assert torch.equal(feats.cpu(), wf) and torch.equal(cls.cpu(), wc) and torch.equal(reg.cpu(), wr)
    # against the reference's order of operations (the round-2 kernel): same numbers up to the association of the sums
    f0 = torch.empty((b, n, 128), device=DEV); c0 = torch.empty((b, n, 1), device=DEV); r0 = torch.empty((b, n, n_reg), device=DEV)
    P.rpn_tail_wrapper(known, idx, wgt, wcat, bcat, wc2, bc2, f0, c0, r0)
    for got, ref in ((feats, f0), (cls, c0), (reg, r0)):
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("b,n,m,wild", [(1, 1, 3, False), (2, 1000, 300, False), (1, 4100, 700, True), (3, 65, 17, True), (8, 16384, 4096, False)])
def test_rpn_tail_lin_with_the_decode_inside_equals_decode_of_the_stored_rows(ext, b, n, m, wild):
    """rpn_tail_lin_kernel<true> (round 5: the proposal layer's decode rides in the fused RPN tail, the (B, N, 76) regression tensor
    never reaches HBM): features and scores are the bits of the row-storing kernel, and the boxes are BIT FOR BIT (a) this package's
    decode_bbox_target (pinned to the reference's by fixture g7) + proposal_layer.py:31's y shift over the rows the other kernel
    stores, and (b) what csrc/proposal.hip's rpn_decode_kernel makes of them -- checked through the whole proposal layer: the RoIs
    of prcnn_rpn_proposals_boxes(boxes) equal those of prcnn_rpn_proposals(reg).  ``wild``: heading / location residuals scaled to
    +-1e5 and a few NaN / inf rows (arg-max with NaN, the branch-free fmod at large quotients); ragged tiles; nothing written
    outside the outputs."""
    C = pkg("config")
    cfg = C.default_eval_cfg()
    dec = pkg("bbox_transform").decode_bbox_target
    rng = np.random.default_rng(b * 1000 + n + 7)
    n_reg = 76
    known, idx, wgt, wcat, bcat, wc2, bc2 = _rpn_tail_case(rng, b, n, m, n_reg)
    if wild:
        wcat = wcat.clone(); bcat = bcat.clone()
        wcat[640:, 24:48] *= 3e4; wcat[640:, 61:73] *= 1e5               # residual columns
        bcat[4, 61] = 3e38; bcat[4, 30] = -1e30
    P = ext.pointnet2
    G = P.packed_layer_wrapper(known.view(b * m, 256), wcat[:256].contiguous(), torch.zeros(128, device=DEV), False,
                               torch.empty((b * m, 128), device=DEV)).view(b, m, 128)
    if wild and b * m > 10:
        G.view(-1, 128)[3] = float("nan"); G.view(-1, 128)[7] = float("inf")     # rows that interpolate these carry NaN / inf through the heads
    wlin = wcat[256:].contiguous()
    xyz = T(rng.uniform([-40, -1, 0.5], [40, 3, 70], (b, n, 3)).astype(np.float32))
    feats = torch.empty((b, n, 128), device=DEV); cls = torch.empty((b, n, 1), device=DEV); reg = torch.empty((b, n, n_reg), device=DEV)
    P.rpn_tail_lin_wrapper(G, idx, wgt, wlin, bcat, wc2, bc2, feats, cls, reg)
    anchor = [float(v) for v in np.asarray(cfg.CLS_MEAN_SIZE[0], dtype=np.float32)]
    assert P.rpn_tail_boxes_supported(n_reg, cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE, cfg.RPN.NUM_HEAD_BIN, cfg.RPN.LOC_XZ_FINE)
    assert not P.rpn_tail_boxes_supported(n_reg, cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE / 2, cfg.RPN.NUM_HEAD_BIN, True)
    guard = torch.full((b * n + 8, 7), float("nan"), device=DEV)
    boxes = guard[:b * n].view(b, n, 7)
    f2 = torch.full((b, n, 128), float("nan"), device=DEV); c2 = torch.full((b, n, 1), float("nan"), device=DEV)
    P.rpn_tail_lin_boxes_wrapper(G, idx, wgt, wlin, bcat, wc2, bc2, n_reg, cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE, cfg.RPN.NUM_HEAD_BIN,
                                 cfg.RPN.LOC_XZ_FINE, anchor, xyz, f2, c2, boxes)
    torch.cuda.synchronize()
    assert torch.isnan(guard[b * n:]).all()
    same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=1.25e38), torch.nan_to_num(y, nan=1.25e38))
    assert same(f2, feats) and same(c2, cls)
    want = dec(xyz.view(-1, 3), reg.view(-1, n_reg), anchor_size=torch.tensor(anchor, device=DEV), loc_scope=cfg.RPN.LOC_SCOPE,
               loc_bin_size=cfg.RPN.LOC_BIN_SIZE, num_head_bin=cfg.RPN.NUM_HEAD_BIN, get_xz_fine=cfg.RPN.LOC_XZ_FINE, get_y_by_bin=False,
               get_ry_fine=False)
    want[:, 1] += want[:, 3] / 2
    got = boxes.view(-1, 7)
    bad = ~((got == want) | (torch.isnan(got) & torch.isnan(want)))
    assert not bool(bad.any()), (int(bad.sum()), got[bad.any(1)][:4], want[bad.any(1)][:4], reg.view(-1, n_reg)[bad.any(1)][:4, 49:73])
    if wild:
        assert float(reg[..., 61:73].abs().max()) > 1e4 or b * n < 10
    # (b) through the proposal layer: decoded-in-the-tail boxes vs rows decoded by rpn_decode_kernel
    if not wild and n >= 1000:
        scores = cls.view(b, n).contiguous()
        M = 100
        r1 = torch.empty((b, M, 7), device=DEV); s1 = torch.empty((b, M), device=DEV)
        r2 = torch.empty((b, M, 7), device=DEV); s2 = torch.empty((b, M), device=DEV)
        ext.iou3d.rpn_proposals(xyz, scores, reg, anchor, cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE, cfg.RPN.NUM_HEAD_BIN, True, 9000, M, 0.8,
                                False, r1, s1)
        ext.iou3d.rpn_proposals_boxes(scores, boxes, 9000, M, 0.8, False, r2, s2)
        assert torch.equal(r1, r2) and torch.equal(s1, s2) and float(r1.abs().sum()) > 0


def test_branch_free_fmod_by_two_pi_equals_fmodf(ext):
    """The fused decode takes torch.remainder's fmod without the library's loop (csrc/rpn_tail.hip fmod_two_pi: three exact f64
    reduction steps): bit for bit fmodf(a, (float)(2 pi)) on the same device -- uniform values, every binade up to 3.4e38 in both
    signs, neighbours of multiples of 2 pi, zeros, denormals, inf, NaN."""
    rng = np.random.default_rng(9)
    two_pi = np.float32(2.0 * np.pi)
    mult = (np.arange(1, 200001, dtype=np.float64) * float(two_pi)).astype(np.float32)
    parts = [rng.uniform(-100, 100, 1 << 20).astype(np.float32),
             (rng.uniform(1, 2, 1 << 20) * np.exp2(rng.integers(-149, 128, 1 << 20))).astype(np.float32) * rng.choice([-1, 1], 1 << 20).astype(np.float32),
             mult, np.nextafter(mult, np.float32(0)), np.nextafter(mult, np.float32(np.inf)), -mult,
             (mult.astype(np.float64) * 1e6).astype(np.float32), (mult.astype(np.float64) * 1e30).astype(np.float32),
             np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan, two_pi, -two_pi], np.float32)]
    a = T(np.concatenate(parts))
    mine, lib = ext.pointnet2.selftest_fmod_two_pi(a)
    mi, li = mine.view(torch.int32), lib.view(torch.int32)
    bad = (mi != li) & ~(torch.isnan(mine) & torch.isnan(lib))
    assert not bool(bad.any()), (int(bad.sum()), a[bad][:8], mine[bad][:8], lib[bad][:8])
    assert bool(torch.isnan(mine[-3])) and bool(torch.isnan(mine[-5])) and float(mine[-2]) == 0.0


def test_rpn_tail_kernel_equals_the_separate_kernels_at_the_batch8_shape(ext):
    """B = 8 x 16384 points from 4096 coarse points (the benchmarked step): the fused kernel gives the SAME BITS as
    three_interpolate_pm -> packed_layer x5 -> rows_dot on the GPU."""
    rng = np.random.default_rng(5)
    b, n, m, n_reg = 8, 16384, 4096, 76
    known, idx, wgt, wcat, bcat, wc2, bc2 = _rpn_tail_case(rng, b, n, m, n_reg)
    feats = torch.empty((b, n, 128), device=DEV); cls = torch.empty((b, n, 1), device=DEV); reg = torch.empty((b, n, n_reg), device=DEV)
    ext.pointnet2.rpn_tail_wrapper(known, idx, wgt, wcat, bcat, wc2, bc2, feats, cls, reg)
    P = ext.pointnet2
    x = torch.empty((b, n, 256), device=DEV)
    P.three_interpolate_pm_wrapper(known, idx, wgt, x, 0)
    lay = lambda a, k0, k1, i, relu, width: P.packed_layer_wrapper(a, wcat[k0:k1].contiguous(), bcat[i].contiguous(), relu,
                                                                  torch.empty((b * n, width), device=DEV))
    h = lay(x.view(b * n, 256), 0, 256, 0, True, 128)
    f2 = lay(h, 256, 384, 1, True, 128)
    hc = lay(f2, 384, 512, 2, True, 128)
    c2 = P.rows_dot_wrapper(hc, wc2.view(128, 1).contiguous(), bc2, torch.empty((b * n, 1), device=DEV))
    hr = lay(f2, 512, 640, 3, True, 128)
    r2 = lay(hr, 640, 768, 4, False, n_reg)
    assert torch.equal(feats.view(-1, 128), f2) and torch.equal(cls.view(-1, 1), c2) and torch.equal(reg.view(-1, n_reg), r2)


@pytest.mark.parametrize("rows,K,N,n_store", [(1000, 128, 128, 128), (70, 256, 256, 256), (513, 512, 128, 77), (256, 128, 384, 384)])
def test_split_bf16_layer_is_an_f32_class_product(ext, rows, K, N, n_store):
    """EXPERIMENT (numerics switch PRCNN_SPLIT_BF16, default off; csrc/split_bf16.hip): act(A @ W + b) as six bf16 MFMAs per k-step
    over operands split exactly into three bf16 pieces.  Not the f32 kernels' bits (those are pinned by the oracle above): held to a
    float64 product of the same f32 operands within 1e-6 of each row's sum of |terms| -- the f32 fma chain's own error level --, on
    ragged row counts (the last 256-row tile partly filled), a narrow store (n_store < N) and several column blocks; and A = I hands
    an asymmetric W back exactly (the operand layouts of v_mfma_f32_32x32x16_bf16)."""
    g = torch.Generator().manual_seed(rows + K)
    a = torch.randn((rows, K), generator=g).relu_().to(DEV)
    w = (torch.randn((K, N), generator=g) / K ** 0.5).to(DEV)
    b = torch.randn((N,), generator=g).to(DEV)
    out = torch.full((rows + 3, n_store), -7.0, device=DEV)
    ext.pointnet2.rows_layer_bf16x3_wrapper(a, w, b, True, out[:rows])
    ref = torch.relu(a.double() @ w.double() + b.double())[:, :n_store]
    scale = (a.double().abs() @ w.double().abs() + b.double().abs())[:, :n_store]
    assert float(((out[:rows].double() - ref).abs() / scale).max()) < 1e-6
    assert (out[rows:] == -7.0).all()                                   # nothing stored past the ragged end
    f32 = torch.empty((rows, n_store), device=DEV)
    ext.pointnet2.packed_layer_wrapper(a, w, b, True, f32)
    assert float((out[:rows] - f32).abs().max()) < 2e-5
    if K == N == 128:
        eye = torch.eye(K, device=DEV)
        w2 = (torch.arange(K * N, dtype=torch.float32, device=DEV).view(K, N) / 7.0).contiguous()
        o2 = torch.empty((K, N), device=DEV)
        ext.pointnet2.rows_layer_bf16x3_wrapper(eye, w2, torch.zeros(N, device=DEV), False, o2)
        assert torch.equal(o2, w2)
