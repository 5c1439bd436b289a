"""-m gpu: the BACKWARD path of the operator API at autograd level (the three gradient kernels
were compared with the reference's kernels, but no test ran ``loss.backward()`` through the autograd Functions that call them --
pointnet2_utils.py:62-71 (GatherOperation), :133-151 (ThreeInterpolate), :179-195 (GroupingOperation) of the reference and the
fused QueryAndGroup Function of this build).  A set-abstraction module (multi-scale grouping) followed by a feature-propagation
module, in training mode, against the SAME modules with every extension operator replaced by its definition in torch.gather
terms: outputs, the gradient of the input features and the gradient of every parameter."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t_group(features, idx):
    """features (B,C,N), idx (B,M,ns) -> (B,C,M,ns) by torch.gather (differentiable: backward = scatter-add)"""
    B, C, N = features.shape
    M, ns = idx.shape[1], idx.shape[2]
    flat = idx.long().view(B, 1, M * ns).expand(-1, C, -1)
    return torch.gather(features, 2, flat).view(B, C, M, ns)


def t_interpolate(feats, idx, w):
    """feats (B,C,m), idx / w (B,n,3) -> (B,C,n): w0 f[i0] + w1 f[i1] + w2 f[i2]"""
    B, C, m = feats.shape
    n = idx.shape[1]
    out = 0
    for k in range(3):
        g = torch.gather(feats, 2, idx[:, :, k].long().view(B, 1, n).expand(-1, C, -1))
        out = out + g * w[:, :, k].unsqueeze(1)
    return out


def sa_reference(mod, xyz, features, pu):
    """_PointnetSAModuleBase.forward (pointnet2_modules.py:19-55) with torch.gather in place of the extension operators; the
    index tensors (FPS, ball query: not differentiable) come from the HIP operators"""
    sel = pu.furthest_point_sample(xyz, mod.npoint)
    new_xyz = torch.gather(xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3))
    outs = []
    for grouper, mlp in zip(mod.groupers, mod.mlps):
        idx = pu.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz.contiguous())
        rel = t_group(xyz.transpose(1, 2), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        grouped = torch.cat([rel, t_group(features, idx)], dim=1)
        y = mlp(grouped)
        outs.append(torch.nn.functional.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1))
    return new_xyz, torch.cat(outs, dim=1)


def fp_reference(mod, unknown, known, skip, known_feats, pu):
    dist, idx = pu.three_nn(unknown, known)
    inv = 1.0 / (dist + 1e-8)
    w = inv / torch.sum(inv, dim=2, keepdim=True)
    carried = t_interpolate(known_feats, idx, w)
    stacked = torch.cat([carried, skip], dim=1).unsqueeze(-1)
    return mod.mlp(stacked).squeeze(-1)


@pytest.mark.parametrize("fused", [True, False], ids=["fused_query_and_group", "reference_order"])
def test_backward_through_sa_and_fp_modules_equals_torch_gather_graph(fused):
    PM, pu = pkg("pointnet2.pointnet2_modules"), pkg("pointnet2.pointnet2_utils")
    torch.manual_seed(4)
    B, N, M, C = 3, 2048, 512, 24
    xyz = torch.from_numpy(pkg("synth").scenes(B, N, seed0=60)).to(DEV)
    sa = PM.PointnetSAModuleMSG(npoint=M, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[C, 32, 48], [C, 32, 64]], use_xyz=True, bn=True).to(DEV)
    fp = PM.PointnetFPModule(mlp=[48 + 64 + C, 64, 32], bn=True).to(DEV)
    sa.train(); fp.train()
    base = torch.randn((B, C, N), device=DEV)
    target = torch.randn((B, 32, N), device=DEV)
    results = []
    saved = pu.REFERENCE_ORDER
    for which in ("modules", "torch"):
        feats = base.clone().requires_grad_(True)
        for p in list(sa.parameters()) + list(fp.parameters()):
            p.grad = None
        if which == "modules":
            pu.REFERENCE_ORDER = not fused
            try:
                new_xyz, coarse = sa(xyz, feats)
                out = fp(xyz, new_xyz, feats, coarse)
            finally:
                pu.REFERENCE_ORDER = saved
        else:
            new_xyz, coarse = sa_reference(sa, xyz, feats, pu)
            out = fp_reference(fp, xyz, new_xyz.contiguous(), feats, coarse, pu)
        loss = ((out - target) ** 2).mean() + coarse.abs().mean()
        loss.backward()
        torch.cuda.synchronize()
        results.append((out.detach().clone(), coarse.detach().clone(), feats.grad.detach().clone(),
                        [p.grad.detach().clone() for p in list(sa.parameters()) + list(fp.parameters())]))
    (o1, c1, g1, p1), (o2, c2, g2, p2) = results
    assert float(g1.abs().max()) > 0 and all(float(p.abs().max()) > 0 for p in p1)

    def close(a, b, what):
        scale = max(1e-6, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, (what, float((a - b).abs().max()), scale)
    close(o1, o2, "fp output"); close(c1, c2, "sa output"); close(g1, g2, "d loss / d features")
    for k, (a, b) in enumerate(zip(p1, p2)):
        close(a, b, "parameter %d" % k)


def test_gather_operation_backward():
    """GatherOperation (sampling.cpp:23-33 / K5): the gradient scatters into the picked columns, summing over repeated picks."""
    pu = pkg("pointnet2.pointnet2_utils")
    torch.manual_seed(5)
    feats = torch.randn((2, 7, 300), device=DEV, requires_grad=True)
    idx = torch.randint(0, 300, (2, 90), device=DEV, dtype=torch.int32)
    idx[:, :10] = idx[:, 10:20]                                              # repeated picks accumulate
    w = torch.randn((2, 7, 90), device=DEV)
    (pu.gather_operation(feats, idx) * w).sum().backward()
    want = torch.zeros_like(feats)
    want.scatter_add_(2, idx.long().unsqueeze(1).expand(-1, 7, -1), w)
    assert float((feats.grad - want).abs().max()) <= 1e-5
