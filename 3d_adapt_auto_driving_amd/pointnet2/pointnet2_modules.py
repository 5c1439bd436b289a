"""Set-abstraction / feature-propagation modules with the reference's public names and
constructor signatures (pointrcnn/pointnet2_lib/pointnet2/pointnet2_modules.py:10-160).
Child names (``groupers``, ``mlps``, ``mlp``) match so reference checkpoints load.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils
from . import fused_mlp


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = "max_pool"

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum(mlp[-1]),npoint)."""
        if new_xyz is None and self.npoint is not None:
            sel = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            xyz_t = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(xyz_t, sel).transpose(1, 2).contiguous()

        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            if not self.training and self.pool_method == "max_pool":
                # inference: GEMM + fused epilogues, pooling fused with the last bias/ReLU
                pooled.append(fused_mlp.run(mlp, grouper(xyz, new_xyz, features), pool=True))
                continue
            x = mlp(grouper(xyz, new_xyz, features))          # (B, mlp[-1], npoint, nsample)
            if self.pool_method == "max_pool":
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
            elif self.pool_method == "avg_pool":
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
            else:
                raise NotImplementedError(self.pool_method)
            pooled.append(x.squeeze(-1))
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale grouping set abstraction."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method="max_pool", instance_norm=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.pool_method = pool_method
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn, instance_norm=instance_norm))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method="max_pool", instance_norm=False):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz, pool_method=pool_method, instance_norm=instance_norm)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation from the 3 nearest known points."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        x = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        if not self.training:
            return fused_mlp.run(self.mlp, x.unsqueeze(-1), pool=False).squeeze(-1)
        return self.mlp(x.unsqueeze(-1)).squeeze(-1)
