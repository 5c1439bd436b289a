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
    """Sample ``npoint`` centres (FPS), group around them at every scale, run the scale's shared MLP, pool over the
    neighbourhood and concatenate the scales."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = "max_pool"

    def _centres(self, xyz):
        chosen = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        as_channels = xyz.transpose(1, 2).contiguous()
        return pointnet2_utils.gather_operation(as_channels, chosen).transpose(1, 2).contiguous()

    def _pool(self, x):
        """(B, C, npoint, nsample) -> (B, C, npoint)"""
        window = [1, x.size(3)]
        if self.pool_method == "max_pool":
            return F.max_pool2d(x, kernel_size=window).squeeze(-1)
        if self.pool_method == "avg_pool":
            return F.avg_pool2d(x, kernel_size=window).squeeze(-1)
        raise NotImplementedError(self.pool_method)

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum(mlp[-1]),npoint)."""
        if new_xyz is None and self.npoint is not None:
            new_xyz = self._centres(xyz)
        fast = (not self.training) and self.pool_method == "max_pool"   # inference: GEMM + fused epilogues + pooling
        per_scale = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped = grouper(xyz, new_xyz, features)                    # (B, 3+C, npoint, nsample)
            per_scale.append(fused_mlp.run(mlp, grouped, pool=True) if fast else self._pool(mlp(grouped)))
        return new_xyz, torch.cat(per_scale, dim=1)


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
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n) skip features, known_feats (B,C2,m) -> (B,mlp[-1],n)."""
        if known is None:                                   # a single global feature: broadcast it to every point
            carried = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        else:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            inv = 1.0 / (dist + 1e-8)
            carried = pointnet2_utils.three_interpolate(known_feats, idx, inv / torch.sum(inv, dim=2, keepdim=True))
        stacked = carried if unknow_feats is None else torch.cat([carried, unknow_feats], dim=1)
        stacked = stacked.unsqueeze(-1)
        out = self.mlp(stacked) if self.training else fused_mlp.run(self.mlp, stacked, pool=False)
        return out.squeeze(-1)
