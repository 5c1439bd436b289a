"""RPN backbone: 4 multi-scale SA levels + 4 FP levels (counterpart of
pointrcnn/lib/net/pointnet2_msg.py:6-70; module names SA_modules / FP_modules are checkpoint keys)."""
import torch.nn as nn

from ..pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG


def get_model(cfg, input_channels=6, use_xyz=True):
    return Pointnet2MSG(cfg, input_channels=input_channels, use_xyz=use_xyz)


class Pointnet2MSG(nn.Module):
    def __init__(self, cfg, input_channels=6, use_xyz=True):
        super().__init__()
        sa = cfg.RPN.SA_CONFIG
        self.SA_modules = nn.ModuleList()
        widths = [input_channels]          # feature width at every resolution level
        for k in range(len(sa.NPOINTS)):
            specs = [[widths[-1]] + list(m) for m in sa.MLPS[k]]
            self.SA_modules.append(PointnetSAModuleMSG(npoint=sa.NPOINTS[k], radii=sa.RADIUS[k],
                                                       nsamples=sa.NSAMPLE[k], mlps=specs, use_xyz=use_xyz,
                                                       bn=cfg.RPN.USE_BN))
            widths.append(sum(s[-1] for s in specs))
        self.FP_modules = nn.ModuleList()
        fp = cfg.RPN.FP_MLPS
        for k in range(len(fp)):
            coarse = fp[k + 1][-1] if k + 1 < len(fp) else widths[-1]
            self.FP_modules.append(PointnetFPModule(mlp=[coarse + widths[k]] + list(fp[k])))

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud):
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features = [xyz], [features]
        for sa in self.SA_modules:
            nx, nf = sa(l_xyz[-1], l_features[-1])
            l_xyz.append(nx)
            l_features.append(nf)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):   # coarse -> fine
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
        return l_xyz[0], l_features[0]
