"""PointNet++ encoder / decoder of the RPN: multi-scale set abstraction down four resolution levels, feature propagation
back up (counterpart of pointrcnn/lib/net/pointnet2_msg.py:6-70).  ``SA_modules`` and ``FP_modules`` are checkpoint key
prefixes and keep their names."""
import torch.nn as nn

from ..pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG


class Pointnet2MSG(nn.Module):
    def __init__(self, cfg, input_channels=6, use_xyz=True):
        super().__init__()
        sa_cfg, fp_widths = cfg.RPN.SA_CONFIG, cfg.RPN.FP_MLPS
        levels = len(sa_cfg.NPOINTS)

        # encoder: channel count after every level (level 0 = the raw per-point features)
        channels = [input_channels]
        encoder = []
        for lvl in range(levels):
            branch_specs = [[channels[lvl]] + list(branch) for branch in sa_cfg.MLPS[lvl]]
            encoder.append(PointnetSAModuleMSG(npoint=sa_cfg.NPOINTS[lvl], radii=sa_cfg.RADIUS[lvl],
                                               nsamples=sa_cfg.NSAMPLE[lvl], mlps=branch_specs, use_xyz=use_xyz,
                                               bn=cfg.RPN.USE_BN))
            channels.append(sum(spec[-1] for spec in branch_specs))      # branches are concatenated
        self.SA_modules = nn.ModuleList(encoder)

        # decoder k refines level k from level k+1: input = skip features of level k + output width of decoder k+1
        decoder = []
        for k, widths in enumerate(fp_widths):
            from_above = fp_widths[k + 1][-1] if k + 1 < len(fp_widths) else channels[-1]
            decoder.append(PointnetFPModule(mlp=[from_above + channels[k]] + list(widths)))
        self.FP_modules = nn.ModuleList(decoder)

    @staticmethod
    def split_cloud(cloud):
        """(B, N, 3 + C) -> coordinates (B, N, 3) and channel-major features (B, C, N) or None."""
        coords = cloud[..., :3].contiguous()
        feats = cloud[..., 3:].transpose(1, 2).contiguous() if cloud.size(-1) > 3 else None
        return coords, feats

    def forward(self, pointcloud):
        coords, feats = self.split_cloud(pointcloud)
        pyramid_xyz, pyramid_feat = [coords], [feats]
        for stage in self.SA_modules:                                   # fine -> coarse
            sub_xyz, sub_feat = stage(pyramid_xyz[-1], pyramid_feat[-1])
            pyramid_xyz.append(sub_xyz)
            pyramid_feat.append(sub_feat)
        for k in reversed(range(len(self.FP_modules))):                 # coarse -> fine
            pyramid_feat[k] = self.FP_modules[k](pyramid_xyz[k], pyramid_xyz[k + 1], pyramid_feat[k], pyramid_feat[k + 1])
        return pyramid_xyz[0], pyramid_feat[0]


def get_model(cfg, input_channels=6, use_xyz=True):
    return Pointnet2MSG(cfg, input_channels=input_channels, use_xyz=use_xyz)
