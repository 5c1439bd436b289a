"""Operator wrappers with the reference's public names
(pointrcnn/pointnet2_lib/pointnet2/pointnet2_utils.py): furthest_point_sample, gather_operation,
three_nn, three_interpolate, grouping_operation, ball_query, QueryAndGroup, GroupAll.

``pointnet2`` below is the extension module (the drop-in ``pointnet2_cuda`` over libprcnn_hip.so),
exactly where the reference has ``import pointnet2_cuda as pointnet2`` (:7).  Outputs are
allocated on the inputs' device with the same dtypes/shapes the reference allocates; there is
no CPU implementation here -- CPU tensors are rejected by the extension.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ..dropin import pointnet2_cuda as pointnet2

# True: QueryAndGroup runs the reference's own sequence over the reference's entry points only (pointnet2_utils.py:241-264:
# ball_query, two grouping_operation calls, centre subtraction, cat) instead of the one fused extension call -- what a user of
# the reference's Python gets from the drop-in modules (eval_rcnn.reference_api_only(); bench.py config.dropin_module_scenes_per_s).
# Also taken when the extension module offers no fused entry (the compiled dropin_native modules export the reference's 9 names).
REFERENCE_ORDER = False


def _contig(*ts):
    for t in ts:
        assert t.is_contiguous(), "operator inputs must be contiguous"


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) i32, starting at index 0 (:12-29)."""
        _contig(xyz)
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint) (:42-60)."""
        _contig(features, idx)
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2, idx (B,n,3)) (:79-98)."""
        _contig(unknown, known)
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((B, N, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, N, 3), dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,C,m), idx/weight (B,n,3) -> (B,C,n) (:111-131)."""
        _contig(features, idx, weight)
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros((B, c, m), dtype=torch.float32, device=grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample) (:159-177)."""
        _contig(features, idx)
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, nfeatures, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample), zero-filled first (:203-221)."""
        _contig(new_xyz, xyz)
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class _QueryAndGroupFused(Function):
    """ball_query + group(xyz) - centre + group(features) + cat in one extension call
    (prcnn_query_and_group); backward re-uses the grouping gradient kernel."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz, features):
        _contig(xyz, new_xyz)
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        C = 0 if features is None else features.size(1)
        if features is not None:
            _contig(features)
        idx = torch.empty((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        out = torch.empty((B, 3 + C, npoint, nsample), dtype=torch.float32, device=xyz.device)
        pointnet2.query_and_group_wrapper(B, N, npoint, C, radius, nsample, new_xyz, xyz, features, idx, out)
        ctx.saved = (idx, N, C)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, N, C = ctx.saved
        if C == 0:
            return None, None, None, None, None
        B, _, npoint, nsample = grad_out.size()
        g = grad_out[:, 3:].contiguous()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, g, idx, grad_features)
        return None, None, None, None, grad_features


class QueryAndGroup(nn.Module):
    """(:228-264) new_features (B, 3+C, npoint, nsample) = cat(xyz[idx]-centre, features[idx])."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        if self.use_xyz and not REFERENCE_ORDER and hasattr(pointnet2, "query_and_group_wrapper"):
            return _QueryAndGroupFused.apply(self.radius, self.nsample, xyz, new_xyz, features)
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if not self.use_xyz:
            return grouping_operation(features, idx)
        # the reference's operation order: neighbours' coordinates as channels, made relative to their centre, features behind
        rel = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        rel -= new_xyz.transpose(1, 2).unsqueeze(-1)
        return rel if features is None else torch.cat([rel, grouping_operation(features, idx)], dim=1)


class GroupAll(nn.Module):
    """(:267-290) one group holding every point: (B, 3+C, 1, N)."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features
