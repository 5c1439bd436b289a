"""Layer builders with the reference's public names and state-dict layout
(pointrcnn/pointnet2_lib/pointnet2/pytorch_utils.py:5-236): ``SharedMLP`` is a Sequential of
``layer{i}`` blocks, each block a Sequential with children named ``conv`` / ``bn`` /
``activation`` (``in`` for instance norm), and ``bn`` is itself a one-child Sequential named
``bn`` -- hence checkpoint keys like ``...layer0.bn.bn.running_mean``.
"""
import torch.nn as nn


class _Norm(nn.Sequential):
    """One-child wrapper: the doubled ``bn.bn`` key of the reference (_BNBase, :104-111)."""

    def __init__(self, channels, kind, name=""):
        super().__init__()
        self.add_module(name + "bn", kind(channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_Norm):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, nn.BatchNorm1d, name)


class BatchNorm2d(_Norm):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm2d, name)


class _ConvBlock(nn.Sequential):
    """conv (+bias unless bn) with optional bn / activation / instance norm before (preact) or
    after it -- ordering and child names as _ConvBase (:35-101)."""

    conv_cls = None
    bn_cls = None
    in_cls = None

    def __init__(self, in_size, out_size, *, kernel_size, stride, padding, activation=None, bn=False,
                 init=nn.init.kaiming_normal_, bias=True, preact=False, name="", instance_norm=False):
        super().__init__()
        conv = self.conv_cls(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                             bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        norm_width = in_size if preact else out_size
        extras = []
        if bn:
            extras.append((name + "bn", self.bn_cls(norm_width)))
        if activation is not None:
            extras.append((name + "activation", activation))
        if not bn and instance_norm:
            extras.append((name + "in", self.in_cls(norm_width, affine=False, track_running_stats=False)))
        parts = extras + [(name + "conv", conv)] if preact else [(name + "conv", conv)] + extras
        for key, mod in parts:
            self.add_module(key, mod)


class Conv1d(_ConvBlock):
    conv_cls, bn_cls, in_cls = nn.Conv1d, BatchNorm1d, nn.InstanceNorm1d

    def __init__(self, in_size, out_size, *, kernel_size=1, stride=1, padding=0, activation=nn.ReLU(inplace=True),
                 bn=False, init=nn.init.kaiming_normal_, bias=True, preact=False, name="", instance_norm=False):
        super().__init__(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                         activation=activation, bn=bn, init=init, bias=bias, preact=preact, name=name,
                         instance_norm=instance_norm)


class Conv2d(_ConvBlock):
    conv_cls, bn_cls, in_cls = nn.Conv2d, BatchNorm2d, nn.InstanceNorm2d

    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name="", instance_norm=False):
        super().__init__(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                         activation=activation, bn=bn, init=init, bias=bias, preact=preact, name=name,
                         instance_norm=instance_norm)


class SharedMLP(nn.Sequential):
    """Chain of 1x1 Conv2d blocks ``layer0..`` over a (B, C, npoint, nsample) tensor (:5-32)."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False,
                 name="", instance_norm=False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # the very first pre-activated layer has no bn/act
            self.add_module(name + "layer%d" % i,
                            Conv2d(args[i], args[i + 1], bn=bn and not plain,
                                   activation=None if plain else activation, preact=preact,
                                   instance_norm=instance_norm))


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False, init=None,
                 preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        extras = []
        if bn:
            extras.append((name + "bn", BatchNorm1d(in_size if preact else out_size)))
        if activation is not None:
            extras.append((name + "activation", activation))
        parts = extras + [(name + "fc", fc)] if preact else [(name + "fc", fc)] + extras
        for key, mod in parts:
            self.add_module(key, mod)
