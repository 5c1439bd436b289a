"""Inference-time execution of a ``SharedMLP`` (pytorch_utils.py:5-32 of the reference): the same
function, computed as  bias-free 1x1 conv (a dense f32 GEMM on MFMA via rocBLAS) + ONE fused
bias/ReLU pass per hidden layer, and -- when the MLP is followed by the max over nsample --
the last layer's bias/ReLU folded into the pooling kernel (csrc/mlp_epilogue.hip).

BatchNorm in eval mode is an affine map per channel; it is folded into the convolution:
    w' = w * gamma / sqrt(var + eps),  b' = beta - mean * gamma / sqrt(var + eps) (+ conv bias)
which changes results only by f32 rounding (~1e-7 relative; parity tolerance is 1e-4).
Anything this module does not recognise (pre-activation, instance norm, non-ReLU activation,
training mode) falls back to calling the nn.Module itself.
"""
import torch
import torch.nn as nn

from . import pointnet2_utils


ENABLED = True   # tests flip this to run the unfused nn.Module path (reference operation order)


def _fold_block(block):
    """-> (w (Cout,Cin), b (Cout)) or None if the block is not conv[+bn]+relu in post-activation order."""
    names = [n for n, _ in block.named_children()]
    if not names or names[0] != "conv" or any(n not in ("conv", "bn", "activation") for n in names):
        return None
    conv = block.conv
    if not isinstance(conv, (nn.Conv1d, nn.Conv2d)) or any(k != 1 for k in conv.kernel_size):
        return None
    if "activation" not in names or not isinstance(block.activation, nn.ReLU):
        return None
    w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels)
    b = conv.bias.detach() if conv.bias is not None else torch.zeros(conv.out_channels, device=w.device, dtype=w.dtype)
    if "bn" in names:
        bn = block.bn[0]
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        w = w * scale[:, None]
        b = (b - bn.running_mean) * scale + bn.bias.detach()
    return w.contiguous(), b.contiguous()


def folded_layers(mlp):
    """Cached list of folded (w, b) for an eval-mode SharedMLP; None if unsupported."""
    cache = mlp.__dict__.get("_prcnn_folded")
    if cache is None or cache[0] != _signature(mlp):
        layers = []
        for block in mlp.children():
            fb = _fold_block(block)
            if fb is None:
                layers = None
                break
            layers.append(fb)
        cache = (_signature(mlp), layers)
        mlp.__dict__["_prcnn_folded"] = cache
    return cache[1]


def _signature(mlp):
    """Identity + version of EVERY parameter and buffer (BN running statistics included): a partial load or an in-place
    edit of any of them invalidates the folded weights."""
    ts = list(mlp.parameters()) + list(mlp.buffers())
    return (ts[0].device,) + tuple((t.data_ptr(), t._version) for t in ts)


def run(mlp, x, pool):
    """x (B, Cin, npoint, nsample).  pool=True -> (B, Cout, npoint) = max over nsample of the MLP
    output; pool=False -> (B, Cout, npoint, nsample)."""
    layers = None if (mlp.training or not ENABLED) else folded_layers(mlp)
    if layers is None:
        y = mlp(x)
        return torch.nn.functional.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1) if pool else y
    ext = pointnet2_utils.pointnet2
    B, _, npoint, ns = x.shape
    cur = x.reshape(B, x.shape[1], npoint * ns)
    last = len(layers) - 1
    for i, (w, b) in enumerate(layers):
        cur = torch.matmul(w, cur)                      # (B, Cout, npoint*ns): f32 GEMM on MFMA
        if i < last or not pool:
            ext.bias_relu_inplace_wrapper(cur, b)
    if not pool:
        return cur.view(B, -1, npoint, ns)
    out = torch.empty((B, cur.shape[1], npoint), dtype=cur.dtype, device=cur.device)
    ext.maxpool_bias_relu_wrapper(cur.view(B, -1, npoint, ns), layers[last][1], out)
    return out
