"""MI355X inference engine for PointRCNN: the same network function as ``PointRCNN.forward``
(pointrcnn/lib/net/point_rcnn.py:26-70 and everything it calls), re-scheduled for the hardware:

  * POINT-MAJOR activations everywhere: features are (B, N, C), a grouped neighbourhood is a
    (B*M*ns, K) row matrix, so every gather is a contiguous C-vector (16-byte lanes on consecutive
    addresses), every shared-MLP layer is ONE f32 GEMM  rows x K @ K x Cout  with the bias + ReLU
    epilogue fused into the GEMM (hipBLASLt), and the max over nsample reduces ns consecutive rows.
    Nothing is transposed between layers; RoI pooling consumes the RPN features as they are.
  * BatchNorm (eval) folded into the GEMM weights; first-layer weights permuted/zero-padded to the
    grouped row layout [features | pad | dx dy dz | 0].
  * GEOMETRY / FEATURE split: FPS, ball query and three-NN of the whole RPN backbone depend on xyz
    only.  ``geometry()`` computes them (it is the latency-bound part: FPS is serial per scene and
    occupies B CUs); ``forward(pts, geo)`` consumes them.  A pipelined caller runs geometry of
    batch i+1 on a second HIP stream while the GEMMs of batch i fill the other CUs
    (eval_rcnn.PipelinedRunner).

Values differ from the module path only by f32 rounding of the GEMMs (different kernels and
summation orders); indices (FPS / ball query / three-NN / NMS keep) are produced by the same
kernels.  Parity is tested against the reference fixtures with the 1e-4 box tolerance.
"""
import os

import torch

from ..pointnet2 import pointnet2_utils as pu
from ..pointnet2 import fused_mlp
from .._lib import has_entry
from .. import kitti_utils
from .. import roipool3d_utils
from .. import iou3d_utils


USE_ROIPOOL_CANONICAL = True   # RCNN input assembly through roipool3d_canonical_kernel (False: torch-op sequence)
USE_RCNN_POINT_MLP = os.environ.get("PRCNN_NO_POINT_MLP") is None      # RCNN entrance chain through csrc/rcnn_point_mlp.hip (False: library GEMMs + concat)
USE_XYZ_MLP = True      # coordinates-only SA scales through csrc/sa_xyz_mlp.hip (False: grouped GEMM chain)
# SA levels over the DISTINCT grouped rows only (csrc/sa_packed.hip: the back-filled copies of a ball's first hit are
# skipped, bit-identical results).  PRCNN_NO_PACK=1 is the A/B switch back to all nsample rows (csrc/sa_mlp_fused.hip).
USE_PACKED = os.environ.get("PRCNN_NO_PACK") is None
# RoI pooling fills a box holding fewer than 512 points by repeating them (roipool3d_kernel.cu:152-159): the per-point
# RCNN entrance chain and SA1 run over the DISTINCT pooled points only (bit-identical results).  PRCNN_NO_POOL_DEDUP=1: A/B.
USE_POOL_DEDUP = os.environ.get("PRCNN_NO_POOL_DEDUP") is None
# ... and the SA1 centres FPS picks among those copies are copies of one another (same coordinates, same ball, same SA1 output):
# the deeper RCNN levels drop the rows of every point that is not the first of its kind (prcnn_dup_rep + prcnn_ball_pack_rep;
# round 3: 7.5x fewer SA2 rows on LiDAR-shaped scenes, bit-identical results).  PRCNN_NO_CENTRE_DEDUP=1: A/B.
USE_CENTRE_DEDUP = os.environ.get("PRCNN_NO_CENTRE_DEDUP") is None
USE_CENTRE_SKIP = os.environ.get("PRCNN_NO_CENTRE_SKIP") is None       # ... and such a centre gets no rows of its own either (A/B)
# sampling, ball query and representative map of both sampled RCNN levels for every RoI cloud in one launch; PRCNN_NO_ROI_GEOMETRY=1:
# the six separate launches (A/B, same results)
USE_ROI_GEOMETRY = os.environ.get("PRCNN_NO_ROI_GEOMETRY") is None
USE_ROI_PACKS = os.environ.get("PRCNN_NO_ROI_PACKS") is None      # ... and their distinct-row lists out of the same launch (round 5)
USE_CENTRE_ROWS = os.environ.get("PRCNN_NO_CENTRE_ROWS") is None  # the RCNN's second level: per-point layer over the representative level-1 centres only (round 5)
USE_POOLED_ROWS = os.environ.get("PRCNN_NO_POOLED_ROWS") is None  # the RCNN entrance over the list of distinct pooled rows, not whole tiles per RoI (round 5)
USE_POINT_LAYER = os.environ.get("PRCNN_LIB_GEMM") is None     # per-point layers (FP modules, heads) on the own MFMA layer kernel
# every per-point width zero-padded to a multiple of 128 (SA level outputs, FP inputs, narrow head outputs), so that NO layer
# of the engine is left to a GEMM library: fixed summation order everywhere, reproduced bit for bit by the oracle
PAD128 = USE_PACKED and USE_POINT_LAYER
# the finest FP module and both RPN heads in one kernel (csrc/rpn_tail.hip); PRCNN_NO_RPN_TAIL=1: layer by layer (A/B, same bits)
USE_RPN_TAIL = os.environ.get("PRCNN_NO_RPN_TAIL") is None
# the scales of a wide MSG level (RPN SA3 / SA4) stage by stage, side by side in one launch per stage
USE_SCALE_BATCH = True      # (a module constant since round 6; tests patch it)
# layers 1-3 + pool of a wide scale in one kernel (csrc/sa_wide.hip); False: gather / layer / layer+pool launches
USE_SA_NARROW = True        # RPN SA2's scales without their zero padding (csrc/sa_packed.hip), their per-point parts in one product
USE_SA2_BATCH = os.environ.get("PRCNN_NO_SA2_BATCH") is None        # the two 128-wide scales of an MSG level (RPN SA2) in one launch per stage (round 5)
USE_WIDE_FUSED = True       # (a module constant since round 6; tests/test_gpu_shadow.py patches it for the layer-by-layer variant)
# ... and layer 1 inside as well where a level groups every point once (the RCNN's GroupAll level; csrc/sa_wide3.hip);
# PRCNN_NO_WIDE_FUSED3=1: the per-point layer as a launch of its own in front of csrc/sa_wide.hip (A/B, same bits)
USE_WIDE_FUSED3 = os.environ.get("PRCNN_NO_WIDE_FUSED3") is None
# RoI pooling culls by 64-point spatial groups of the scene (built with the geometry chain); PRCNN_NO_POOL_GROUPS=1: full sweep
USE_XYZ_LEVEL_EARLY = os.environ.get("PRCNN_NO_XYZ_EARLY") != "1"    # leading SA levels of a coordinates-only backbone computed with the geometry (side stream)
EARLY_LEVELS = int(os.environ.get("PRCNN_EARLY_LEVELS", "4"))
EARLY_FP = int(os.environ.get("PRCNN_EARLY_FP", "2"))                 # ... plus this many of the coarsest FP modules, over the 32 clouds of a group in one launch each (round 3: 1 -- the geometry chains were the longer side then; round 4: 3, see EARLY_TAIL; round 5: 2 -- with the RCNN's padded tiles gone the feature stream has room again and the geometry chains bind: profiles/sensitivity_probe.py, 7860 -> 7980 / LiDAR-shaped 5500 -> 5570 at K = 100)
# ... and the finest FP module + both RPN heads as well (round 4 experiment, needs EARLY_FP >= number of FP modules - 1): the RPN backbone has
# no input features, so the WHOLE RPN stage is a function of xyz and the weights.  With the geometry chains twice as fast as in round 3 the
# feature stream binds (1.10 ms of kernels per step against 0.6 on each geometry stream): EARLY_FP = 3 (all FP modules but the finest over
# the 32 clouds of a group) buys +4.6 % at K = 100 (6382 against 6103 scenes/s; K = 20: level).  The fused tail on top of it -- a 256-workgroup
# MFMA kernel of 0.7 ms per group on a geometry stream -- costs it again: 6042 at K = 100, 5158 against 5314 at K = 20.  Off.
# (round 6: the switch PRCNN_EARLY_TAIL and its branch are gone.)
# ... in between: only the coarse-level product G of the finest FP module (4096 rows per cloud, K = 256) goes with the geometry; the fused
# tail kernel, which reads it, stays on the feature stream
EARLY_G0 = os.environ.get("PRCNN_EARLY_G0", "1") == "1"
GROUP_SA = os.environ.get("PRCNN_NO_GROUP_SA") != "1"                 # ... and over all batches of a geometry group at once
USE_POOL_GROUPS = os.environ.get("PRCNN_NO_POOL_GROUPS") is None
# feature-propagation modules: the first layer is linear in front of its ReLU and the interpolation is a weighted sum, so the
# interpolated columns of the layer are applied at the COARSE level (a quarter of the rows) and the product is interpolated in the
# epilogue of the layer over the skip features (prcnn_packed_layer_interp).  Another association of the same sums than the
# reference's (~1e-7 relative); PRCNN_NO_FP_LINEAR=1: interpolate, concatenate, then the layer, as the reference does (A/B).
USE_FP_LINEAR = os.environ.get("PRCNN_NO_FP_LINEAR") is None
# round 5: the proposal layer's box decode inside the fused RPN tail kernel (csrc/rpn_tail.hip rpn_tail_lin_kernel<true>): the (B, N, 76)
# regression tensor never reaches HBM and rpn_decode_kernel leaves the proposal stream; 0: tail writes the rows, the proposal layer
# decodes them (same boxes, bit for bit: A/B)
USE_TAIL_DECODE = os.environ.get("PRCNN_TAIL_DECODE", "1") != "0"


class ZeroArena:
    """ONE zero-filled allocation handed out in slices: the pooled outputs that the packed kernels fill through atomicMax and the
    headers of the packed row lists of a whole chain of launches (a geometry group; the RCNN geometry of a pair of batches) -- one
    fill launch per chain instead of one per tensor (round 5: 5.2 fills per step -> 0.75).  Sized by a dry run of the chain's
    requests: the first pass of a shape allocates per request and records the total, later passes carve one arena."""
    SIZES = {}

    def __init__(self, key, device):
        self.key, self.device = key, device
        self.used = 0
        self.parts = []
        total = ZeroArena.SIZES.get(key)
        self.buf = torch.zeros((total,), dtype=torch.float32, device=device) if total else None

    def take(self, shape, dtype=torch.float32):
        n = 1
        for d in shape:
            n *= int(d)
        n4 = (n + 3) // 4 * 4                                        # 16-byte slices
        if self.buf is None:
            self.used += n4
            part = torch.zeros(tuple(shape), dtype=dtype, device=self.device)
            self.parts.append(part)
            return part
        if self.used + n4 > self.buf.numel():
            # the chain asks for more than its dry run did (an engine switch changed in between): a fill of its own, and a larger
            # arena from the next pass on
            self.used += n4
            part = torch.zeros(tuple(shape), dtype=dtype, device=self.device)
            self.parts.append(part)
            return part
        out = self.buf[self.used:self.used + n].view(dtype).view(tuple(shape))
        self.used += n4
        return out

    def done(self):
        if self.buf is None or self.used > self.buf.numel():
            ZeroArena.SIZES[self.key] = self.used

    def rezero(self):
        """zero everything handed out so far again (a second pass over the same slices)"""
        if self.buf is not None:
            self.buf.zero_()
        for part in self.parts:
            part.zero_()


def pl_ext():
    """the iou3d operator backend in force (looked up per call: the test suite swaps it)"""
    return iou3d_utils.iou3d_cuda


def _round4(c):
    return (c + 3) // 4 * 4


def _round128(c):
    return (c + 127) // 128 * 128


def _pad2(t, rows, cols):
    o = t.new_zeros((rows, cols))
    o[:t.shape[0], :t.shape[1]] = t
    return o.contiguous()


def _pad1(t, n):
    o = t.new_zeros((n,))
    o[:t.shape[0]] = t
    return o


# 128-wide layers (FP level 0, first layers of the RPN heads, per-point parts of SA levels) on the own tiled MFMA layer kernel
# instead of library GEMMs: measured neutral (837-840 vs 844 scenes/s) at these smaller shapes, so it is opt-in
USE_ROWS_GEMM128 = os.environ.get("PRCNN_ROWS_GEMM") is not None


_STRICT = [False]        # set while an engine whose every layer is covered by the hand-written kernels runs a stage (FastPointRCNN._strictly)


def _lib_gemm_allowed():
    """A GEMM library may only appear on the engine's path when an A/B switch asked for it (all-rows mode, PRCNN_LIB_GEMM,
    PRCNN_NO_POINT_MLP, ...) or when the network's shapes are not the ones the hand-written kernels cover (tiny test
    configurations).  With a covered network (default.yaml) and every product switch at its default, a layer that lands here means
    a silently different engine was about to run."""
    return (not _STRICT[0] or not USE_PACKED or not USE_POINT_LAYER or not USE_RCNN_POINT_MLP or not USE_POOL_DEDUP
            or not USE_ROIPOOL_CANONICAL or os.environ.get("PRCNN_ALLOW_LIB_GEMM") == "1")


def gemm_bias_act(a, wt, bias, relu):
    """a (rows, K) @ wt (K, Cout) + bias, optional ReLU.  128-wide layers with K in (128, 256) run on the tiled MFMA layer
    kernel of csrc/rcnn_point_mlp.hip when USE_ROWS_GEMM128 is set; everything else is a
    library GEMM with the epilogue fused when the backend offers it (hipBLASLt through torch._addmm_activation)."""
    if (USE_ROWS_GEMM128 and a.is_cuda and wt.shape[1] == 128 and wt.shape[0] in (128, 256) and a.shape[1] == wt.shape[0]
            and a.shape[0] % 64 == 0 and a.shape[0] >= 4096 and a.stride(1) == 1 and a.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0):
        return pu.pointnet2.rows_gemm128_wrapper(a, wt, bias, relu)
    if a.is_cuda and not _lib_gemm_allowed():
        raise RuntimeError("FastPointRCNN: a %dx%d layer over %d rows would run on a GEMM library although no A/B switch selects one "
                           "(shape not covered by csrc/packed_layer.hip?); set PRCNN_ALLOW_LIB_GEMM=1 to permit it"
                           % (wt.shape[0], wt.shape[1], a.shape[0]))
    if relu:
        try:
            return torch._addmm_activation(bias, a, wt)
        except (AttributeError, RuntimeError):
            return torch.addmm(bias, a, wt).relu_()
    return torch.addmm(bias, a, wt)


SPLIT_BF16 = os.environ.get("PRCNN_SPLIT_BF16") == "1"      # numerics experiment (profiles/r06_split_bf16.md): plain per-point layers as three-way bf16 splits


def point_layer(a, wt, bias, relu, n_out=None):
    """act(a @ wt + bias)[:, :n_out] for a per-point (row-major) matrix: the tiled MFMA layer kernel of
    csrc/packed_layer.hip when K and N are multiples of 128 (fixed summation order, reproduced bit for bit by the oracle),
    else a library GEMM.  n_out < N: the weights of a narrow last layer were zero-padded to N."""
    K, N = wt.shape
    n_out = N if n_out is None else n_out
    if (USE_PACKED and K % 128 == 0 and N % 128 == 0 and a.dim() == 2 and a.shape[1] == K and a.stride(1) == 1
            and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0):
        out = torch.empty((a.shape[0], n_out), dtype=torch.float32, device=a.device)
        if SPLIT_BF16 and getattr(pu.pointnet2, "IS_HIP_EXTENSION", False):
            # EXPERIMENT (numerics switch, default off): the layer on the bf16 matrix cores, operands split in three (csrc/split_bf16.hip)
            return pu.pointnet2.rows_layer_bf16x3_wrapper(a, wt, bias, relu, out)
        return pu.pointnet2.packed_layer_wrapper(a, wt, bias, relu, out)
    y = gemm_bias_act(a, wt, bias, relu)
    return y if n_out == N else y[:, :n_out].contiguous()


class _Mlp:
    """Folded weights of one SharedMLP / Conv1d chain in (K, Cout) form."""

    def __init__(self, layers, grouped_c=None, pad128=False, in_parts=None, pad_out=False):
        """layers: [(w (Cout,Cin), b (Cout), relu)].  grouped_c: if not None the first layer
        consumes a grouped row [features(grouped_c) | pad | xyz | 0] whereas the module's weight
        columns are [xyz(3) | features].  pad128 (per-point chains): every K and N zero-padded to a multiple of 128 -- inputs
        then carry zero columns up to the padded K, ``n_out`` is the real width of the last layer."""
        self.layers = []
        self.n_out = layers[-1][0].shape[0]
        if pad128 and grouped_c is None:
            for i, (w, b, relu) in enumerate(layers):
                np_ = _round128(w.shape[0])
                if i == 0 and in_parts:
                    # the input is a concatenation of tensors that are EACH padded to 128s (FP modules: [interpolated |
                    # skip]): the weight rows of every part go where that part starts, zero rows in between
                    assert sum(in_parts) == w.shape[1]
                    wt = w.new_zeros((sum(_round128(c) for c in in_parts), np_))
                    src = dst = 0
                    for c in in_parts:
                        wt[dst:dst + c, :w.shape[0]] = w[:, src:src + c].t()
                        src, dst = src + c, dst + _round128(c)
                else:
                    wt = _pad2(w.t(), _round128(w.shape[1]), np_)
                self.layers.append((wt.contiguous(), _pad1(b, np_), relu))
            self.split = self.packed = self.wide = self.wide_cat = None
            self.padded = not pad_out          # pad_out: the last layer's output keeps its zero columns (feeds another chain)
            # a 1-wide (<= 4) last layer without ReLU is a GEMV per output: its own small kernel instead of a 128-wide MFMA tile
            self.narrow = None
            w, b, relu = layers[-1]
            if self.padded and w.shape[0] <= 4 and not relu:
                self.narrow = (_pad2(w.t(), _round128(w.shape[1]), w.shape[0]), b.contiguous())
            return
        self.padded = False
        for i, (w, b, relu) in enumerate(layers):
            if i == 0 and grouped_c is not None:
                c, c4 = grouped_c, _round4(grouped_c)
                wt = w.new_zeros((c4 + 4, w.shape[0]))
                if c:
                    wt[:c] = w[:, 3:3 + c].t()
                wt[c4:c4 + 3] = w[:, 0:3].t()
            else:
                k = w.shape[1]
                wt = w.new_zeros((_round4(k), w.shape[0]))
                wt[:k] = w.t()
            self.layers.append((wt.contiguous(), b.contiguous(), relu))

        # split form of the first layer for the "linear before ReLU" shortcut (see _sa_scale)
        self.split = None
        if grouped_c is not None and grouped_c >= 32 and grouped_c % 4 == 0 and self.layers[0][2]:
            wt, b, _ = self.layers[0]
            c4 = _round4(grouped_c)
            self.split = (wt[:grouped_c].contiguous(), wt[c4:c4 + 3].contiguous(), b)   # (C,Cout), (3,Cout), (Cout)
        # 128-wide (zero-padded) form for the fused MFMA kernels: c1, c2 <= 128, c3 in {128, 256}.  Zero columns give
        # relu(0) = 0 activations that meet zero weight rows in the next layer: the padded chain adds exact zeros.
        self.packed = self.wide = self.wide_cat = None
        if self.split is not None and len(self.layers) == 3 and all(l[2] for l in self.layers):
            wf, wx, b1 = self.split
            (w2, b2, _), (w3, b3, _) = self.layers[1], self.layers[2]
            c1, c2, c3 = wf.shape[1], w2.shape[1], w3.shape[1]
            kin = _round128(wf.shape[0]) if PAD128 else wf.shape[0]      # the feature tensor arrives padded to 128s
            if c1 <= 128 and c2 <= 128 and c3 in (128, 256) and w2.shape[0] == c1 and w3.shape[0] == c2:
                self.packed = (_pad2(wf, kin, 128), _pad2(wx, 3, 128), _pad1(b1, 128),
                               _pad2(w2, 128, 128), _pad1(b2, 128), _pad2(w3, 128, c3), b3)
                self.packed_widths = (c1, c2)                          # what the padding hides (the batched kernel skips it)
            elif c3 % 128 == 0 and w2.shape[0] == c1 and w3.shape[0] == c2:
                # wider levels: layer by layer over the packed rows (csrc/packed_layer.hip), widths padded to 128s
                c1p, c2p = _round128(c1), _round128(c2)
                self.wide = (_pad2(wf, kin, c1p), _pad2(wx, 3, c1p), _pad1(b1, c1p),
                             _pad2(w2, c1p, c2p), _pad1(b2, c2p), _pad2(w3, c2p, c3), b3)
                # w1 | w2 | w3 in one allocation: csrc/sa_wide3.hip streams all three through one buffer resource
                self.wide_cat = torch.cat([t.reshape(-1) for t in (self.wide[0], self.wide[3], self.wide[5])]).contiguous()

    def __call__(self, a, start=0):
        last = len(self.layers) - 1
        for i, (wt, b, relu) in enumerate(self.layers[start:], start):
            if USE_POINT_LAYER and i == last and getattr(self, "narrow", None) is not None and a.stride(1) == 1:
                out = torch.empty((a.shape[0], self.n_out), dtype=torch.float32, device=a.device)
                a = pu.pointnet2.rows_dot_wrapper(a, self.narrow[0], self.narrow[1], out)
            elif USE_POINT_LAYER:
                a = point_layer(a, wt, b, relu, self.n_out if (self.padded and i == last) else None)
            else:
                a = gemm_bias_act(a, wt, b, relu)
        return a


def _fold_shared_mlp(mlp):
    layers = fused_mlp.folded_layers(mlp)
    if layers is None:
        raise NotImplementedError("fast path: unsupported SharedMLP structure")
    return [(w, b, True) for w, b in layers]


def _fold_head(seq):
    """nn.Sequential of pt_utils.Conv1d (+Dropout): -> [(w, b, relu)]"""
    out = []
    for block in seq:
        if isinstance(block, torch.nn.Dropout):
            continue
        fb = fused_mlp._fold_block(block)
        if fb is not None:
            out.append((fb[0], fb[1], True))
            continue
        names = [n for n, _ in block.named_children()]
        if names != ["conv"]:
            raise NotImplementedError("fast path: unsupported head block %s" % names)
        conv = block.conv
        w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels)
        b = conv.bias.detach() if conv.bias is not None else w.new_zeros(conv.out_channels)
        out.append((w, b, False))
    return out


def _state_tensors(model):
    return list(model.parameters()) + list(model.buffers())


def _state_version(tensors):
    return sum(t._version for t in tensors)


class FastPointRCNN:
    def __init__(self, model, cfg):
        assert not model.training, "FastPointRCNN is an inference engine: call model.eval() first"
        if cfg.RCNN.ENABLED and cfg.RCNN.USE_INTENSITY:
            raise NotImplementedError("fast path: cfg.RCNN.USE_INTENSITY (reflectance as an RCNN input feature) is not covered")
        self.model, self.cfg = model, cfg
        # per-point input features of the backbone (cfg.RPN.USE_INTENSITY: one reflectance column, rpn.py:17 /
        # pointnet2_msg.py:151-160; lib/config.py:40 turns it on by default, every shipped yaml turns it off).  Round 4: the
        # engine takes them on its GENERAL kernels -- SA level 0 groups [xyz | feature] rows (csrc/pointmajor.hip) and runs its
        # three layers on the layer kernels, the finest FP module gets the features as its skip input (linear-first form) and the
        # heads run layer by layer; the coordinates-only specialisations (csrc/sa_xyz_mlp.hip, the early SA levels on the
        # geometry stream, csrc/rpn_tail.hip) are for the shipped configurations.
        self.in_feat = int(model.rpn.backbone_net.SA_modules[0].mlps[0][0].conv.in_channels) - 3
        self._state = _state_tensors(model)
        self._folded_at = _state_version(self._state)       # BN is folded into the weights HERE: see check_weights()
        rpn = model.rpn
        bb = rpn.backbone_net
        self.sa = []
        for sa in bb.SA_modules:
            scales = []
            for grouper, mlp in zip(sa.groupers, sa.mlps):
                cin = mlp[0].conv.in_channels - 3
                scales.append((grouper.radius, grouper.nsample, _Mlp(_fold_shared_mlp(mlp), grouped_c=cin), cin))
            self.sa.append((sa.npoint, scales))
        # FP module k consumes [features interpolated from level k+1 | skip features of level k]
        sa_w = [sum(sc[2].n_out for sc in scales) for _, scales in self.sa]
        folded = [_fold_shared_mlp(fp.mlp) for fp in bb.FP_modules]
        self.fp = []
        for k, lay in enumerate(folded):
            known = sa_w[-1] if k == len(folded) - 1 else folded[k + 1][-1][0].shape[0]
            skip = self.in_feat if k == 0 else sa_w[k - 1]
            self.fp.append(_Mlp(lay, pad128=PAD128, in_parts=[c for c in (known, skip) if c], pad_out=True))
        self.rpn_cls = _Mlp(_fold_head(rpn.rpn_cls_layer), pad128=PAD128)
        self.rpn_reg = _Mlp(_fold_head(rpn.rpn_reg_layer), pad128=PAD128)
        self.rpn_tail = self._fold_rpn_tail() if (USE_RPN_TAIL and PAD128 and USE_POINT_LAYER) else None
        if cfg.RCNN.ENABLED:
            r = model.rcnn_net
            self.xyz_up = _Mlp(_fold_shared_mlp(r.xyz_up_layer))
            self.merge_down = _Mlp(_fold_shared_mlp(r.merge_down_layer))
            self.rcnn_sa = []
            for sa in r.SA_modules:
                mlp = sa.mlps[0]
                cin = mlp[0].conv.in_channels - 3
                g = sa.groupers[0]
                self.rcnn_sa.append((sa.npoint, getattr(g, "radius", None), getattr(g, "nsample", None),
                                     _Mlp(_fold_shared_mlp(mlp), grouped_c=cin), cin))
            self.rcnn_cls = _Mlp(_fold_head(r.cls_layer), pad128=PAD128)
            self.rcnn_reg = _Mlp(_fold_head(r.reg_layer), pad128=PAD128)
            # both heads read the same 512 features: their first layers side by side are ONE layer of twice the width (a launch less
            # on a chain of six launches over 800 rows; every output column is the same dot product as before)
            self.rcnn_head1 = None
            c0, r0 = self.rcnn_cls.layers[0], self.rcnn_reg.layers[0]
            if (USE_POINT_LAYER and USE_PACKED and len(self.rcnn_cls.layers) > 1 and len(self.rcnn_reg.layers) > 1 and c0[2] and r0[2] and
                    c0[0].shape == r0[0].shape and c0[0].shape[0] % 128 == 0 and c0[0].shape[1] % 128 == 0):
                self.rcnn_head1 = (torch.cat([c0[0], r0[0]], 1).contiguous(), torch.cat([c0[1], r0[1]]).contiguous(), c0[0].shape[1])

    def _covered(self):
        """True when every MLP of this network runs on a kernel of this build with the product switches at their defaults:
        RPN SA scales on the packed / wide / coordinates-only kernels, the fused RPN tail, the RCNN entrance chain and SA levels."""
        try:
            ok = bool(PAD128 and self.rpn_tail is not None)
            for _, scales in self.sa:
                ok = ok and all(sc[2].packed is not None or sc[2].wide is not None or sc[3] == 0 for sc in scales)
            if self.cfg.RCNN.ENABLED:
                ok = ok and self._point_mlp_ok() and all(m[3].packed is not None or m[3].wide is not None for m in self.rcnn_sa)
            return ok
        except Exception:
            return False

    def _strictly(self):
        """context manager: while it is open, a layer of a covered network that falls to a GEMM library raises (gemm_bias_act)"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            if getattr(self, "_is_covered", None) is None:
                self._is_covered = self._covered()
            prev = _STRICT[0]
            _STRICT[0] = bool(self._is_covered)
            try:
                yield
            finally:
                _STRICT[0] = prev
        return cm()

    def _fold_rpn_tail(self):
        """Weights of csrc/rpn_tail.hip (finest FP module + both RPN heads in one kernel) when the network has the shape that
        kernel is written for -- 256 interpolated channels, no skip features, 128-wide layers, a 1-wide score, a regression
        vector of a multiple of 4 (<= 128) channels: default.yaml / double.yaml.  Anything else runs layer by layer."""
        fp0, cl, rg = self.fp[0], self.rpn_cls, self.rpn_reg
        shapes = lambda m: [tuple(l[0].shape) for l in m.layers]
        relus = lambda m: [bool(l[2]) for l in m.layers]
        if (shapes(fp0) != [(256, 128), (128, 128)] or relus(fp0) != [True, True] or
                shapes(cl) != [(128, 128), (128, 128)] or relus(cl) != [True, False] or cl.narrow is None or cl.n_out != 1 or
                shapes(rg) != [(128, 128), (128, 128)] or relus(rg) != [True, False] or rg.n_out % 4 or not 4 <= rg.n_out <= 128):
            return None
        mats = [fp0.layers[0], fp0.layers[1], cl.layers[0], rg.layers[0], rg.layers[1]]
        return {"wcat": torch.cat([m[0] for m in mats], dim=0).contiguous(), "bcat": torch.stack([m[1] for m in mats]).contiguous(),
                "wc2": cl.narrow[0].contiguous().view(-1), "bc2": cl.narrow[1].contiguous(), "n_reg": rg.n_out,
                # linear-first form (USE_FP_LINEAR): layer 1 at the coarse level, the kernel starts at layer 2
                "w1": mats[0][0].contiguous(), "wcat_lin": torch.cat([m[0] for m in mats[1:]], dim=0).contiguous(),
                "zero128": torch.zeros((128,), dtype=torch.float32, device=mats[0][0].device)}

    def check_weights(self):
        """The engine folds BatchNorm into its own copies of the weights at construction.  Loading a checkpoint (or editing
        a parameter in place) afterwards would silently evaluate stale weights: refuse instead."""
        if _state_version(self._state) != self._folded_at:
            raise RuntimeError("FastPointRCNN: the model's parameters changed after the engine was built "
                               "(load the checkpoint first, then construct the engine / PipelinedRunner)")

    # ------------------------------------------------------------------ geometry (xyz only)
    @torch.no_grad()
    def geometry_begin(self, xyz):
        """First (and by far longest) link of the xyz-only chain: FPS + ball queries of SA level 0."""
        state = {"l_xyz": [xyz], "sa": []}
        self._geometry_level(state, 0)
        return state

    @torch.no_grad()
    def geometry_finish(self, state):
        """Remaining SA levels and the three-NN of every FP level -> the geometry dict ``forward`` consumes."""
        for k in range(len(state["sa"]), len(self.sa)):
            self._geometry_level(state, k)
        l_xyz = state["l_xyz"]
        interp = []
        ext = pu.pointnet2
        for k in range(len(self.fp)):      # FP level k: unknown = l_xyz[k], known = l_xyz[k+1]
            unknown, known = l_xyz[k], l_xyz[k + 1]
            if has_entry(ext, "three_nn_weights_wrapper"):
                # neighbours and inverse-distance weights from one kernel (round 4: the weights were sqrt, add, reciprocal, sum and
                # divide launches of torch, five per FP level on the geometry stream)
                B, n, m = unknown.shape[0], unknown.shape[1], known.shape[1]
                idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
                weight = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
                ext.three_nn_weights_wrapper(B, n, m, unknown, known, idx, weight)
            else:
                dist, idx = pu.three_nn(unknown, known)
                dist_recip = 1.0 / (dist + 1e-8)
                weight = (dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)).contiguous()
            interp.append((idx, weight))
        return {"l_xyz": l_xyz, "sa": state["sa"], "fp": interp}

    def _geometry_level(self, state, k):
        npoint, scales = self.sa[k]
        cur = state["l_xyz"][-1]
        ext = pu.pointnet2
        if has_entry(ext, "fps_new_xyz_wrapper") and has_entry(ext, "fps_new_xyz_supported") and ext.fps_new_xyz_supported(cur.shape[1], npoint):
            sel, new_xyz = ext.fps_new_xyz_wrapper(cur, npoint)        # sampling + the centres' coordinates, one launch (round 4: every level)
        else:
            sel = pu.furthest_point_sample(cur, npoint)
            new_xyz = torch.gather(cur, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        if has_entry(ext, "ball_query_full_wrapper"):
            # every slot written by the kernel: no zero fill in front of each query
            B_, N_ = cur.shape[0], cur.shape[1]
            idxs = []
            for radius, ns, _, _ in scales:
                ix = torch.empty((B_, npoint, ns), dtype=torch.int32, device=cur.device)
                ext.ball_query_full_wrapper(B_, N_, npoint, radius, ns, new_xyz, cur, ix)
                idxs.append(ix)
        else:
            idxs = [pu.ball_query(radius, ns, cur, new_xyz) for radius, ns, _, _ in scales]
        lev = {"sel": sel, "new_xyz": new_xyz, "idx": idxs, "pack": [None] * len(scales)}
        if not state.get("defer_packs"):
            self._pack_level(k, cur, lev)
        state["sa"].append(lev)
        state["l_xyz"].append(new_xyz)

    def _pack_level(self, k, cur, lev, arena=None):
        """the distinct-row lists of the scales that run on the packed MFMA kernels (they depend on the indices only);
        arena: a ZeroArena the headers come from (zero already: no memset per list)"""
        # (positional: test proxies around the extension module forward *args only)
        hdr = (lambda: (None, None, None, arena.take((4,), torch.int32))) if (arena is not None and getattr(pu.pointnet2, "IS_HIP_EXTENSION", False)) else (lambda: ())
        lev["pack"] = [pu.pointnet2.ball_pack_wrapper(ix, cur, lev["new_xyz"], *hdr())
                       if (USE_PACKED and (sc[2].packed is not None or sc[2].wide is not None or sc[3] == 0)) else None
                       for ix, sc in zip(lev["idx"], self.sa[k][1])]

    @torch.no_grad()
    def geometry_group(self, xyz_list, on_batch_done=None, group_sa=None):
        """The xyz-only chain for SEVERAL batches in one pass: the serial FPS of a scene occupies one CU for ~6 ms whatever
        the batch, so the chain's latency does not grow with the number of scenes -- its throughput does.  Returns one
        geometry dict per batch (views into the group's tensors; the packed row lists are built per batch)."""
        sizes = [x.shape[0] for x in xyz_list]
        if len(xyz_list) == 1:
            geo1 = [self.geometry(xyz_list[0])]
            if on_batch_done is not None:
                on_batch_done(0)
            return geo1
        # the batches of a group slot of the graphed runner lie side by side in ONE tensor: take the view, not a copy
        x0, whole = xyz_list[0], None
        if (x0.is_contiguous() and all(x.is_contiguous() and x.shape[1:] == x0.shape[1:] and x.dtype == x0.dtype for x in xyz_list)
                and all(xyz_list[i + 1].data_ptr() == xyz_list[i].data_ptr() + xyz_list[i].numel() * 4 for i in range(len(xyz_list) - 1))
                and x0._base is not None and x0._base.is_contiguous() and x0._base.dim() == 3 and x0._base.shape[1:] == x0.shape[1:]):
            lo_ = (x0.data_ptr() - x0._base.data_ptr()) // (x0[0].numel() * 4)
            if 0 <= lo_ and lo_ + sum(sizes) <= x0._base.shape[0] and x0._base[lo_].data_ptr() == x0.data_ptr():
                whole = x0._base[lo_:lo_ + sum(sizes)]
        state = {"l_xyz": [whole if whole is not None else torch.cat(list(xyz_list), dim=0)], "sa": [], "defer_packs": True}
        arena = ZeroArena(("geo", tuple(sizes), tuple(x0.shape[1:]), str(x0.device)), x0.device)
        self._geometry_level(state, 0)
        geo = self.geometry_finish(state)
        groups = self._point_groups(geo["l_xyz"][0])
        # the packed row lists: one list per batch (each batch's kernels walk their own tiles), all batches of the group in ONE
        # launch per (level, scale) when the batches have the same size (4 x 8 launches + 4 x 8 header memsets otherwise)
        ext = pu.pointnet2
        same = len(set(sizes)) == 1 and has_entry(ext, "ball_pack_groups_wrapper")
        gpacks = None
        # GROUP_SA: the early SA levels (see _xyz_level) over ALL clouds of the group in one pass -- one row list per (level, scale)
        # over the group's clouds, one launch per stage instead of one per batch: a quarter of the host's launches for these levels
        # (the host's enqueue time is what limits a step now).  A cloud's rows do not depend on which list holds them: same bits.
        # Every batch then gets views of the group's outputs and needs no row lists of its own for those levels.
        n_early = 0
        if GROUP_SA if group_sa is None else group_sa:
            for k in range(min(EARLY_LEVELS, len(self.sa))):
                self._pack_level(k, geo["l_xyz"][k], geo["sa"][k], arena)
            self._xyz_level(geo, arena)
            while n_early < len(geo["sa"]) and geo["sa"][n_early].get("out") is not None:
                n_early += 1
        arena.done()
        if same and n_early < len(geo["sa"]):
            gpacks = [[ext.ball_pack_groups_wrapper(ix, geo["l_xyz"][k], lev["new_xyz"], sizes[0])
                       if (USE_PACKED and (sc[2].packed is not None or sc[2].wide is not None or sc[3] == 0)) else None
                       for ix, sc in zip(lev["idx"], self.sa[k][1])] for k, lev in enumerate(geo["sa"])]
        out, lo = [], 0
        for bi, b in enumerate(sizes):
            hi = lo + b
            g = {"l_xyz": [t[lo:hi] for t in geo["l_xyz"]], "fp": [(i[lo:hi], w[lo:hi]) for i, w in geo["fp"]], "sa": [],
                 "groups": None if groups is None else (groups[0][lo:hi], groups[1][lo:hi])}
            if geo.get("fp_out"):
                g["fp_out"] = {kk: v[lo:hi] for kk, v in geo["fp_out"].items()}
            if geo.get("tail_G") is not None:
                g["tail_G"] = geo["tail_G"][lo:hi]
            for k, lev in enumerate(geo["sa"]):
                part = {"sel": lev["sel"][lo:hi], "new_xyz": lev["new_xyz"][lo:hi], "idx": [ix[lo:hi] for ix in lev["idx"]],
                        "pack": [None] * len(lev["idx"])}
                if k < n_early:
                    part["out"] = lev["out"][lo:hi]             # computed over the whole group above
                elif gpacks is not None:
                    part["pack"] = [None if p is None else p[bi] for p in gpacks[k]]
                else:
                    self._pack_level(k, g["l_xyz"][k], part)
                g["sa"].append(part)
            if n_early == 0:
                self._xyz_level(g)
            out.append(g)
            if on_batch_done is not None:
                on_batch_done(bi)                               # (the runner records an event here: batch bi can start before the group's last batch is done)
            lo = hi
        return out

    def _point_groups(self, xyz):
        """Spatial groups of the input clouds for the RCNN's RoI pooling (csrc/fps.hip prcnn_point_groups): xyz only, so they
        ride with the geometry chain on its side stream."""
        rp = roipool3d_utils.roipool3d_cuda
        if not (USE_POOL_GROUPS and self.cfg.RCNN.ENABLED and has_entry(rp, "point_groups") and xyz.shape[1] % 64 == 0 and xyz.shape[1] <= 65536):
            return None
        return rp.point_groups(xyz)

    @torch.no_grad()
    def geometry(self, xyz):
        """FPS / ball-query / three-NN of the RPN backbone for xyz (B,N,3)."""
        geo = self.geometry_finish(self.geometry_begin(xyz))
        geo["groups"] = self._point_groups(xyz)
        self._xyz_level(geo)
        return geo

    def _xyz_level(self, geo, arena=None):
        """The first EARLY_LEVELS SA levels of a coordinates-only backbone (USE_INTENSITY False: no input features) depend on xyz
        and the model's weights only -- so does the whole backbone -- and are computed WITH the geometry, on the geometry's stream
        (the pipelined runner: a side stream that has slack, off the feature stream's critical path).  Stored as
        geo["sa"][k]["out"]; `_backbone` starts behind them.  Same kernels, same arguments as in `_backbone`: same bits."""
        if not (USE_XYZ_LEVEL_EARLY and USE_PACKED and USE_XYZ_MLP and self.sa and self.sa[0][1] and
                all(sc[3] == 0 for sc in self.sa[0][1])):
            return
        l_xyz, prev = geo["l_xyz"], None
        B = l_xyz[0].shape[0]
        for k in range(min(EARLY_LEVELS, len(self.sa))):
            npoint, scales = self.sa[k]
            lev = geo["sa"][k]
            packs = lev.get("pack") or [None] * len(scales)
            pre = bool(all(sc[2].packed is not None or sc[2].wide is not None or sc[3] == 0 for sc in scales))
            if not pre or any(pk is None for pk in packs):
                return                                          # a level off the packed kernels: it (and what follows) stays in _backbone
            width = sum(sc[2].layers[-1][0].shape[1] for sc in scales)
            oshape = (B, npoint, _round128(width) if PAD128 else width)
            out = arena.take(oshape) if arena is not None else torch.zeros(oshape, dtype=torch.float32, device=l_xyz[0].device)
            self._sa_level(scales, lev, l_xyz[k], prev, out, True)
            lev["out"] = prev = out
        # ... and, with every SA level done, the coarsest EARLY_FP feature-propagation modules (a few thousand rows each: launches
        # that leave most of the chip idle on the feature stream, and depend on xyz only like everything else here)
        if EARLY_FP > 0 and all(lev.get("out") is not None for lev in geo["sa"]) and len(geo["sa"]) == len(self.sa):
            l_feat = [None] + [lev["out"] for lev in geo["sa"]]
            geo["fp_out"] = {}
            for i in range(-1, -(min(EARLY_FP, len(self.fp) - 1) + 1), -1):      # coarse -> fine, never the finest (fused with the heads)
                kk = len(self.fp) + i
                idx, weight = geo["fp"][kk]
                l_feat[kk] = geo["fp_out"][kk] = self._fp_module(kk, l_feat[kk + 1], l_feat[kk], idx, weight)
            if (EARLY_G0 and EARLY_FP >= len(self.fp) - 1 and self.rpn_tail is not None and self.in_feat == 0
                    and l_feat[1].shape[2] == 256 and USE_FP_LINEAR and has_entry(pu.pointnet2, "rpn_tail_lin_wrapper")):
                kf = l_feat[1]
                geo["tail_G"] = point_layer(kf.view(-1, kf.shape[2]), self.rpn_tail["w1"], self.rpn_tail["zero128"], False).view(kf.shape[0], kf.shape[1], 128)

    # ------------------------------------------------------------------ building blocks
    @staticmethod
    def _sa_scale(xyz, new_xyz, feats, idx, mlp, cin, out, out_col, P_pre=None, pack=None, zeroed=False, dense=False):
        """one (radius, nsample) scale: group -> GEMM chain -> max over nsample into out[..., slice]"""
        ext = pu.pointnet2
        B, N, _ = xyz.shape
        M, ns = idx.shape[1], idx.shape[2]
        if USE_PACKED and mlp.packed is not None:
            # whole scale in ONE hand-written MFMA kernel over the DISTINCT rows of every group (csrc/sa_packed.hip)
            wf, wx, b1, w2, b2, w3, b3 = mlp.packed
            P = P_pre if P_pre is not None else point_layer(feats.view(B * N, feats.shape[2]), wf, b1, False).view(B, N, 128)
            pk = pack if pack is not None else ext.ball_pack_wrapper(idx, xyz, new_xyz)
            ext.sa_packed_mlp_wrapper(new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, out, out_col, zeroed)
            return
        if USE_PACKED and mlp.wide is not None:
            # wider level: the same distinct rows, layer by layer (gather+affine -> MFMA layer -> MFMA layer + segmented max)
            wf, wx, b1, w2, b2, w3, b3 = mlp.wide
            pk = pack if pack is not None else ext.ball_pack_wrapper(idx, xyz, new_xyz)
            if (dense and USE_WIDE_FUSED and USE_WIDE_FUSED3 and has_entry(ext, "sa_wide_fused3_wrapper") and feats.is_contiguous() and
                    feats.shape[2] == wf.shape[0] and ext.sa_wide_fused3_supported(wf.shape[0], wf.shape[1], w2.shape[1], w3.shape[1])):
                # a level that groups every point once (GroupAll): layer 1 runs inside the kernel as well -- no per-point launch, no P
                # through HBM (csrc/sa_wide3.hip); same bits as the two calls below
                ext.sa_wide_fused3_wrapper(new_xyz, xyz, feats, mlp.wide_cat, b1, wx, pk, b2, b3,
                                           (wf.shape[0], wf.shape[1], w2.shape[1], w3.shape[1]), out, out_col, zeroed)
                return
            P = point_layer(feats.view(B * N, feats.shape[2]), wf, b1, False).view(B, N, -1)
            if (dense and USE_WIDE_FUSED and has_entry(ext, "sa_wide_fused_wrapper") and
                    ext.sa_wide_fused_supported(wf.shape[1], w2.shape[1], w3.shape[1])):
                # layers 1-3 + pool in ONE kernel: the packed rows stay in LDS (csrc/sa_wide.hip).  `dense`: the caller knows that
                # (nearly) every row is distinct -- many units of work, where serialising a unit's column blocks costs nothing
                ext.sa_wide_fused_wrapper(new_xyz, xyz, P, wx, pk, w2, b2, w3, b3, out, out_col, zeroed)
                return
            rows = pk.max_tiles * 64
            a1 = torch.empty((rows, wf.shape[1]), dtype=torch.float32, device=xyz.device)
            ext.packed_gather_affine_wrapper(new_xyz, xyz, P, wx, pk, a1)
            y2 = torch.empty((rows, w2.shape[1]), dtype=torch.float32, device=xyz.device)
            ext.packed_layer_wrapper(a1, w2, b2, True, y2, pk)
            ext.packed_layer_segmax_wrapper(y2, w3, b3, pk, B, M, out, out_col, zeroed)
            return
        # the formulations below are for scales the packed kernels do not cover (and for PRCNN_NO_PACK): they read the
        # feature tensor in its true width
        if feats is not None and feats.shape[2] != cin:
            feats = feats[:, :, :cin].contiguous()
        if (mlp.split is not None and len(mlp.layers) == 3 and mlp.layers[1][2] and mlp.layers[2][2] and
                ext.sa_mlp_fused_supported(mlp.split[0].shape[1], mlp.layers[1][0].shape[1],
                                           mlp.layers[2][0].shape[1], ns)):
            # whole scale in ONE hand-written MFMA kernel: gather -> 3 layers -> max, no HBM activations
            wf, wx, b1 = mlp.split
            P = P_pre if P_pre is not None else gemm_bias_act(feats.view(B * N, cin), wf, b1, False).view(B, N, -1)
            ext.sa_mlp_fused_wrapper(new_xyz, xyz, P, wx, idx, mlp.layers[1][0], mlp.layers[1][1],
                                     mlp.layers[2][0], mlp.layers[2][1], out, out_col)
            return
        if (cin == 0 and USE_XYZ_MLP and len(mlp.layers) == 3 and all(l[2] for l in mlp.layers) and
                ext.sa_xyz_mlp_supported(mlp.layers[0][0].shape[1], mlp.layers[1][0].shape[1], mlp.layers[2][0].shape[1], ns)):
            # coordinates-only level (RPN SA1): one VALU kernel, a grouped row never leaves its lane
            (w1, b1, _), (w2, b2, _), (w3, b3, _) = mlp.layers
            if USE_PACKED:
                pk = pack if pack is not None else ext.ball_pack_wrapper(idx, xyz, new_xyz)
                ext.sa_xyz_mlp_packed_wrapper(new_xyz, xyz, pk, w1, b1, w2, b2, w3, b3, out, out_col, zeroed)
            else:
                ext.sa_xyz_mlp_wrapper(new_xyz, xyz, idx, w1, b1, w2, b2, w3, b3, out, out_col)
            return
        if mlp.split is not None and M * ns > N:
            # layer 1 is linear before its ReLU: its feature part is one GEMM over the N points,
            # the grouped rows of the layer-1 OUTPUT are formed by a gather (csrc/pointmajor.hip)
            wf, wx, b1 = mlp.split
            P = torch.addmm(b1, feats.view(B * N, cin), wf).view(B, N, -1)
            y = torch.empty((B, M * ns, wf.shape[1]), dtype=torch.float32, device=xyz.device)
            ext.gather_affine_relu_pm_wrapper(new_xyz, xyz, P, wx, idx, y)
            y = mlp(y.view(B * M * ns, -1), start=1)
        else:
            grouped = torch.empty((B, M * ns, _round4(cin) + 4), dtype=torch.float32, device=xyz.device)
            ext.group_cat_pm_wrapper(B, N, M, cin, ns, new_xyz, xyz, feats, idx, grouped)
            y = mlp(grouped.view(B * M * ns, -1))
        ext.maxpool_pm_wrapper(y, ns, out, out_col)

    @staticmethod
    def _sa_level_wide(xyz, new_xyz, feats, scales, idxs, packs, out, zeroed):
        """All scales of one MSG level whose layers are wider than the register-resident fused kernels take (RPN SA3 / SA4),
        STAGE BY STAGE with the scales side by side in one launch per stage: per-point parts, gather + affine over the packed
        rows, layer 2, layer 3 + segmented max.  On sparse levels every one of these launches is latency-bound (a handful of
        live tiles), so 4 launches per level instead of 4 per scale.  Same kernels, same bits as `_sa_scale`'s wide branch."""
        ext = pu.pointnet2
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        dev = xyz.device
        wides = [sc[2].wide for sc in scales]                  # (wf, wx, b1, w2, b2, w3, b3)
        flat = feats.view(B * N, feats.shape[2])
        Ps = [torch.empty((B * N, w[0].shape[1]), dtype=torch.float32, device=dev) for w in wides]
        ext.packed_layer_batch_wrapper([(flat, w[0], w[2], False, P, None) for w, P in zip(wides, Ps)])
        pks = [pk if pk is not None else ext.ball_pack_wrapper(idx, xyz, new_xyz) for pk, idx in zip(packs, idxs)]
        cols, col = [], 0
        for sc in scales:
            cols.append(col)
            col += sc[2].layers[-1][0].shape[1]
        # (not csrc/sa_wide.hip here: it walks a unit's column blocks one after the other, which is the right trade when there are
        # hundreds of units -- the RCNN's GroupAll level -- and the wrong one for the handful of live tiles of these levels: 1.16 vs
        # 1.13 ms for the RPN stage)
        a1s = [torch.empty((pk.max_tiles * 64, w[0].shape[1]), dtype=torch.float32, device=dev) for pk, w in zip(pks, wides)]
        ext.packed_gather_affine_batch_wrapper([(new_xyz, xyz, P.view(B, N, -1), w[1], pk, a1) for P, w, pk, a1 in zip(Ps, wides, pks, a1s)])
        y2s = [torch.empty((pk.max_tiles * 64, w[3].shape[1]), dtype=torch.float32, device=dev) for pk, w in zip(pks, wides)]
        ext.packed_layer_batch_wrapper([(a1, w[3], w[4], True, y2, pk) for a1, w, y2, pk in zip(a1s, wides, y2s, pks)])
        ext.packed_layer_segmax_batch_wrapper([(y2, w[5], w[6], pk, B, M, out, c, zeroed) for y2, w, pk, c in zip(y2s, wides, pks, cols)])

    def _sa_level(self, scales, lev, cur_xyz, cur_feat, out, pre):
        """all scales of one MSG level into out (pre: out is zeroed, the packed kernels pool through atomicMax)"""
        packs = lev.get("pack") or [None] * len(scales)
        if (USE_SCALE_BATCH and USE_PACKED and 2 <= len(scales) <= 4 and all(sc[2].wide is not None for sc in scales) and
                has_entry(pu.pointnet2, "packed_layer_batch_wrapper")):
            self._sa_level_wide(cur_xyz, lev["new_xyz"], cur_feat, scales, lev["idx"], packs, out, pre)
        elif (USE_SCALE_BATCH and USE_SA2_BATCH and USE_PACKED and len(scales) == 2 and cur_feat is not None and cur_feat.shape[2] % 128 == 0 and
              all(sc[2].packed is not None and sc[2].packed[5].shape[1] == 128 and sc[2].packed[0].shape[0] == cur_feat.shape[2] for sc in scales) and
              all(pk is not None for pk in packs) and has_entry(pu.pointnet2, "sa_packed_mlp_batch_wrapper")
              and has_entry(pu.pointnet2, "packed_layer_batch_wrapper")):
            # both 128-wide scales of the level side by side (round 5): their per-point parts in one layer launch, their fused
            # gather -> layer 2 -> layer 3 -> pool kernels in one launch -- 2 launches for the level instead of 4; same kernels, same bits
            ext = pu.pointnet2
            B, N, _ = cur_xyz.shape
            flat = cur_feat.view(B * N, cur_feat.shape[2])
            if (USE_SA_NARROW and all(sc[2].packed_widths[0] == 64 and sc[2].packed_widths[1] in (64, 96) for sc in scales)
                    and getattr(ext, "IS_HIP_EXTENSION", False)):
                # both scales' first layers are 64 wide under their padding to 128 (RPN SA2): their per-point parts side by side in ONE
                # 128-wide product (a column's chain does not depend on its neighbours: same bits), half the flops and bytes of two
                # padded ones; the narrow kernel reads its 64 columns out of the shared rows
                m0 = scales[0][2]
                if getattr(m0, "_pcat", None) is None or m0._pcat[2] is not scales[1][2].packed[0]:
                    m0._pcat = (torch.cat([sc[2].packed[0][:, :64] for sc in scales], 1).contiguous(),
                                torch.cat([sc[2].packed[2][:64] for sc in scales]).contiguous(), scales[1][2].packed[0])
                Pcat = torch.empty((B * N, 128), dtype=torch.float32, device=cur_xyz.device)
                ext.packed_layer_wrapper(flat, m0._pcat[0], m0._pcat[1], False, Pcat)
                Ps = [Pcat.view(B, N, 128)[:, :, 64 * k:64 * k + 64] for k in range(2)]
            else:
                Ps = [torch.empty((B * N, 128), dtype=torch.float32, device=cur_xyz.device) for _ in scales]
                ext.packed_layer_batch_wrapper([(flat, sc[2].packed[0], sc[2].packed[2], False, P, None) for sc, P in zip(scales, Ps)])
                Ps = [P.view(B, N, 128) for P in Ps]
            probs, col = [], 0
            for (radius, ns, mlp, cin), pack, P in zip(scales, packs, Ps):
                wf, wx, b1, w2, b2, w3, b3 = mlp.packed
                probs.append((lev["new_xyz"], cur_xyz, P, wx, pack, w2, b2, w3, b3, out, col, pre, mlp.packed_widths))
                col += mlp.layers[-1][0].shape[1]
            ext.sa_packed_mlp_batch_wrapper(probs)
        else:
            col = 0
            for (radius, ns, mlp, cin), idx, pack in zip(scales, lev["idx"], packs):
                self._sa_scale(cur_xyz, lev["new_xyz"], cur_feat, idx, mlp, cin, out, col, pack=pack, zeroed=pre)
                col += mlp.layers[-1][0].shape[1]

    def _backbone(self, xyz, geo, fuse_tail=False, feats0=None):
        """-> the (B, N, 128) point features; fuse_tail: -> (features, None), or (None, inputs of the fused last stretch).
        feats0: per-point input features (B, N, C padded to 128) of a backbone built with input channels, else None."""
        l_xyz, l_feat = geo["l_xyz"], [feats0]
        # the packed kernels deliver through atomicMax into zeros: ONE fill for all levels of the backbone (all scales, the padding
        # columns) instead of one per level -- a 5 us launch each on the feature stream
        B = xyz.shape[0]
        shapes, pres = [], []
        for npoint, scales in self.sa:
            width = sum(s[2].layers[-1][0].shape[1] for s in scales)
            shapes.append((B, npoint, _round128(width) if PAD128 else width))   # consumers (next level's per-point part, FP skip) read 128s
            pres.append(bool(USE_PACKED and all(sc[2].packed is not None or sc[2].wide is not None or sc[3] == 0 for sc in scales)))
        # levels that came with the geometry (the coordinates-only level 0, `_xyz_level`)
        early = [lev.get("out") if (lev.get("out") is not None and tuple(lev["out"].shape) == sh) else None for lev, sh in zip(geo["sa"], shapes)]
        sizes = [sh[0] * sh[1] * sh[2] if (pr and e is None) else 0 for sh, pr, e in zip(shapes, pres, early)]
        arena = torch.zeros((sum(sizes),), dtype=torch.float32, device=xyz.device) if sum(sizes) else None
        offs = [sum(sizes[:k]) for k in range(len(sizes))]
        for k, ((npoint, scales), lev) in enumerate(zip(self.sa, geo["sa"])):
            if early[k] is not None:
                l_feat.append(early[k])
                continue
            cur_xyz, cur_feat = l_xyz[len(l_feat) - 1], l_feat[-1]
            width = sum(s[2].layers[-1][0].shape[1] for s in scales)
            wpad = shapes[k][2]
            pre = pres[k]
            out = (arena[offs[k]:offs[k] + sizes[k]].view(shapes[k]) if pre
                   else torch.empty(shapes[k], dtype=torch.float32, device=xyz.device))
            if wpad > width and not pre:
                out[:, :, width:] = 0
            self._sa_level(scales, lev, cur_xyz, cur_feat, out, pre)
            l_feat.append(out)
        ext = pu.pointnet2
        for i in range(-1, -(len(self.fp) + 1), -1):          # coarse -> fine
            k = len(self.fp) + i                               # FP module index == fine level
            if geo.get("fp_out", {}).get(k) is not None:       # came with the geometry (`_xyz_level`)
                l_feat[k] = geo["fp_out"][k]
                continue
            known_feat, skip = l_feat[k + 1], l_feat[k]
            idx, weight = geo["fp"][k]
            if fuse_tail and k == 0 and self.rpn_tail is not None and skip is None and known_feat.shape[2] == 256:
                return None, (known_feat, idx, weight)         # finest level: fused with the heads (rpn_stage)
            l_feat[k] = self._fp_module(k, known_feat, skip, idx, weight)
        return (l_feat[0], None) if fuse_tail else l_feat[0]   # (B, N, 128) point-major (zero-padded to 128s under PAD128)

    def _fp_module(self, k, known_feat, skip, idx, weight):
        """feature-propagation module k: interpolate the coarse features, concatenate the skip features, two layers"""
        ext = pu.pointnet2
        B, n = idx.shape[0], idx.shape[1]
        c2 = known_feat.shape[2]
        c1 = 0 if skip is None else skip.shape[2]
        mlp = self.fp[k]
        wt, b1, relu1 = mlp.layers[0]
        if (USE_FP_LINEAR and USE_POINT_LAYER and PAD128 and c1 and c1 % 128 == 0 and c2 % 128 == 0 and wt.shape[0] == c2 + c1
                and wt.shape[1] % 128 == 0 and len(mlp.layers) >= 2 and has_entry(ext, "packed_layer_interp_wrapper")):
            # G = known_feat @ W[:c2] at the coarse level (no bias, no activation), then layer 1 over the skip features with the
            # interpolated G added in its epilogue
            m = known_feat.shape[1]
            N1 = wt.shape[1]
            if getattr(self, "_zero_bias", None) is None or self._zero_bias.numel() < N1 or self._zero_bias.device != wt.device:
                self._zero_bias = torch.zeros((max(N1, 1024),), dtype=torch.float32, device=wt.device)
            G = point_layer(known_feat.view(B * m, c2), wt[:c2], self._zero_bias[:N1], False).view(B, m, N1)
            y1 = torch.empty((B * n, N1), dtype=torch.float32, device=known_feat.device)
            ext.packed_layer_interp_wrapper(skip.view(B * n, c1), wt[c2:], b1, relu1, y1, G, idx, weight)
            return mlp(y1, start=1).view(B, n, -1)
        buf = torch.empty((B, n, c2 + c1), dtype=torch.float32, device=known_feat.device)
        if c1 and c1 % 4 == 0 and c2 % 4 == 0 and has_entry(ext, "three_interpolate_cat_pm_wrapper"):
            ext.three_interpolate_cat_pm_wrapper(known_feat, idx, weight, skip, buf)      # interpolation + concat, one launch
        else:
            ext.three_interpolate_pm_wrapper(known_feat, idx, weight, buf, 0)
            if c1:
                buf[:, :, c2:] = skip
        return self.fp[k](buf.view(B * n, c2 + c1)).view(B, n, -1)

    # ------------------------------------------------------------------ full forward
    @torch.no_grad()
    def _tail_decode_cfg(self, N):
        """-> the arguments of the fused tail's decode when this configuration's proposal layer can take decoded boxes (the fused
        device-side layer of net/proposal_layer.py over the served regression layout, clouds of N <= 65536 points: ProposalLayer's own
        guard -- a larger cloud keeps its regression rows and goes through ProposalLayer.forward's general path), else None"""
        cfg, pl = self.cfg, self.model.rpn.proposal_layer
        M = cfg[pl.mode].RPN_POST_NMS_TOP_N
        ext3 = pl_ext()
        if not (USE_TAIL_DECODE and USE_FP_LINEAR and self.rpn_tail is not None and getattr(pl, "fused", False) and N <= 65536
                and cfg.TEST.RPN_DISTANCE_BASED_PROPOSE and M <= 128 and cfg.RPN.NMS_TYPE in ("normal", "rotate")
                and has_entry(pu.pointnet2, "rpn_tail_lin_boxes_wrapper") and has_entry(ext3, "rpn_proposals_boxes")
                and pu.pointnet2.rpn_tail_boxes_supported(self.rpn_tail["n_reg"], cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE,
                                                         cfg.RPN.NUM_HEAD_BIN, cfg.RPN.LOC_XZ_FINE)):
            return None
        return (cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE, cfg.RPN.NUM_HEAD_BIN, cfg.RPN.LOC_XZ_FINE, pl._anchor)

    def rpn_stage(self, pts_input, geo=None, want_reg=False):
        """Backbone + RPN heads: everything up to (not including) the proposal layer.  ``want_reg`` = False (the runners): where the
        fused tail can decode the boxes itself the state carries "rpn_boxes" (B, N, 7) and "rpn_reg" is None; True: the regression
        rows are always produced (forward(): the dict of the reference's PointRCNN.forward)."""
        cfg = self.cfg
        self.check_weights()
        if pts_input.shape[-1] != 3 + self.in_feat:
            raise ValueError("pts_input has %d channels, the backbone was built for 3 + %d" % (pts_input.shape[-1], self.in_feat))
        feats0 = None
        if self.in_feat:
            xyz = pts_input[..., 0:3].contiguous()
            # point-major like every feature tensor of the engine, zero-padded to 128 columns (what the FP module's padded
            # weights expect of each part of its input)
            feats0 = pts_input.new_zeros((pts_input.shape[0], pts_input.shape[1], _round128(self.in_feat) if PAD128 else self.in_feat))
            feats0[..., :self.in_feat] = pts_input[..., 3:]
        else:
            xyz = pts_input.contiguous()
        if geo is None:
            geo = self.geometry(xyz)
        B, N, _ = xyz.shape
        with self._strictly():
            feats, tail = self._backbone(xyz, geo, fuse_tail=True, feats0=feats0)
        if tail is not None:
            # interpolation + FP module 0 + both heads: one kernel, a 64-point tile never leaves LDS (csrc/rpn_tail.hip)
            known_feat, idx, weight = tail
            tw = self.rpn_tail
            feats = torch.empty((B, N, 128), dtype=torch.float32, device=xyz.device)
            rpn_cls = torch.empty((B, N, 1), dtype=torch.float32, device=xyz.device)
            dec = None if want_reg or self.in_feat else self._tail_decode_cfg(N)
            rpn_reg = torch.empty((B, N, tw["n_reg"]), dtype=torch.float32, device=xyz.device) if dec is None else None
            rpn_boxes = None
            if USE_FP_LINEAR and has_entry(pu.pointnet2, "rpn_tail_lin_wrapper"):
                # FP layer 1 over the coarse points (a quarter of the rows), interpolated inside the fused kernel
                m = known_feat.shape[1]
                G = geo.get("tail_G")                          # came with the geometry (EARLY_G0)
                if G is None:
                    G = point_layer(known_feat.view(B * m, known_feat.shape[2]), tw["w1"], tw["zero128"], False).view(B, m, 128)
                if dec is not None:
                    # ... and the proposal layer's decode too: the 7-float box leaves the kernel instead of the 76-float row
                    rpn_boxes = torch.empty((B, N, 7), dtype=torch.float32, device=xyz.device)
                    pu.pointnet2.rpn_tail_lin_boxes_wrapper(G, idx, weight, tw["wcat_lin"], tw["bcat"], tw["wc2"], tw["bc2"], tw["n_reg"],
                                                            dec[0], dec[1], dec[2], dec[3], dec[4], xyz, feats, rpn_cls, rpn_boxes)
                else:
                    pu.pointnet2.rpn_tail_lin_wrapper(G, idx, weight, tw["wcat_lin"], tw["bcat"], tw["wc2"], tw["bc2"], feats, rpn_cls, rpn_reg)
            else:
                pu.pointnet2.rpn_tail_wrapper(known_feat, idx, weight, tw["wcat"], tw["bcat"], tw["wc2"], tw["bc2"], feats, rpn_cls, rpn_reg)
        else:
            flat = feats.view(B * N, -1)
            rpn_cls = self.rpn_cls(flat).view(B, N, -1)
            rpn_reg = self.rpn_reg(flat).view(B, N, -1)
            if feats.shape[2] != self.fp[0].n_out:            # narrow configurations: drop the zero padding again
                feats = feats[:, :, :self.fp[0].n_out].contiguous()
            rpn_boxes = None
        out = {"rpn_cls": rpn_cls, "rpn_reg": rpn_reg, "rpn_boxes": rpn_boxes, "backbone_xyz": xyz, "rpn_features": feats, "groups": geo.get("groups")}
        if cfg.RCNN.ENABLED:
            out["rpn_scores_raw"] = rpn_cls[:, :, 0].contiguous()
        return out

    @torch.no_grad()
    def point_aux(self, st):
        """Foreground mask and depth of every point (rcnn input features, point_rcnn.py:44-52).  Half a dozen tiny elementwise
        launches that only the RCNN stage reads: the proposal stage computes them (on ITS stream in the pipelined runner),
        they are off the feature stream's critical path."""
        if "seg_result" not in st:
            ext = pu.pointnet2
            if has_entry(ext, "point_aux_wrapper"):
                # one launch (round 4) instead of sigmoid, compare, cast, norm, divide, subtract
                sc, xyz = st["rpn_scores_raw"], st["backbone_xyz"]
                seg, depth, dn = torch.empty_like(sc), torch.empty_like(sc), torch.empty_like(sc)
                ext.point_aux_wrapper(sc, xyz, float(self.cfg.RPN.SCORE_THRESH), seg, depth, dn)
                st["seg_result"], st["pts_depth"], st["depth_norm"] = seg, depth, dn
                return
            st["seg_result"] = (torch.sigmoid(st["rpn_scores_raw"]) > self.cfg.RPN.SCORE_THRESH).float()
            st["pts_depth"] = torch.norm(st["backbone_xyz"], p=2, dim=2)
            st["depth_norm"] = (st["pts_depth"] / 70.0 - 0.5).contiguous()        # the RCNN input feature (rcnn_net.py:131-137)

    @torch.no_grad()
    def propose(self, st):
        """The proposal layer on the RPN stage's outputs -> (rois, roi_scores_raw); also fills in the per-point RCNN inputs."""
        if self.cfg.RCNN.ENABLED:
            self.point_aux(st)
        if st.get("rpn_boxes") is not None:                  # decoded by the fused tail (rpn_stage): sort, bands, NMS, assembly
            cfg, pl = self.cfg, self.model.rpn.proposal_layer
            boxes, scores = st["rpn_boxes"], st["rpn_scores_raw"]
            M = cfg[pl.mode].RPN_POST_NMS_TOP_N
            rois = torch.empty((boxes.shape[0], M, 7), dtype=torch.float32, device=boxes.device)
            roi_scores = torch.empty((boxes.shape[0], M), dtype=torch.float32, device=boxes.device)
            pl_ext().rpn_proposals_boxes(scores, boxes, cfg[pl.mode].RPN_PRE_NMS_TOP_N, M, cfg[pl.mode].RPN_NMS_THRESH,
                                         cfg.RPN.NMS_TYPE == "rotate", rois, roi_scores)
            return rois, roi_scores
        return self.model.rpn.proposal_layer(st["rpn_scores_raw"], st["rpn_reg"], st["backbone_xyz"])

    @torch.no_grad()
    def rcnn_stage(self, st, rois):
        return self.rcnn_features(self.rcnn_geometry(st, rois))

    @torch.no_grad()
    def rcnn_geometry(self, st, rois):
        """RoI pooling + the sampling / grouping geometry of the RCNN stage (no MLP): may run on the proposal stream."""
        self.point_aux(st)
        return self._rcnn_geometry(st["backbone_xyz"], st["rpn_features"], st["seg_result"], st["pts_depth"], rois,
                                   depth_norm=st["depth_norm"], groups=st.get("groups"))

    @torch.no_grad()
    def rcnn_features(self, rg):
        with self._strictly():
            return self._rcnn_features(rg)

    @torch.no_grad()
    def forward(self, pts_input, geo=None, want_reg=True):
        """pts_input (B,N,3) -> the dict PointRCNN.forward returns in TEST mode (rpn_cls, rpn_reg,
        backbone_xyz, rois, roi_scores_raw, seg_result, rcnn_cls, rcnn_reg); backbone features are
        returned point-major under 'rpn_features' (B,N,C).  ``want_reg`` = False (eval_rcnn.infer_batch: only the detections are
        wanted): rpn_reg may be None -- the boxes were decoded inside the fused tail kernel (rpn_stage)."""
        out = self.rpn_stage(pts_input, geo, want_reg=want_reg)
        if not self.cfg.RCNN.ENABLED:
            return out
        rois, roi_scores_raw = self.propose(out)
        out.update({"rois": rois, "roi_scores_raw": roi_scores_raw})
        out.update(self.rcnn_stage(out, rois))
        return out

    def _point_mlp_ok(self):
        """Shapes csrc/rcnn_point_mlp.hip is written for: xyz_up 5(8) -> 128 -> 128, merge 256 -> 128, SA1 through the
        fused MFMA kernel with a 128 -> 128 per-point part."""
        if getattr(self, "_pm_ok", None) is None:
            ok = len(self.xyz_up.layers) == 2 and len(self.merge_down.layers) == 1 and len(self.rcnn_sa) > 0
            if ok:
                (wu1, _, r1), (wu2, _, r2) = self.xyz_up.layers
                (wm, _, r3), = self.merge_down.layers
                mlp1, ns1 = self.rcnn_sa[0][3], self.rcnn_sa[0][2]
                ok = (tuple(wu1.shape) == (8, 128) and tuple(wu2.shape) == (128, 128) and tuple(wm.shape) == (256, 128)
                      and r1 and r2 and r3 and mlp1.split is not None and tuple(mlp1.split[0].shape) == (128, 128)
                      and len(mlp1.layers) == 3 and self.rcnn_sa[0][0] is not None
                      and pu.pointnet2.sa_mlp_fused_supported(128, mlp1.layers[1][0].shape[1], mlp1.layers[2][0].shape[1], ns1))
            self._pm_ok = bool(ok)
        return self._pm_ok

    def _rcnn(self, xyz, feats, seg_mask, pts_depth, rois, depth_norm=None, groups=None):
        return self._rcnn_features(self._rcnn_geometry(xyz, feats, seg_mask, pts_depth, rois, depth_norm=depth_norm, groups=groups))

    @staticmethod
    def _groupall_fused3_ok(mlp, below):
        """the GroupAll level runs on csrc/sa_wide3.hip (the only consumer of a list whose rows carry their cloud at that level):
        `_sa_scale`'s condition for that branch, known before the level's features exist (they are the level below's output)"""
        ext = pu.pointnet2
        if not (USE_PACKED and mlp.packed is None and mlp.wide is not None and USE_WIDE_FUSED and USE_WIDE_FUSED3
                and has_entry(ext, "sa_wide_fused3_wrapper") and getattr(mlp, "wide_cat", None) is not None):
            return False
        wf, wx, b1, w2, b2, w3, b3 = mlp.wide
        c_below = below.layers[-1][0].shape[1]
        return bool(c_below == wf.shape[0] and ext.sa_wide_fused3_supported(wf.shape[0], wf.shape[1], w2.shape[1], w3.shape[1]))

    def _rcnn_geometry(self, xyz, feats, seg_mask, pts_depth, rois, depth_norm=None, groups=None):
        """Everything of the RCNN stage (rcnn_net.py:127-185) that needs no MLP result: RoI pooling into the canonical row layout,
        then per SA level sampling, ball query and the distinct-row lists.  All of it hangs on the RoIs and on coordinates only
        (the pooled FEATURES are copied, never computed on), so the pipelined runner launches it on the proposal stream right
        behind the proposal layer -- a chain of ten latency-bound launches, 0.3 ms when it sat on the feature stream in front of
        the MFMA kernels (profiles/r02_bench_step_kernel_stats.md).  -> state for `_rcnn_features`."""
        R = self.cfg.RCNN
        if not (R.ROI_SAMPLE_JIT and R.USE_RPN_FEATURES and not R.USE_INTENSITY):
            raise NotImplementedError("fast path covers the default.yaml RCNN input configuration")
        nin = self.model.rcnn_net.rcnn_input_channel                           # xyz + mask + depth = 5
        rp = roipool3d_utils.roipool3d_cuda
        ext = pu.pointnet2
        C = feats.shape[2]
        pooled_cnt = None
        if (USE_ROIPOOL_CANONICAL and has_entry(rp, "forward_canonical") and R.USE_DEPTH and nin == 5 and C % 4 == 0):
            # enlarge + pool + canonical transform + aligned row layout [x',y',z',mask,depth,0,0,0 | feats] in ONE kernel
            B, M = rois.shape[0], rois.shape[1]
            P, W = R.NUM_POINTS, 8 + C
            pooled = torch.empty((B, M, P, W), dtype=torch.float32, device=xyz.device)
            empty = torch.empty((B, M), dtype=torch.int32, device=xyz.device)
            if USE_POOL_DEDUP and USE_PACKED and USE_RCNN_POINT_MLP and P % 64 == 0 and self._point_mlp_ok():
                pooled_cnt = torch.empty((B, M), dtype=torch.int32, device=xyz.device)
            # ... and the rows' coordinates once more as dense clouds (what sampling and ball queries read; was a strided copy)
            xyz_dense = torch.empty((B, M, P, 3), dtype=torch.float32, device=xyz.device)
            rp.forward_canonical(xyz, rois.contiguous(), feats, seg_mask.contiguous(),
                                 depth_norm if depth_norm is not None else (pts_depth / 70.0 - 0.5).contiguous(),
                                 R.POOL_EXTRA_WIDTH, pooled, empty, pooled_cnt, groups, xyz_dense)
            flat = pooled.view(B * M, P, W)
            rows = flat.view(B * M * P, W)
            a = rows[:, 0:8]                                                   # strided view: columns 5..7 are zero
            rpn_part = rows[:, 8:]
        else:
            extra = [seg_mask.unsqueeze(2)]
            if R.USE_DEPTH:
                extra.append((pts_depth / 70.0 - 0.5).unsqueeze(2))
            pts_feature = torch.cat(extra + [feats], dim=2)                   # (B,N,2+128), already point-major
            pooled, _ = roipool3d_utils.roipool3d_gpu(xyz, pts_feature, rois, R.POOL_EXTRA_WIDTH, sampled_pt_num=R.NUM_POINTS)
            B, M, P, W = pooled.shape
            xyz_dense = None
            pooled[:, :, :, 0:3] -= rois[:, :, 0:3].unsqueeze(2)
            flat = pooled.view(B * M, P, W)
            flat[:, :, 0:3] = kitti_utils.rotate_pc_along_y_torch(flat[:, :, 0:3], rois.reshape(-1, 7)[:, 6])
            rows = flat.view(B * M * P, W)
            a = rows.new_zeros((rows.shape[0], _round4(nin)))
            a[:, :nin] = rows[:, :nin]
            rpn_part = rows[:, nin:]
        point_mlp = bool(USE_RCNN_POINT_MLP and W == 136 and rows.shape[0] % 64 == 0 and self._point_mlp_ok())
        use_rows = bool(point_mlp and pooled_cnt is not None and USE_POOLED_ROWS and has_entry(ext, "pooled_rows_wrapper"))
        tiles = ext.pooled_tiles_wrapper(pooled_cnt.view(-1), P) if (point_mlp and pooled_cnt is not None and not use_rows) else None
        cur_xyz = xyz_dense.view(B * M, P, 3) if xyz_dense is not None else flat[:, :, 0:3].contiguous()
        levels = []
        # one zero fill for this stage: the headers of its three row lists and the pooled outputs of its levels (see ZeroArena)
        zarena = ZeroArena(("rcnn", B, M, P, str(rows.device)), rows.device)
        zhdr = (lambda: (zarena.take((4,), torch.int32),)) if getattr(ext, "IS_HIP_EXTENSION", False) else (lambda: ())   # positional (7th) argument
        rowlist = ext.pooled_rows_wrapper(pooled_cnt.view(-1), P, *zhdr()) if use_rows else None       # the distinct pooled rows of all RoIs, back to back
        # representative map of the CURRENT level's points (None: every point counts as distinct): which of them are exact copies
        # of one another.  Level 0: pooled point k >= count is a copy of k % count (`limit`); deeper: the centres the sampling
        # picked from copies of one source are copies of one another -- coordinates, ball and therefore features (dup_rep).
        centre_dedup = bool(USE_CENTRE_DEDUP and pooled_cnt is not None and point_mlp and has_entry(ext, "dup_rep_wrapper"))
        rep = None
        # The two sampled levels' geometry for every RoI cloud in ONE launch (a wave per RoI: csrc/fps.hip rcnn_roi_geometry_kernel)
        # instead of FPS, ball query, representative map -- twice -- as six latency-bound launches
        fused_geo = None
        sa = self.rcnn_sa
        if (USE_ROI_GEOMETRY and centre_dedup and USE_CENTRE_SKIP and len(sa) >= 2 and sa[0][0] is not None and sa[1][0] is not None
                and has_entry(ext, "rcnn_roi_geometry_wrapper") and USE_PACKED
                and all(m_[3].packed is not None or m_[3].wide is not None for m_ in sa[:2])
                and ext.rcnn_roi_geometry_supported(cur_xyz.shape[1], sa[0][0], sa[0][2], sa[1][0], sa[1][2])):
            if USE_ROI_PACKS and has_entry(ext, "rcnn_roi_geometry_packs_wrapper"):
                # ... and both levels' row lists written by the wave that holds the hit lists (no pack launches for these levels)
                # (the index tensors themselves are not written on the HIP path: the packed kernels read the lists, `idx` is asked for its shape;
                #  the lists' rows carry their cloud: no padded last tile per RoI cloud)
                hd = zhdr() + zhdr()
                rc = all(m_[3].packed is not None for m_ in sa[:2])        # the consumers that read lists whose rows carry their cloud
                # ... and, in that form, the list of the GroupAll level above them (one group per RoI: no clouds merged to fill tiles)
                ga = bool(rc and hd and len(sa) == 3 and sa[2][0] is None and self._groupall_fused3_ok(sa[2][3], sa[1][3]))
                # ... and the level-1 centres that are their own representatives as a row list: level 2's per-point layer over those rows only
                cr = bool(rc and hd and USE_CENTRE_ROWS and has_entry(ext, "rows_gemm128_rows_wrapper") and sa[1][3].packed is not None
                          and tuple(sa[1][3].packed[0].shape) == (128, 128))
                extra = ()
                if hd:
                    extra = hd + (False, rc) + ((zhdr()[0], True) if ga else (None, False)) + ((zhdr()[0], True) if cr else ())
                fused_geo = ext.rcnn_roi_geometry_packs_wrapper(cur_xyz, pooled_cnt.view(-1), sa[0][0], sa[0][1], sa[0][2], sa[1][0], sa[1][1],
                                                                sa[1][2], *extra)
            else:
                fused_geo = ext.rcnn_roi_geometry_wrapper(cur_xyz, pooled_cnt.view(-1), sa[0][0], sa[0][1], sa[0][2], sa[1][0], sa[1][1], sa[1][2])
        fused_p3 = next((x for x in (fused_geo or ())[8:] if hasattr(x, "rowinfo")), None)
        fused_crows = next((x for x in (fused_geo or ())[8:] if isinstance(x, tuple)), None)
        for k, (npoint, radius, ns, mlp, cin) in enumerate(self.rcnn_sa):
            lev = {"xyz": cur_xyz, "new_xyz": None, "idx": None, "pack": None}
            if k == 1 and fused_crows is not None:
                lev["crows"] = fused_crows
            if npoint is not None and fused_geo is not None and k < 2:
                new_xyz, idx, rep_out = fused_geo[3 * k:3 * k + 3]
                if len(fused_geo) >= 8:
                    lev["pack"] = fused_geo[6 + k]
                elif k == 0:
                    lev["pack"] = ext.ball_pack_wrapper(idx, cur_xyz, new_xyz, pooled_cnt.view(-1), None, rep_out, *zhdr())
                else:
                    lev["pack"] = ext.ball_pack_wrapper(idx, cur_xyz, new_xyz, None, rep, rep_out, *zhdr())
                rep = rep_out
                lev["new_xyz"], lev["idx"] = new_xyz, idx
                cur_xyz = new_xyz
            elif npoint is not None:
                Bc, n = cur_xyz.shape[0], cur_xyz.shape[1]
                if n <= 1024 and has_entry(ext, "fps_new_xyz_wrapper"):
                    sel, new_xyz = ext.fps_new_xyz_wrapper(cur_xyz, npoint)    # sampling + the centres' coordinates, one launch
                else:
                    sel = pu.furthest_point_sample(cur_xyz, npoint)
                    new_xyz = torch.gather(cur_xyz, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
                dedup = k == 0 and pooled_cnt is not None and point_mlp
                if dedup and has_entry(ext, "ball_query_limit_wrapper"):
                    # pooled rows k >= count are copies of row k % count: scanning the distinct rows finds every ball's points
                    # (the row list below drops the copies anyway); a RoI holds ~60 of its 512 rows at this scene size
                    idx = torch.empty((Bc, npoint, ns), dtype=torch.int32, device=cur_xyz.device)     # every slot is written
                    ext.ball_query_limit_wrapper(Bc, n, npoint, radius, ns, new_xyz, cur_xyz, pooled_cnt.view(-1), idx)
                else:
                    idx = pu.ball_query(radius, ns, cur_xyz, new_xyz)
                # which of THIS level's centres are copies of one another (= the next level's point map): a centre that copies an
                # earlier one gets no rows of its own -- the next level never lists it, so its output is never read
                rep_in = rep
                if centre_dedup and (k == 0 or rep is not None):
                    rep = ext.dup_rep_wrapper(sel, n, pooled_cnt.view(-1) if k == 0 else None, rep_in if k > 0 else None)
                else:
                    rep = None
                crep = rep if USE_CENTRE_SKIP else None
                if dedup:
                    lev["pack"] = ext.ball_pack_wrapper(idx, cur_xyz, new_xyz, pooled_cnt.view(-1), None, crep)   # copies of pooled points are dropped too
                elif USE_PACKED and (mlp.packed is not None or mlp.wide is not None) and has_entry(ext, "ball_pack_wrapper"):
                    lev["pack"] = (ext.ball_pack_wrapper(idx, cur_xyz, new_xyz, None, rep_in, crep) if (rep_in is not None and ns <= 64)
                                   else ext.ball_pack_wrapper(idx, cur_xyz, new_xyz))
                lev["new_xyz"], lev["idx"] = new_xyz, idx
                cur_xyz = new_xyz
            elif fused_p3 is not None:
                # GroupAll over the list the fused launch wrote: every RoI one group (centre 0) of its own 32 centres
                Bc, n = cur_xyz.shape[0], cur_xyz.shape[1]
                cache = self.__dict__.setdefault("_groupall", {})
                key = ("origin", Bc, str(cur_xyz.device))
                if key not in cache:
                    cache[key] = torch.zeros((Bc, 1, 3), dtype=torch.float32, device=cur_xyz.device)
                lev.update({"new_xyz": cache[key], "idx": fused_p3.idx, "pack": fused_p3, "f": 1})
                cur_xyz = None
            elif USE_PACKED and (mlp.packed is not None or mlp.wide is not None):
                # GroupAll (pointnet2_utils.py:267-288): ONE group holding all n points, no centre subtraction == a ball
                # query answer 0..n-1 around the origin; same packed kernels as the other levels
                # f clouds side by side form one "cloud" with f centres (no centre is subtracted, so cloud borders mean nothing
                # here): n = 32 points per RoI would leave every 64-row MFMA tile half full of copies.  The index tensor and the
                # origins are constants of the shape (cached); the row list holds coordinates and is built per batch.
                Bc, n = cur_xyz.shape[0], cur_xyz.shape[1]
                f = 1
                while 2 * f * n <= 64 and Bc % (2 * f) == 0:
                    f *= 2
                key = (Bc, n, f, str(cur_xyz.device))
                # (one entry per shape, never replaced: a captured hipGraph of another batch size may hold these addresses)
                cache = self.__dict__.setdefault("_groupall", {})
                if key not in cache:
                    ga_idx = torch.arange(f * n, dtype=torch.int32, device=cur_xyz.device).view(1, f, n).expand(Bc // f, f, n).contiguous()
                    cache[key] = (key, ga_idx, torch.zeros((Bc // f, f, 3), dtype=torch.float32, device=cur_xyz.device),
                                  (torch.arange(f, dtype=torch.int32, device=cur_xyz.device) * n).view(1, f, 1))
                _, ga_idx, origin, shift = cache[key]
                xyz_v = cur_xyz.view(Bc // f, f * n, 3)
                rep_v = None
                if rep is not None and n <= 64:          # the f clouds' maps side by side, shifted to the merged cloud's numbering
                    rep_v = (rep.view(Bc // f, f, n) + shift).view(Bc // f, f * n)
                lev.update({"xyz": xyz_v, "new_xyz": origin, "idx": ga_idx,
                            "pack": ext.ball_pack_wrapper(ga_idx, xyz_v, origin, None, rep_v, None, *zhdr()) if rep_v is not None else ext.ball_pack_wrapper(ga_idx, xyz_v, origin, None, None, None, *zhdr()),
                            "f": f})
                cur_xyz = None
            levels.append(lev)
        shapes = []
        for (npoint, radius, ns, mlp, cin), lev in zip(self.rcnn_sa, levels):
            cout = mlp.layers[-1][0].shape[1]
            if lev["pack"] is not None and USE_PACKED and (mlp.packed is not None or mlp.wide is not None):
                rows_out = lev["xyz"].shape[0] * (npoint if npoint is not None else lev["f"])
                shapes.append(rows_out * cout)
            else:
                shapes.append(0)
        arena = None
        if sum(shapes):
            parts = [zarena.take((n_,)) if n_ else None for n_ in shapes]
            arena = {"zero": zarena, "parts": parts}
        zarena.done()
        return {"B": B, "M": M, "P": P, "W": W, "rows": rows, "a": a, "rpn_part": rpn_part, "pooled": pooled, "pooled_cnt": pooled_cnt,
                "point_mlp": point_mlp, "tiles": tiles, "rowlist": rowlist, "levels": levels, "arena": arena, "arena_shapes": shapes}

    def _rcnn_features(self, rg):
        """The MLPs of the RCNN stage over the rows and row lists of `_rcnn_geometry`: xyz_up + merge_down + SA levels + heads."""
        B, M, P = rg["B"], rg["M"], rg["P"]
        rows = rg["rows"]
        P_pre = None
        sa1 = self.rcnn_sa[0]
        ext = pu.pointnet2
        if rg["point_mlp"]:
            # xyz_up (2 layers) + concat + merge_down + the per-point part of SA1's layer 1: tiled MFMA layer kernels
            (wu1, bu1, _), (wu2, bu2, _) = self.xyz_up.layers
            (wm, bm, _), = self.merge_down.layers
            wf, _, b1 = sa1[3].split
            P_pre = torch.empty((rows.shape[0], 128), dtype=torch.float32, device=rows.device)
            if rg.get("rowlist") is not None:
                ext.rcnn_point_mlp_rows_wrapper(rows, 8, wu1, bu1, wu2, bu2, wm, bm, wf, b1, P_pre, rg["rowlist"])     # only P, only the distinct rows
            else:
                ext.rcnn_point_mlp_wrapper(rows, 8, wu1, bu1, wu2, bu2, wm, bm, wf, b1, None, None, P_pre, rg["tiles"])   # only P is needed
            P_pre = P_pre.view(B * M, P, 128)
            l_feat = [None]
        else:
            xyz_feature = self.xyz_up(rg["a"])                                 # (rows, 128)
            merged = self.merge_down(torch.cat((xyz_feature, rg["rpn_part"]), dim=1))
            l_feat = [merged.view(B * M, P, -1)]
        # the levels that pool through atomicMax (packed rows) want zeroed outputs: ONE fill for all of them -- made with the geometry
        # (80 MB per batch of 800 RoIs: 16 us that the proposal stream has to spare and the feature stream has not)
        shapes, arena = rg["arena_shapes"], rg["arena"]
        if arena is not None:
            if rg.get("arena_used"):           # a second pass over the same geometry (another set of weights, a probe): the atomicMax pools
                for part in arena["parts"]:    # must not start from the first pass's maxima -- the first pass stays fill-free
                    if part is not None:
                        part.zero_()
            rg["arena_used"] = True
        for k, ((npoint, radius, ns, mlp, cin), lev) in enumerate(zip(self.rcnn_sa, rg["levels"])):
            cur_xyz, cur_feat = lev["xyz"], l_feat[-1]
            cout = mlp.layers[-1][0].shape[1]
            first = len(l_feat) == 1
            pre = shapes[k] > 0
            if npoint is not None:
                Bc = cur_xyz.shape[0]
                out = (arena["parts"][k].view(Bc, npoint, cout) if pre
                       else torch.empty((Bc, npoint, cout), dtype=torch.float32, device=cur_xyz.device))
                P_lev = P_pre if first else None
                if lev.get("crows") is not None and not first and mlp.packed is not None and cur_feat.shape[2] == 128:
                    # the level's per-point part P = f W1 + b1 over the rows its lists name (the centres that are their own representatives)
                    P_lev = torch.empty((Bc * cur_xyz.shape[1], 128), dtype=torch.float32, device=cur_xyz.device)
                    ext.rows_gemm128_rows_wrapper(cur_feat.view(-1, 128), mlp.packed[0], mlp.packed[2], False, P_lev, lev["crows"])
                    P_lev = P_lev.view(Bc, cur_xyz.shape[1], 128)
                self._sa_scale(cur_xyz, lev["new_xyz"], cur_feat, lev["idx"], mlp, cin, out, 0, P_pre=P_lev, pack=lev["pack"],
                               zeroed=pre)
            elif lev["pack"] is not None:                                       # GroupAll over f RoIs per "cloud" (see _rcnn_geometry)
                f = lev["f"]
                Bc = cur_xyz.shape[0] * f
                feat_v = cur_feat.view(Bc // f, cur_xyz.shape[1], cur_feat.shape[2])
                out = (arena["parts"][k].view(Bc, 1, cout) if pre
                       else torch.empty((Bc, 1, cout), dtype=torch.float32, device=cur_xyz.device))
                self._sa_scale(cur_xyz, lev["new_xyz"], feat_v, lev["idx"], mlp, cin, out.view(Bc // f, f, cout), 0, pack=lev["pack"], dense=True,
                               zeroed=pre)
            else:                                                               # GroupAll: one group of n points
                Bc, n = cur_xyz.shape[0], cur_xyz.shape[1]
                c4 = _round4(cin)
                g = cur_feat.new_zeros((Bc, n, c4 + 4))
                g[:, :, :cin] = cur_feat
                g[:, :, c4:c4 + 3] = cur_xyz
                y = mlp(g.view(Bc * n, c4 + 4))
                out = torch.empty((Bc, 1, cout), dtype=torch.float32, device=cur_xyz.device)
                ext.maxpool_pm_wrapper(y, n, out, 0)
            l_feat.append(out)
        top = l_feat[-1].view(l_feat[-1].shape[0], -1)                         # (B*M, 512)
        kp = self.rcnn_cls.layers[0][0].shape[0]
        if top.shape[1] != kp:                                                  # narrow configurations under PAD128
            top = torch.nn.functional.pad(top, (0, kp - top.shape[1]))
        if self.rcnn_head1 is not None and top.stride(1) == 1:
            wt, b, n1 = self.rcnn_head1
            h = point_layer(top, wt, b, True)
            hc, hr = h[:, :n1], h[:, n1:]
            lc, lr = self.rcnn_cls.layers, self.rcnn_reg.layers
            if (USE_POINT_LAYER and USE_PACKED and len(lc) >= 3 and len(lr) >= 3 and has_entry(ext, "packed_layer_batch_wrapper")
                    and all(w_.shape[0] % 128 == 0 and w_.shape[1] % 128 == 0 for w_ in (lc[1][0], lr[1][0]))
                    and hc.shape[1] == lc[1][0].shape[0] and hr.shape[1] == lr[1][0].shape[0] and hc.stride(0) % 4 == 0
                    and hc.data_ptr() % 16 == 0 and hr.data_ptr() % 16 == 0):
                # the second layers of the two branches side by side in ONE launch (800 rows: each is a 20-us launch of a few dozen
                # workgroups on the feature stream); same kernel, same arguments per problem: same bits
                yc = torch.empty((hc.shape[0], lc[1][0].shape[1]), dtype=torch.float32, device=h.device)
                yr = torch.empty((hr.shape[0], lr[1][0].shape[1]), dtype=torch.float32, device=h.device)
                ext.packed_layer_batch_wrapper([(hc, lc[1][0], lc[1][1], lc[1][2], yc, None), (hr, lr[1][0], lr[1][1], lr[1][2], yr, None)])
                return {"rcnn_cls": self.rcnn_cls(yc, start=2), "rcnn_reg": self.rcnn_reg(yr, start=2)}
            return {"rcnn_cls": self.rcnn_cls(hc, start=1), "rcnn_reg": self.rcnn_reg(hr, start=1)}
        return {"rcnn_cls": self.rcnn_cls(top), "rcnn_reg": self.rcnn_reg(top)}

    __call__ = forward
