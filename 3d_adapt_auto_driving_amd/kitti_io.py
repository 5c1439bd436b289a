"""KITTI-format input stage of the eval loop (TEST-mode branch of
pointrcnn/lib/datasets/kitti_rcnn_dataset.py:249-342 ``get_rpn_sample`` with kitti_dataset.py:12-80 readers
and lib/utils/calibration.py:5-105): read velodyne ``.bin`` + ``calib`` text, move points to the rectified
camera frame, keep those that project into the image and lie inside ``PC_AREA_SCOPE``, sample
``RPN.NUM_POINTS`` of them with the reference's near/far policy.  Host-side numpy, as in the reference
(its DataLoader workers); the result is the ``pts_input`` tensor the device path consumes.

Directory layout expected (same as the reference): ``<root>/KITTI/object/{training,testing}/{velodyne,calib,
image_2}/%06d.*`` and ``<root>/KITTI/ImageSets/<split>.txt``.
"""
import os

import numpy as np

from . import synth


class Calibration:
    """P2 (3x4), R0 (3x3), Tr_velo_to_cam (3x4) from a KITTI calib file or a dict with those keys."""

    def __init__(self, calib):
        if isinstance(calib, str):
            calib = self.read_file(calib)
        self.P2 = np.asarray(calib["P2"], dtype=np.float32).reshape(3, 4)
        self.R0 = np.asarray(calib["R0"], dtype=np.float32).reshape(3, 3)
        self.V2C = np.asarray(calib["Tr_velo2cam"], dtype=np.float32).reshape(3, 4)

    @staticmethod
    def read_file(path):
        """Lines 2..5 of the file are P2, P3, R0_rect, Tr_velo_to_cam (calibration.py:5-21)."""
        with open(path) as f:
            lines = f.readlines()

        def row(i):
            return np.array(lines[i].strip().split(" ")[1:], dtype=np.float32)
        return {"P2": row(2), "P3": row(3), "R0": row(4), "Tr_velo2cam": row(5)}

    @staticmethod
    def _hom(pts):
        return np.hstack((pts, np.ones((pts.shape[0], 1), dtype=np.float32)))

    def lidar_to_rect(self, pts_lidar):
        return np.dot(self._hom(pts_lidar), np.dot(self.V2C.T, self.R0.T))

    def rect_to_img(self, pts_rect):
        """-> pixel coordinates (N,2) and depth in the rect frame (N)."""
        hom = self._hom(pts_rect)
        p2d = np.dot(hom, self.P2.T)
        hom[:, 2][hom[:, 2] == 0] = 1e-9
        pts_img = (p2d[:, 0:2].T / hom[:, 2]).T
        depth = p2d[:, 2] - self.P2.T[3, 2]
        return pts_img, depth

    def corners3d_to_img_boxes(self, corners3d):
        n = corners3d.shape[0]
        hom = np.concatenate((corners3d, np.ones((n, 8, 1))), axis=2)
        img = np.matmul(hom, self.P2.T)
        x, y = img[:, :, 0] / img[:, :, 2], img[:, :, 1] / img[:, :, 2]
        boxes = np.stack((np.min(x, axis=1), np.min(y, axis=1), np.max(x, axis=1), np.max(y, axis=1)), axis=1)
        return boxes, np.stack((x, y), axis=2)


def valid_flag(pts_rect, pts_img, depth, img_shape, area_scope=None):
    """In the image, in front of the camera and (optionally) inside PC_AREA_SCOPE
    (kitti_rcnn_dataset.py:201-222)."""
    flag = (pts_img[:, 0] >= 0) & (pts_img[:, 0] < img_shape[1]) & (pts_img[:, 1] >= 0) & (pts_img[:, 1] < img_shape[0])
    flag &= depth >= 0
    if area_scope is not None:
        (x0, x1), (y0, y1), (z0, z1) = area_scope
        flag &= (pts_rect[:, 0] >= x0) & (pts_rect[:, 0] <= x1) & (pts_rect[:, 1] >= y0) & (pts_rect[:, 1] <= y1) \
            & (pts_rect[:, 2] >= z0) & (pts_rect[:, 2] <= z1)
    return flag


class KittiSource:
    """Scene provider over a KITTI tree: ``ids`` from ImageSets/<split>.txt, ``load(id)`` ->
    (pts_input (npoints,3) f32, calib, image_shape)."""

    def __init__(self, root_dir, cfg, split="val", npoints_faraway=4000, seed=1024):
        self.cfg = cfg
        sub = "testing" if split == "test" else "training"
        self.dir = os.path.join(root_dir, "KITTI", "object", sub)
        with open(os.path.join(root_dir, "KITTI", "ImageSets", split + ".txt")) as f:
            self.ids = [int(x.strip()) for x in f if x.strip()]
        self.npoints_faraway = npoints_faraway
        self.seed = seed

    def image_shape(self, idx):
        path = os.path.join(self.dir, "image_2", "%06d.png" % idx)
        if os.path.exists(path):
            from PIL import Image
            w, h = Image.open(path).size
            return (h, w, 3)
        return (375, 1242, 3)          # the usual KITTI size when images are not shipped

    def calib_and_shape(self, idx):
        return Calibration(os.path.join(self.dir, "calib", "%06d.txt" % idx)), self.image_shape(idx)

    def label_lines(self, idx):
        """Ground-truth label_2 lines of a scene (for the AP evaluator)."""
        with open(os.path.join(self.dir, "label_2", "%06d.txt" % idx)) as f:
            return [l for l in f.read().split("\n") if l.strip()]

    def gt_boxes3d(self, idx, train_mode=False):
        """(n,7) f32 [x, y_bottom, z, h, w, l, ry] of the labelled objects of class cfg.CLASSES, for the recall statistics.
        As the reference's ``filtrate_objects`` (kitti_rcnn_dataset.py:155-176): in EVAL mode the ONLY filter is the class
        -- every labelled object of the class counts towards total_gt_bbox, in range or not; the PC_AREA_SCOPE test
        (``check_pc_range`` :186-196, all three axes) applies in TRAIN mode only (``train_mode=True``).  Round 2 dropped
        out-of-range GT boxes in eval as well (and tested x / z only), which inflated the recall figures."""
        classes = (self.cfg.CLASSES,)
        out = []
        for l in self.label_lines(idx):
            f = l.split()
            if f[0] not in classes:
                continue
            h, w, ln, x, y, z, ry = (float(v) for v in f[8:15])
            if train_mode and self.cfg.PC_REDUCE_BY_RANGE:
                (x0, x1), (y0, y1), (z0, z1) = self.cfg.PC_AREA_SCOPE
                if not (x0 <= x <= x1 and y0 <= y <= y1 and z0 <= z <= z1):
                    continue
            out.append([x, y, z, h, w, ln, ry])
        return np.asarray(out, dtype=np.float32).reshape(-1, 7)

    def load_raw(self, idx):
        """Raw velodyne points (n,4) as stored + calib + image shape, for DeviceInputStage (lidar_frame=True)."""
        calib, shape = self.calib_and_shape(idx)
        lidar = np.fromfile(os.path.join(self.dir, "velodyne", "%06d.bin" % idx), dtype=np.float32).reshape(-1, 4)
        return lidar, calib, shape

    raw_in_lidar_frame, raw_needs_image_filter = True, True

    def rect_and_flags(self, idx):
        """-> (raw velodyne (n,4), rectified coordinates (n,3) f32, validity flags (n,) bool, calib, image shape): the part of
        get_rpn_sample in front of the sampler (kitti_rcnn_dataset.py:252-269)."""
        cfg = self.cfg
        calib = Calibration(os.path.join(self.dir, "calib", "%06d.txt" % idx))
        lidar = np.fromfile(os.path.join(self.dir, "velodyne", "%06d.bin" % idx), dtype=np.float32).reshape(-1, 4)
        shape = self.image_shape(idx)
        pts_rect = calib.lidar_to_rect(lidar[:, 0:3])
        pts_img, depth = calib.rect_to_img(pts_rect)
        scope = cfg.PC_AREA_SCOPE if cfg.PC_REDUCE_BY_RANGE else None
        return lidar, pts_rect, valid_flag(pts_rect, pts_img, depth, shape, scope), calib, shape

    def load(self, idx, rng=None):
        """``rng``: the random stream of the sampler.  Default: a LEGACY ``np.random.RandomState(seed + idx)`` per scene -- the
        reference samples from the global legacy stream (np.random.choice / shuffle, seeded once with 1024 at
        tools/eval_rcnn.py:26), so with the same generator type and the same call sequence (synth.subsample_rpn) the chosen
        rows are the reference's, bit for bit, given the same stream state (pinned by tests/golden g11 -- both per-scene
        seeding and one stream consumed in scene order, the single-process loader's behaviour); per-scene seeding keeps the
        result independent of which loader process serves which scene."""
        cfg = self.cfg
        lidar, pts_rect, keep, calib, shape = self.rect_and_flags(idx)
        pts_rect = pts_rect[keep][:, 0:3]
        if cfg.RPN.USE_INTENSITY:
            # pts_input = xyz | intensity - 0.5 (kitti_rcnn_dataset.py:274-275, 321-338); the sampler picks rows, so the column rides along
            pts_rect = np.concatenate([pts_rect, lidar[keep][:, 3:4] - np.float32(0.5)], axis=1)
        pts = synth.subsample_rpn(pts_rect, cfg.RPN.NUM_POINTS, self.npoints_faraway,
                                  rng=rng if rng is not None else np.random.RandomState(self.seed + idx))
        return np.ascontiguousarray(pts, dtype=np.float32), calib, shape


class DeviceInputStage:
    """The same stage on the device (csrc/input_stage.hip): raw points in, ``pts_input`` (B, npoints, 3) out.
    The host only reads files; transform, validity filter and the near/far sampler run as one workgroup per scene.
    ``__call__(raws, calibs, shapes, scene_ids, lidar_frame, image_filter)`` with ``raws`` a list of (n_i, 3|4)
    float32 arrays; returns (pts, stats) device tensors, stats (B,3) = #valid, #near, #far (and the raw index of every
    output point with ``return_choice``)."""

    def __init__(self, cfg, device, npoints_faraway=4000, seed=1024, far_depth=40.0):
        import torch
        if cfg.RPN.USE_INTENSITY:
            raise NotImplementedError("DeviceInputStage produces coordinates only; with cfg.RPN.USE_INTENSITY use the host stage "
                                      "(eval_scenes(device_input=False))")
        self.cfg, self.device = cfg, torch.device(device)
        self.npoints_faraway, self.seed, self.far_depth = npoints_faraway, seed, far_depth

    @staticmethod
    def pack_calib(calib, shape):
        v2c = getattr(calib, "V2C", None)
        r0 = getattr(calib, "R0", None)
        row = np.zeros(35, dtype=np.float32)
        row[0:12] = (np.asarray(v2c, np.float32) if v2c is not None else np.eye(3, 4, dtype=np.float32)).reshape(-1)
        row[12:21] = (np.asarray(r0, np.float32) if r0 is not None else np.eye(3, dtype=np.float32)).reshape(-1)
        row[21:33] = np.asarray(calib.P2, np.float32).reshape(-1)
        row[33], row[34] = shape[0], shape[1]
        return row

    def _staging(self, B, n_max, stride):
        """Persistent PINNED staging buffers (grown on demand, two of them alternating so that the upload of batch i can
        still be in flight while batch i+1 is packed): pinning 23 MB per call cost more than the whole device pass."""
        import torch
        need = B * n_max * stride
        bufs = self.__dict__.setdefault("_pinned", [None, None])
        k = self.__dict__.get("_flip", 0)
        self._flip = k ^ 1
        if bufs[k] is not None and bufs[k][2] is not None:
            bufs[k][2].synchronize()               # the upload that last used this buffer has completed
        raw, small = (None, None) if bufs[k] is None else bufs[k][:2]
        if need and (raw is None or raw.numel() < need):      # the two buffers grow INDEPENDENTLY (a later call with more,
            raw = torch.empty((need,), dtype=torch.float32).pin_memory()     # smaller clouds overran the small one)
        if small is None or small.numel() < B * 40 + 64:
            small = torch.empty((B * 40 + 64,), dtype=torch.float32).pin_memory()
        bufs[k] = (raw, small, None)
        return k, raw, small

    def __call__(self, raws, calibs, shapes, scene_ids, lidar_frame=True, image_filter=True, return_choice=False):
        return self._run(raws, None, calibs, shapes, scene_ids, lidar_frame, image_filter, return_choice)

    def from_packed(self, host, counts, calibs, shapes, scene_ids, lidar_frame=True, image_filter=True, return_choice=False):
        """The same stage over raw clouds that already lie packed in page-locked memory: ``host`` (B, n_max, stride) f32 -- a slot of the
        loaders' shared buffer (eval_rcnn._ShmFeed), rows beyond ``counts[i]`` unused -- is uploaded as it is, no staging copy."""
        return self._run(None, (host, [int(c) for c in counts]), calibs, shapes, scene_ids, lidar_frame, image_filter, return_choice)

    def _run(self, raws, packed, calibs, shapes, scene_ids, lidar_frame, image_filter, return_choice):
        import ctypes
        import torch
        from . import _lib
        cfg = self.cfg
        dev = self.device
        if packed is None:
            B = len(raws)
            stride = raws[0].shape[1]
            n_max = max(1, max(r.shape[0] for r in raws))
            lengths = [r.shape[0] for r in raws]
            k, pin_raw, pin_small = self._staging(B, n_max, stride)
            host = pin_raw[:B * n_max * stride].view(B, n_max, stride)
            hnp = host.numpy()
            for i, r in enumerate(raws):               # rows beyond a cloud's length are never read (counts)
                hnp[i, :r.shape[0]] = r
        else:
            # (no staging copy: only the small argument block needs page-locked memory of this stage's own -- from a RING of 32, so
            #  that waiting for a block's previous upload never throttles the feeding thread: with the two alternating buffers of the
            #  staged form it blocked until the device had reached the call before last, ~1 ms per batch, round 6)
            host, lengths = packed
            B, n_max, stride = host.shape
            ring = self.__dict__.setdefault("_small_ring", [])
            if len(ring) < 32:
                ring.append([torch.empty((B * 40 + 64,), dtype=torch.float32).pin_memory(), None])
                self._ring_at = len(ring) - 1
            else:
                self._ring_at = (self._ring_at + 1) % 32
                if ring[self._ring_at][0].numel() < B * 40 + 64:
                    ring[self._ring_at] = [torch.empty((B * 40 + 64,), dtype=torch.float32).pin_memory(), None]
                if ring[self._ring_at][1] is not None:
                    ring[self._ring_at][1].synchronize()
            k, pin_raw, pin_small = None, None, ring[self._ring_at][0]
        raw = host.to(dev, non_blocking=True)
        # the per-scene calibration rows travel as one pinned block; counts and seeds are two tiny uploads
        # ... in ONE upload (round 6: counts and seeds used to leave from pageable memory, two staged copies that block the feeding
        # thread): [calibration rows B x 35 f32 | counts B i32 | pad to 8 bytes | seeds B i64] as 4-byte words of the pinned block
        so = (B * 36 + 1) & ~1                        # seeds start on an 8-byte boundary
        words = so + 2 * B
        blk = pin_small[:words].numpy()
        blk[:B * 35] = np.stack([self.pack_calib(c, s) for c, s in zip(calibs, shapes)], 0).reshape(-1)
        blk[B * 35:B * 35 + B].view(np.int32)[:] = np.asarray(lengths, dtype=np.int32)
        blk[so:words].view(np.int64)[:] = np.asarray([self.seed + int(i) for i in scene_ids], dtype=np.int64)
        small_dev = pin_small[:words].to(dev, non_blocking=True)
        cal = small_dev[:B * 35].view(B, 35)
        counts = small_dev[B * 35:B * 35 + B].view(torch.int32)
        seeds = small_dev[so:words].view(torch.int64)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        if k is None:
            self._small_ring[self._ring_at][1] = done
        else:
            self._pinned[k] = (pin_raw, pin_small, done)
        self.last_done = done                         # (the uploads of this call have completed behind it: eval_rcnn._ShmFeed recycles the slot)
        npoints = cfg.RPN.NUM_POINTS
        out = torch.empty((B, npoints, 3), dtype=torch.float32, device=dev)
        stats = torch.empty((B, 3), dtype=torch.int32, device=dev)
        choice = torch.empty((B, npoints), dtype=torch.int32, device=dev) if return_choice else None
        scope = None
        if cfg.PC_REDUCE_BY_RANGE:
            (x0, x1), (y0, y1), (z0, z1) = cfg.PC_AREA_SCOPE
            self._scope = (ctypes.c_float * 6)(x0, x1, y0, y1, z0, z1)       # kept alive until the async upload is done
            scope = ctypes.cast(self._scope, ctypes.c_void_p)
        with torch.cuda.device(dev):
            _lib.call("prcnn_input_stage", B, n_max, stride, int(bool(lidar_frame)), int(bool(image_filter)),
                      counts.data_ptr(), raw.data_ptr(), cal.data_ptr(), scope, npoints, float(self.far_depth),
                      int(self.npoints_faraway), seeds.data_ptr(), out.data_ptr(), stats.data_ptr(),
                      _lib.ptr(choice), _lib.current_stream(out))
        return (out, stats, choice) if return_choice else (out, stats)


def device_valid_flags(cfg, device, raws, calibs, shapes, lidar_frame=True, image_filter=True, far_depth=40.0):
    """get_valid_flag + lidar_to_rect on the device for a list of raw clouds (csrc/input_stage.hip ``prcnn_valid_flags``):
    -> (cls (B, n_max) uint8 device tensor: 0 invalid / 1 near / 2 far, rect (B, n_max, 3) f32).  Bitwise the reference's
    numpy results (tests/golden g11)."""
    import ctypes
    import torch
    from . import _lib
    dev = torch.device(device)
    B, stride = len(raws), raws[0].shape[1]
    n_max = max(1, max(r.shape[0] for r in raws))
    host = np.zeros((B, n_max, stride), dtype=np.float32)
    for i, r in enumerate(raws):
        host[i, :r.shape[0]] = r
    raw = torch.from_numpy(host).to(dev)
    cal = torch.from_numpy(np.stack([DeviceInputStage.pack_calib(c, s) for c, s in zip(calibs, shapes)], 0)).to(dev)
    counts = torch.tensor([r.shape[0] for r in raws], dtype=torch.int32, device=dev)
    cls = torch.empty((B, n_max), dtype=torch.uint8, device=dev)
    rect = torch.empty((B, n_max, 3), dtype=torch.float32, device=dev)
    scope = None
    if cfg.PC_REDUCE_BY_RANGE:
        (x0, x1), (y0, y1), (z0, z1) = cfg.PC_AREA_SCOPE
        keep = (ctypes.c_float * 6)(x0, x1, y0, y1, z0, z1)
        scope = ctypes.cast(keep, ctypes.c_void_p)
    with torch.cuda.device(dev):
        _lib.call("prcnn_valid_flags", B, n_max, stride, int(bool(lidar_frame)), int(bool(image_filter)), counts.data_ptr(),
                  raw.data_ptr(), cal.data_ptr(), scope, float(far_depth), rect.data_ptr(), cls.data_ptr(), _lib.current_stream(cls))
        torch.cuda.current_stream(dev).synchronize()          # the scope upload reads a host buffer of this frame
    return cls, rect


class SyntheticSource:
    """Scene provider over the synthetic generator (seed = scene id)."""

    def __init__(self, cfg, num_scenes, raw_points=None):
        self.cfg = cfg
        self.ids = list(range(num_scenes))
        self.raw_points = raw_points
        self.calib = synth.SyntheticCalib()

    def calib_and_shape(self, idx):
        return self.calib, self.calib.image_shape

    def label_lines(self, idx):
        """The generator's car boxes as KITTI label lines (fully visible, not truncated)."""
        from . import kitti_utils
        n = self.raw_points or self.cfg.RPN.NUM_POINTS
        boxes = synth.scene_with_labels(idx, n, 20 if self.raw_points else 10)[1]
        corners = kitti_utils.boxes3d_to_corners3d(boxes.astype(np.float32))
        img, _ = self.calib.corners3d_to_img_boxes(corners)
        h, w = self.calib.image_shape[0], self.calib.image_shape[1]
        lines = []
        for b, ib in zip(boxes, img):
            x1, y1, x2, y2 = np.clip(ib, [0, 0, 0, 0], [w - 1, h - 1, w - 1, h - 1])
            beta = np.arctan2(b[2], b[0])
            alpha = -np.sign(beta) * np.pi / 2 + beta + b[6]
            lines.append("Car 0.00 0 %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" %
                         (alpha, x1, y1, x2, y2, b[3], b[4], b[5], b[0], b[1], b[2], b[6]))
        return lines

    def gt_boxes3d(self, idx):
        """(n,7) f32 [x, y_bottom, z, h, w, l, ry]: the generator's car boxes (for the recall statistics)."""
        n = self.raw_points or self.cfg.RPN.NUM_POINTS
        return synth.scene_with_labels(idx, n, 20 if self.raw_points else 10)[1].astype(np.float32)

    def load_raw(self, idx):
        """The generated cloud before sampling (rect frame), for DeviceInputStage (lidar_frame=False, no image filter)."""
        n = self.raw_points or self.cfg.RPN.NUM_POINTS
        pts = synth.dense_scene(idx, n) if self.raw_points else synth.scene(idx, n)
        return pts, self.calib, self.calib.image_shape

    raw_in_lidar_frame, raw_needs_image_filter = False, False

    def load(self, idx):
        n = self.cfg.RPN.NUM_POINTS
        if self.raw_points:
            pts = synth.subsample_rpn(synth.dense_scene(idx, self.raw_points), n, rng=np.random.default_rng(1024 + idx))
        else:
            pts = synth.scene(idx, n)
        if self.cfg.RPN.USE_INTENSITY:                      # a synthetic reflectance column, already shifted to [-0.5, 0.5)
            refl = np.random.default_rng(77000 + idx).random((len(pts), 1)).astype(np.float32) - np.float32(0.5)
            pts = np.concatenate([pts, refl], axis=1)
        return pts, self.calib, self.calib.image_shape
