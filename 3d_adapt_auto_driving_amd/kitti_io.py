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

    def load(self, idx):
        cfg = self.cfg
        calib = Calibration(os.path.join(self.dir, "calib", "%06d.txt" % idx))
        lidar = np.fromfile(os.path.join(self.dir, "velodyne", "%06d.bin" % idx), dtype=np.float32).reshape(-1, 4)
        shape = self.image_shape(idx)
        pts_rect = calib.lidar_to_rect(lidar[:, 0:3])
        pts_img, depth = calib.rect_to_img(pts_rect)
        scope = cfg.PC_AREA_SCOPE if cfg.PC_REDUCE_BY_RANGE else None
        pts_rect = pts_rect[valid_flag(pts_rect, pts_img, depth, shape, scope)][:, 0:3]
        pts = synth.subsample_rpn(pts_rect, cfg.RPN.NUM_POINTS, self.npoints_faraway,
                                  rng=np.random.default_rng(self.seed + idx))
        return np.ascontiguousarray(pts, dtype=np.float32), calib, shape


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

    def load(self, idx):
        n = self.cfg.RPN.NUM_POINTS
        if self.raw_points:
            pts = synth.subsample_rpn(synth.dense_scene(idx, self.raw_points), n, rng=np.random.default_rng(1024 + idx))
        else:
            pts = synth.scene(idx, n)
        return pts, self.calib, self.calib.image_shape
