"""Synthetic KITTI-shaped scenes (there is no dataset on the box).

A scene is N points in rect-camera coordinates: uniform clutter inside PC_AREA_SCOPE
(x in [-40,40], y in [-1,3], z in [0,70.4]), a noisy ground plane near y = 1.6, and a few
car-sized dense boxes standing on it, so that proposals and RoI pooling see non-empty boxes.
Deterministic in the seed (seed = scene id).  Also a KITTI-like synthetic calibration used by the
result writer."""
import numpy as np


def scene_with_labels(seed, n=16384, n_cars=10):
    """-> (pts (n,3) f32, car boxes (n_cars,7) f64 = [x, y_bottom, z, h, w, l, ry] in the camera frame)."""
    rng = np.random.default_rng(seed)
    per_car = max(1, min(200, n // (4 * n_cars)))
    n_car_pts = per_car * n_cars
    n_ground = (n - n_car_pts) // 2
    n_bg = n - n_car_pts - n_ground
    bg = rng.uniform([-40, -1, 0], [40, 3, 70.4], (n_bg, 3))
    ground = np.stack([rng.uniform(-40, 40, n_ground), 1.6 + 0.05 * rng.standard_normal(n_ground),
                       rng.uniform(0, 70.4, n_ground)], 1)
    cars, boxes = [], []
    for _ in range(n_cars):
        c = np.array([rng.uniform(-20, 20), 0.8, rng.uniform(5, 60)])
        ry = rng.uniform(-np.pi, np.pi)
        loc = rng.uniform([-1.95, -0.75, -0.8], [1.95, 0.75, 0.8], (per_car, 3))  # l, h, w
        x = loc[:, 0] * np.cos(ry) + loc[:, 2] * np.sin(ry)
        z = -loc[:, 0] * np.sin(ry) + loc[:, 2] * np.cos(ry)
        cars.append(np.stack([x, loc[:, 1], z], 1) + c)
        boxes.append([c[0], c[1] + 0.75, c[2], 1.5, 1.6, 3.9, ry])
    pts = np.concatenate([bg, ground] + cars, 0).astype(np.float32)
    rng.shuffle(pts)
    return pts, np.array(boxes, dtype=np.float64).reshape(-1, 7)


def scene(seed, n=16384, n_cars=10):
    return scene_with_labels(seed, n, n_cars)[0]


def scenes(b, n=16384, seed0=0):
    return np.stack([scene(seed0 + i, n) for i in range(b)], 0)


def dense_scene(seed, n_raw=180000):
    """A denser (Waymo-like) raw cloud for the cross-domain config: same generator, more points."""
    return scene(seed, n_raw, n_cars=20)


def subsample_rpn(pts, npoints=16384, npoints_faraway=4000, rng=None):
    """The reference's host-side 16384-point sampler restated (kitti_rcnn_dataset.py:288-324):
    keep every point beyond 40 m depth (at most npoints_faraway of them), fill the rest from the
    near points without replacement; if the scene has fewer than npoints, pad by re-sampling;
    shuffle."""
    rng = rng or np.random.default_rng(0)
    if len(pts) > npoints:
        near_flag = pts[:, 2] < 40.0
        far = np.where(~near_flag)[0]
        near = np.where(near_flag)[0]
        if len(far) > npoints_faraway:
            far = rng.choice(far, npoints_faraway, replace=False)
        need = npoints - len(far)
        near = rng.choice(near, need, replace=len(near) < need)   # with replacement only if short
        choice = np.concatenate((near, far), axis=0) if len(far) > 0 else near
        rng.shuffle(choice)
    else:
        choice = np.arange(0, len(pts), dtype=np.int32)
        if npoints > len(pts):
            need = npoints - len(pts)
            extra = rng.choice(choice, need, replace=len(choice) < need)
            choice = np.concatenate((choice, extra), axis=0)
        rng.shuffle(choice)
    return pts[choice]


class SyntheticCalib:
    """KITTI-like P2 (f = 707.05, cu = 604, cv = 180), 375 x 1242 image; rect == camera frame.
    corners3d_to_img_boxes follows pointrcnn/lib/utils/calibration.py:107-125."""
    image_shape = (375, 1242)

    def __init__(self):
        self.P2 = np.array([[707.05, 0., 604., 45.], [0., 707.05, 180., 0.2], [0., 0., 1., 0.003]], dtype=np.float32)

    def corners3d_to_img_boxes(self, corners3d):
        n = corners3d.shape[0]
        hom = np.concatenate((corners3d, np.ones((n, 8, 1))), axis=2)       # (N,8,4)
        img = np.matmul(hom, self.P2.T)                                      # (N,8,3)
        x, y = img[:, :, 0] / img[:, :, 2], img[:, :, 1] / img[:, :, 2]
        x1, y1, x2, y2 = np.min(x, axis=1), np.min(y, axis=1), np.max(x, axis=1), np.max(y, axis=1)
        boxes = np.stack((x1, y1, x2, y2), axis=1)
        boxes_corner = np.stack((x, y), axis=2)
        return boxes, boxes_corner
