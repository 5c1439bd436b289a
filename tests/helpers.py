"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import numpy as np


def scene(seed, n=16384, n_cars=10):
    """KITTI-shaped synthetic cloud: uniform background in PC_AREA_SCOPE, a ground
    plane near y=1.6 and a few car-sized dense boxes."""
    rng = np.random.default_rng(seed)
    per_car = max(1, min(200, n // (4 * n_cars)))
    n_car_pts = per_car * n_cars
    n_ground = (n - n_car_pts) // 2
    n_bg = n - n_car_pts - n_ground
    bg = rng.uniform([-40, -1, 0], [40, 3, 70.4], (n_bg, 3))
    ground = np.stack([rng.uniform(-40, 40, n_ground), 1.6 + 0.05 * rng.standard_normal(n_ground),
                       rng.uniform(0, 70.4, n_ground)], 1)
    cars = []
    for _ in range(n_cars):
        c = np.array([rng.uniform(-20, 20), 0.8, rng.uniform(5, 60)])
        ry = rng.uniform(-np.pi, np.pi)
        loc = rng.uniform([-1.95, -0.75, -0.8], [1.95, 0.75, 0.8], (per_car, 3))  # l, h, w
        x = loc[:, 0] * np.cos(ry) + loc[:, 2] * np.sin(ry)
        z = -loc[:, 0] * np.sin(ry) + loc[:, 2] * np.cos(ry)
        cars.append(np.stack([x, loc[:, 1], z], 1) + c)
    pts = np.concatenate([bg, ground] + cars, 0).astype(np.float32)
    rng.shuffle(pts)
    return pts


def scenes(b, n=16384, seed0=0):
    return np.stack([scene(seed0 + i, n) for i in range(b)], 0)


def bev_boxes(rng, n, spread=20.0, rotated=True):
    """(n,5) [x1,y1,x2,y2,ry] boxes with plenty of overlaps."""
    cx, cy = rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n)
    l, w = rng.uniform(3.0, 5.0, n), rng.uniform(1.4, 2.2, n)
    ry = rng.uniform(-np.pi, np.pi, n) if rotated else np.zeros(n)
    return np.stack([cx - l / 2, cy - w / 2, cx + l / 2, cy + w / 2, ry], 1).astype(np.float32)


def boxes3d(rng, n, xz_scope=((-20, 20), (5, 60))):
    x = rng.uniform(*xz_scope[0], n); z = rng.uniform(*xz_scope[1], n)
    y = rng.uniform(1.2, 2.0, n)
    h = rng.uniform(1.3, 1.8, n); w = rng.uniform(1.4, 1.9, n); l = rng.uniform(3.2, 4.6, n)
    ry = rng.uniform(-np.pi, np.pi, n)
    return np.stack([x, y, z, h, w, l, ry], 1).astype(np.float32)
