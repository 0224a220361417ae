"""Inputs of fixture g10 (the `_distill` augmentations and the sweep assembly), shared by the generator (reference functions) and the
tests (this build's radardistill_amd/datasets.py): seeded scenes, seeded nuScenes-style info dicts and the files they point to."""
import numpy as np

SEEDS = (0, 1, 2, 3, 4, 5, 6, 7)
ROT_RANGE = [-0.78539816, 0.78539816]
SCALE_RANGE = [0.9, 1.1]
TRANSLATE_STD = [0.5, 0.5, 0.5]


def scene(seed, n_boxes=7, n_pts=300, n_radar=90):
    g = np.random.default_rng(1000 + seed)
    boxes = np.concatenate([g.uniform(-40, 40, (n_boxes, 2)), g.normal(0, 1, (n_boxes, 1)), g.uniform(0.5, 8, (n_boxes, 3)),
                            g.uniform(-np.pi, np.pi, (n_boxes, 1)), g.normal(0, 3, (n_boxes, 2))], axis=1).astype(np.float32)
    pts = np.concatenate([g.uniform(-50, 50, (n_pts, 2)), g.normal(-1, 1, (n_pts, 1)), g.uniform(0, 255, (n_pts, 1)),
                          g.integers(0, 10, (n_pts, 1)) * 0.05], axis=1).astype(np.float32)
    radar = np.concatenate([g.uniform(-50, 50, (n_radar, 2)), g.normal(0.5, 0.2, (n_radar, 1)), g.uniform(-10, 40, (n_radar, 1)),
                            g.normal(0, 5, (n_radar, 2))], axis=1).astype(np.float32)
    return boxes, pts, radar


def write_sample_files(root, seed=0, n_sweeps=12, radars=("RADAR_FRONT", "RADAR_BACK_LEFT", "RADAR_BACK_RIGHT")):
    """A nuScenes-style sample on disk: key-frame LiDAR file + past sweeps (each with a 4x4 sweep -> key-frame transform and a time lag),
    and per radar 7 sweeps of (x, y, z, rcs, vx_comp, vy_comp) float32 records with sensor -> LiDAR calibration.  -> info dict."""
    import os
    g = np.random.default_rng(2000 + seed)
    os.makedirs(os.path.join(root, "samples"), exist_ok=True)

    def lidar_file(name, n):
        a = np.concatenate([g.uniform(-30, 30, (n, 2)), g.normal(-1, 1, (n, 1)), g.uniform(0, 255, (n, 1)), g.integers(0, 32, (n, 1))], axis=1)
        a[: n // 10, :2] = g.uniform(-0.9, 0.9, (n // 10, 2))          # returns from the ego vehicle (removed from past sweeps only)
        a.astype(np.float32).tofile(os.path.join(root, name))
        return name

    def transform():
        th = g.uniform(-0.2, 0.2)
        m = np.eye(4)
        m[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        m[:3, 3] = g.normal(0, 1.5, 3)
        return m

    info = {"lidar_path": lidar_file("samples/key.bin", 400), "token": "tok%d" % seed, "sweeps": [], "radars": {}}
    for k in range(n_sweeps):
        info["sweeps"].append({"lidar_path": lidar_file("samples/sweep%02d.bin" % k, 200 + 10 * k),
                               "transform_matrix": None if k == 3 else transform(), "time_lag": 0.05 * (k + 1)})
    for name in radars:
        sweeps = []
        for k in range(7):
            n = 20 + k
            a = np.concatenate([g.uniform(-40, 40, (n, 2)), g.normal(0.5, 0.3, (n, 1)), g.uniform(-10, 40, (n, 1)), g.normal(0, 4, (n, 2))], axis=1)
            path = "samples/%s_%d.bin" % (name, k)
            a.astype(np.float32).tofile(os.path.join(root, path))
            th = g.uniform(-np.pi, np.pi)
            rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
            sweeps.append({"data_path": path, "timestamp": 1.5e15 - 7.7e4 * k - g.integers(0, 1000), "sensor2lidar_rotation": rot,
                           "sensor2lidar_translation": g.normal(0, 1.0, 3)})
        info["radars"][name] = sweeps
    info["radars"]["RADAR_SHORT"] = info["radars"][radars[0]][:4]          # fewer sweeps than max_sweeps: all of them are used
    return info
