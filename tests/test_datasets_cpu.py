"""Host-side data front end (radardistill_amd/datasets.py) against fixture g10 -- produced by the reference's own
augmentor_utils / data_augmentor / nuscenes_dataset_distill functions under seeded numpy generators (tests/golden/make_golden.py::g10)
-- plus defining properties of the pieces whose reference implementation is a compiled extension that is not built here
(roiaware `points_in_boxes_cpu`, iou3d `boxes_bev_iou_cpu`: parity unpinned against the binaries)."""
import os

import numpy as np
import pytest

from radardistill_amd import datasets as DS
from tests.golden import augment_case as AC


def test_world_transforms_bit_exact_vs_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_augment.npz"))
    flips = set()
    for seed in AC.SEEDS:
        b, p, r = AC.scene(seed)
        np.random.seed(seed)          # the same global generator, the same calls in the same order as the reference
        b, p, r, fx = DS.flip_world(b, p, r, "x")
        b, p, r, fy = DS.flip_world(b, p, r, "y")
        b, p, r, rot = DS.rotate_world(b, p, r, AC.ROT_RANGE)
        b, p, r, sc = DS.scale_world(b, p, r, AC.SCALE_RANGE)
        b, p, r, tr = DS.translate_world(b, p, r, AC.TRANSLATE_STD)
        assert bool(fx) == bool(g[f"flip_x_{seed}"]) and bool(fy) == bool(g[f"flip_y_{seed}"])
        flips.add((bool(fx), bool(fy)))
        assert rot == float(g[f"rot_{seed}"]) and sc == float(g[f"scale_{seed}"]) and np.array_equal(tr, g[f"translate_{seed}"])
        assert p.dtype == np.float32 and np.array_equal(p, g[f"points_{seed}"]) and np.array_equal(r, g[f"radar_{seed}"])
    assert len(flips) >= 3          # the seeds cover several flip combinations
    b, p, r = AC.scene(0)
    assert DS.scale_world(b, p, r, [1.0, 1.0005])[3] is None          # degenerate range: nothing drawn, nothing scaled


def test_whole_augmentor_queue_and_tail_vs_reference_fixture(golden_dir):
    """One DataAugmentorDistill.forward over the four world transforms + the tail (heading wrapped to [-pi, pi), class mask): boxes
    bit-exact against the reference's functions run in the same order."""
    g = np.load(os.path.join(golden_dir, "g10_augment.npz"))
    cfg = {"DISABLE_AUG_LIST": ["placeholder", "gt_sampling_distill"],
           "AUG_CONFIG_LIST": [{"NAME": "gt_sampling_distill"},
                               {"NAME": "random_world_flip_distill", "ALONG_AXIS_LIST": ["x", "y"]},
                               {"NAME": "random_world_rotation_distill", "WORLD_ROT_ANGLE": AC.ROT_RANGE},
                               {"NAME": "random_world_scaling_distill", "WORLD_SCALE_RANGE": AC.SCALE_RANGE},
                               {"NAME": "random_world_translation_distill", "NOISE_TRANSLATE_STD": AC.TRANSLATE_STD}]}
    for seed in AC.SEEDS:
        b, p, r = AC.scene(seed)
        np.random.seed(seed)
        aug = DS.DataAugmentorDistill(cfg, ["car", "bus", "truck"])
        assert len(aug.queue) == 4          # the disabled sampler is skipped (DisableAugmentationHook's mechanism)
        # run the transforms, then add the fixture's heading offset and the class mask before the tail -- as the generator did
        tail_free = DS.DataAugmentorDistill(cfg, ["car"])
        d = {"gt_boxes": b, "points": p, "radar_points": r}
        heading_in = None
        for name, c in aug.queue:
            bb, pp, rr = d["gt_boxes"], d["points"], d["radar_points"]
            if name == "random_world_flip_distill":
                for axis in c["ALONG_AXIS_LIST"]:
                    bb, pp, rr, _ = DS.flip_world(bb, pp, rr, axis)
            elif name == "random_world_rotation_distill":
                bb, pp, rr, _ = DS.rotate_world(bb, pp, rr, c["WORLD_ROT_ANGLE"])
            elif name == "random_world_scaling_distill":
                bb, pp, rr, _ = DS.scale_world(bb, pp, rr, c["WORLD_SCALE_RANGE"])
            else:
                bb, pp, rr, _ = DS.translate_world(bb, pp, rr, c["NOISE_TRANSLATE_STD"])
            d["gt_boxes"], d["points"], d["radar_points"] = bb, pp, rr
        d["gt_boxes"][:, 6] += 3.0 * (seed - 3)
        d["gt_names"] = np.array(["car", "bus", "truck"] * 3)[:len(b)]
        d["gt_boxes_mask"] = np.arange(len(b)) % 3 != 1
        tail_free.queue = []
        d = tail_free.forward(d)
        assert np.array_equal(d["gt_boxes"], g[f"boxes_{seed}"]) and len(d["gt_names"]) == len(d["gt_boxes"])
        assert np.all(d["gt_boxes"][:, 6] >= -np.pi - 1e-6) and np.all(d["gt_boxes"][:, 6] < np.pi + 1e-6)
        assert np.array_equal(d["points"], g[f"points_{seed}"])
    with pytest.raises(NotImplementedError):
        DS.DataAugmentorDistill([{"NAME": "random_image_flip"}], ["car"])
    with pytest.raises(ValueError):
        DS.DataAugmentorDistill([{"NAME": "gt_sampling_distill", "SAMPLE_GROUPS": []}], ["car"])


def test_sweep_assembly_bit_exact_vs_reference_fixture(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "g10_augment.npz"))
    root = tmp_path / "data"
    info = AC.write_sample_files(str(root), seed=0)
    np.random.seed(5)
    lidar = DS.lidar_with_sweeps(root, info, max_sweeps=10)
    assert lidar.dtype == g["lidar_sweeps"].dtype and np.array_equal(lidar, g["lidar_sweeps"])
    radar = DS.radar_with_sweeps(info, lambda path: np.fromfile(str(root / path), dtype=np.float32), max_sweeps=6)
    assert np.array_equal(radar, g["radar_sweeps"])
    pts, lag = DS.lidar_sweep(root, info["sweeps"][3])          # the sweep without a transform
    assert np.array_equal(pts, g["sweep3_points"]) and np.array_equal(lag, g["sweep3_times"])
    assert not np.any((np.abs(pts[:, 0]) < 1.0) & (np.abs(pts[:, 1]) < 1.0))          # ego returns removed


def test_points_in_boxes_and_bev_overlap_known_answers():
    box = np.array([[1.0, 2.0, 0.5, 4.0, 2.0, 1.0, np.pi / 2]], dtype=np.float32)          # long axis along y after the quarter turn
    pts = np.array([[1.0, 2.0, 0.5], [1.0, 3.9, 0.5], [1.9, 2.0, 0.5], [2.1, 2.0, 0.5], [1.0, 2.0, 1.01], [1.0, 4.005, 0.5], [1.0, 4.02, 0.5]], dtype=np.float32)
    assert DS.points_in_boxes(pts, box)[0].tolist() == [1, 1, 1, 0, 0, 1, 0]          # footprint margin 1e-2, none in z
    a = np.array([0, 0, 0, 4, 2, 1, 0.0]); b = np.array([1, 0, 0, 4, 2, 1, 0.0])
    assert abs(DS.bev_overlap_area(a, b) - 6.0) < 1e-12
    assert abs(DS.bev_overlap_area(a, np.array([0, 0, 0, 2, 4, 1, np.pi / 2])) - 8.0) < 1e-9          # the same rectangle, turned
    assert abs(DS.bev_overlap_area(a, np.array([0, 0, 0, 2, 2, 1, np.pi / 4])) - 4.0 * (1 - (np.sqrt(2) - 1) ** 2 * 0) ) <= 4.0          # contained or clipped: bounded by its area
    assert DS.bev_overlap_area(a, np.array([10, 0, 0, 4, 2, 1, 0.3])) == 0.0
    iou = DS.bev_iou_matrix(np.stack([a, b]).astype(np.float32), np.stack([a, np.array([10, 0, 0, 1, 1, 1, 0.0])]).astype(np.float32))
    assert abs(iou[0, 0] - 1.0) < 1e-6 and abs(iou[1, 0] - 6.0 / 10.0) < 1e-6 and iou[0, 1] == 0 and iou[1, 1] == 0


def _db(n_per_class, rng):
    db = {}
    for ci, name in enumerate(["car", "bus"]):
        infos = []
        for k in range(n_per_class):
            box = np.array([rng.uniform(-40, 40), rng.uniform(-40, 40), 0.0, 4.0, 2.0, 1.5, rng.uniform(-3, 3), 0.0, 0.0], dtype=np.float32)
            infos.append({"name": name, "box3d_lidar": box, "num_points_in_gt": 4 + k % 5, "num_radar_points_in_gt": k % 3,
                          "points": rng.normal(0, 0.5, (6, 5)).astype(np.float32), "radar_points": rng.normal(0, 0.5, (2, 6)).astype(np.float32)})
        db[name] = infos
    return db


def test_gt_sampler_quota_rejection_and_point_removal():
    rng = np.random.default_rng(3)
    db = _db(40, rng)
    cfg = {"PREPARE": {"filter_by_min_points": ["car:5", "bus:5"]}, "SAMPLE_GROUPS": ["car:4", "bus:3", "tram:9"], "NUM_POINT_FEATURES": 5,
           "REMOVE_EXTRA_WIDTH": [0.0, 0.0, 0.0], "LIMIT_WHOLE_SCENE": True}
    s = DS.GtSamplerDistill(db, cfg, ["car", "bus"])
    assert all(i["num_points_in_gt"] >= 5 and i["num_radar_points_in_gt"] >= 1 for infos in s.db.values() for i in infos)
    assert set(s.groups) == {"car", "bus"}          # classes outside class_names are ignored
    gt = np.array([[0, 0, 0, 4, 2, 1.5, 0, 0, 0], [5, 5, 0, 4, 2, 1.5, 1, 0, 0]], dtype=np.float32)
    pts = np.concatenate([rng.uniform(-45, 45, (4000, 2)), np.zeros((4000, 1)), rng.uniform(0, 1, (4000, 2))], axis=1).astype(np.float32)
    rad = np.concatenate([rng.uniform(-45, 45, (500, 2)), np.zeros((500, 1)), rng.uniform(0, 1, (500, 3))], axis=1).astype(np.float32)
    np.random.seed(1)
    d = s({"gt_boxes": gt.copy(), "gt_names": np.array(["car", "ignored"]), "gt_boxes_mask": np.array([True, False]), "points": pts, "radar_points": rad})
    assert "gt_boxes_mask" not in d
    names = d["gt_names"].tolist()
    assert names[0] == "car" and "ignored" not in names          # masked-out boxes are dropped when samples are pasted
    assert names.count("car") <= 4 and names.count("bus") <= 3 and len(names) > 1          # LIMIT_WHOLE_SCENE: quota minus what the scene has
    new = d["gt_boxes"][1:]
    iou = DS.bev_iou_matrix(d["gt_boxes"][:, :7], d["gt_boxes"][:, :7])
    np.fill_diagonal(iou, 0)
    assert iou.max() == 0.0          # nothing pasted overlaps the scene's boxes (the masked one included: it still occupied space) or another sample
    assert DS.bev_iou_matrix(new[:, :7], gt[:, :7]).max() == 0.0
    n_crop = 6 * len(new)
    scene_pts = d["points"][n_crop:]
    assert DS.points_in_boxes(scene_pts[:, :3], new[:, :7]).sum() == 0          # scene points inside pasted boxes were removed
    assert len(d["points"]) < len(pts) + n_crop and len(d["radar_points"]) <= len(rad) + 2 * len(new)
    # the pointer walks a permutation and reshuffles when it is used up
    grp = s.groups["bus"]
    seen = []
    for _ in range(30):
        grp["sample_num"] = 3
        seen += [id(i) for i in s._draw("bus", grp)]
    assert len(set(seen)) == len(s.db["bus"])


def test_samples_dataset_pipeline_feeds_collate(tmp_path):
    from radardistill_amd.data import collate_batch
    from radardistill_amd.pcdet.config import AttrDict
    root = tmp_path / "data"
    infos = []
    for k in range(3):
        info = AC.write_sample_files(str(root / f"s{k}"), seed=k)
        for sw in info["sweeps"]:
            sw["lidar_path"] = f"s{k}/" + sw["lidar_path"]
        info["lidar_path"] = f"s{k}/" + info["lidar_path"]
        for sweeps in info["radars"].values():
            for sw in sweeps:
                if not sw["data_path"].startswith(f"s{k}/"):
                    sw["data_path"] = f"s{k}/" + sw["data_path"]
        g = np.random.default_rng(k)
        info["gt_boxes"] = np.concatenate([g.uniform(-20, 20, (6, 2)), np.zeros((6, 1)), g.uniform(1, 4, (6, 3)), g.uniform(-3, 3, (6, 1)),
                                           g.normal(0, 1, (6, 2))], axis=1).astype(np.float32)
        info["gt_boxes"][0, 7] = np.nan
        info["gt_names"] = np.array(["car", "bus", "animal", "car", "truck", "bus"])
        info["num_lidar_pts"] = np.array([5, 0, 9, 3, 7, 2])
        infos.append(info)
    cfg = AttrDict({"POINT_CLOUD_RANGE": [-25.6, -25.6, -5.0, 25.6, 25.6, 3.0], "FILTER_MIN_POINTS_IN_GT": 1, "SET_NAN_VELOCITY_TO_ZEROS": True,
                    "PRED_VELOCITY": True,
                    "POINT_FEATURE_ENCODING": {"encoding_type": "absolute_coordinates_encoding", "used_feature_list": ["x", "y", "z", "intensity", "timestamp"],
                                               "src_feature_list": ["x", "y", "z", "intensity", "timestamp"],
                                               "radar_used_feature_list": ["x", "y", "z", "rcs", "vx_comp", "vy_comp"],
                                               "radar_src_feature_list": ["x", "y", "z", "rcs", "vx_comp", "vy_comp"]},
                    "DATA_AUGMENTOR": {"DISABLE_AUG_LIST": ["gt_sampling_distill"],
                                       "AUG_CONFIG_LIST": [{"NAME": "gt_sampling_distill"},
                                                           {"NAME": "random_world_flip_distill", "ALONG_AXIS_LIST": ["x", "y"]},
                                                           {"NAME": "random_world_rotation_distill", "WORLD_ROT_ANGLE": [-0.3, 0.3]},
                                                           {"NAME": "random_world_scaling_distill", "WORLD_SCALE_RANGE": [0.95, 1.05]},
                                                           {"NAME": "random_world_translation_distill", "NOISE_TRANSLATE_STD": [0.2, 0.2, 0.2]}]},
                    "DATA_PROCESSOR": [{"NAME": "mask_points_and_boxes_outside_range", "REMOVE_OUTSIDE_BOXES": True},
                                       {"NAME": "shuffle_points", "SHUFFLE_ENABLED": {"train": True, "test": False}},
                                       {"NAME": "transform_points_to_voxels_placeholder", "VOXEL_SIZE": [0.2, 0.2, 8.0]}]})
    ds = DS.NuScenesDistillSamples(infos, cfg, ["car", "truck", "bus"], root, training=True)
    assert list(ds.grid_size[:2]) == [256, 256] and ds.point_feature_encoder.radar_num_point_features == 6
    np.random.seed(0)
    samples = [ds[i] for i in range(3)]
    for s in samples:
        assert s["points"].shape[1] == 5 and s["radar_points"].shape[1] == 6 and s["gt_boxes"].shape[1] == 10
        assert np.all(np.abs(s["points"][:, :2]) <= 25.6) and np.all(np.abs(s["radar_points"][:, :2]) <= 25.6)
        assert set(s["gt_boxes"][:, -1].astype(int)) <= {1, 2, 3} and not np.isnan(s["gt_boxes"]).any()
        assert "gt_names" not in s and "gt_boxes_mask" not in s
    batch = collate_batch(samples)
    assert batch["batch_size"] == 3 and batch["points"].shape[1] == 6 and batch["radar_points"].shape[1] == 7
    assert set(batch["points"][:, 0].astype(int)) == {0, 1, 2} and batch["gt_boxes"].shape[0] == 3
    ev = DS.NuScenesDistillSamples(infos, cfg, ["car", "truck", "bus"], root, training=False)
    np.random.seed(0)
    e = ev[0]
    assert e["gt_boxes"].shape[1] == 10 and "noise_rot" not in e
