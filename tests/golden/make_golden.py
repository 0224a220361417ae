"""Generate golden vectors from the reference's own leaf modules (CPU, build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference tree (/root/reference) is imported leaf-by-leaf through tests/golden/_ref_loader.py;
it never travels to the GPU box -- only the small .npz fixtures written here do.  Inputs are
regenerated in the tests from seeds (radardistill_amd.synthetic / numpy default_rng); weights are
regenerated with tests/seeded.py::seeded_fill_, so fixtures hold expected OUTPUTS only (plus the
few inputs that are cheaper to store than to re-derive).

What each fixture pins is listed in DESIGN.md ("Oracle").  Third-party pieces that are absent here
(torch_scatter, the DCN and iou3d CUDA extensions) are bridged by the oracle's restatement, so
those specific arithmetic kernels are NOT pinned by these fixtures (they have their own
known-answer tests).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden import _ref_loader as L  # noqa: E402
from tests.seeded import seeded_fill_  # noqa: E402
from radardistill_amd.synthetic import make_batch, bench_geometry  # noqa: E402
from oracle import bev as obev, head as ohead  # noqa: E402

torch.set_num_threads(4)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


def g1_vfe():
    mod = L.load("pcdet.models.backbones_3d.vfe.dynamic_pillar_vfe")
    pc_range, voxel, grid = bench_geometry(128)
    cfg = L.AttrDict(WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, USE_CLUSTER_XYZ=True, USE_NORM=True, NUM_FILTERS=[32])
    batch = make_batch(batch_size=2, n_lidar=2000, n_radar=1000, n_boxes=4, grid=128, seed=1)
    out = {}
    for tag, cls, key, nfeat in (("radar", mod.Radar_DynamicPillarVFESimple2D, "radar_points", 6),
                                 ("lidar", mod.DynamicPillarVFESimple2D, "points", 5)):
        m = cls(model_cfg=cfg, num_point_features=nfeat, voxel_size=voxel, grid_size=grid, point_cloud_range=pc_range)
        sd = m.state_dict(); seeded_fill_(sd, seed=11); m.load_state_dict(sd)
        pts = torch.from_numpy(batch[key]).clone()
        # a few points outside the range / on the border to exercise the mask
        pts[:5, 1] = torch.tensor([pc_range[3] + 0.05, pc_range[0] - 0.01, pc_range[3], pc_range[0], 0.0])
        for mode in ("eval", "train"):
            m.train(mode == "train")
            bd = m({key: pts.clone(), "batch_size": 2})
            pre = "radar_" if tag == "radar" else ""
            out[f"{tag}_{mode}_features"] = bd[pre + "pillar_features"]
            out[f"{tag}_{mode}_coords"] = bd[pre + "pillar_coords"]
        out[f"{tag}_running_mean_after"] = m.state_dict()["pfn_layers.0.norm.running_mean"]
        out[f"{tag}_running_var_after"] = m.state_dict()["pfn_layers.0.norm.running_var"]
    save("g1_vfe.npz", **out)


BEV_CFG = dict(LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[256, 256], UPSAMPLE_STRIDES=[1, 2],
               NUM_UPSAMPLE_FILTERS=[128, 128])


def _bev_inputs(seed, B=1, S=16):
    g = np.random.default_rng(seed)
    x4 = g.normal(0, 1, size=(B, 256, S, S)).astype(np.float32)
    x4 *= (g.uniform(size=(B, 1, S, S)) < 0.4)            # sparse-looking map, like x_conv4.dense()
    x5 = g.normal(0, 1, size=(B, 256, S // 2, S // 2)).astype(np.float32)
    return torch.from_numpy(x4), torch.from_numpy(x5)


def g2_dense_enc():
    mod = L.load("pcdet.models.backbones_2d.base_bev_backbone")
    m = mod.BaseBEVBackboneV2(L.AttrDict(BEV_CFG), input_channels=256)
    sd = m.state_dict(); seeded_fill_(sd, seed=12); m.load_state_dict(sd)
    x4, x5 = _bev_inputs(21)
    out = {}
    for mode in ("eval", "train"):
        m.train(mode == "train")
        d = m({"multi_scale_2d_features": {"x_conv4": x4, "x_conv5": x5}})
        out[f"{mode}_2d_8x"] = d["spatial_features_2d_8x"]
        out[f"{mode}_2d"] = d["spatial_features_2d"]
    save("g2_dense_enc.npz", **out)


def _install_dcn_bridge():
    """The reference DCN op needs the compiled `DCN` extension (absent).  Bridge the module-level class
    with the oracle restatement so the rest of ConvNeXtBlock / Radar_Distill runs as shipped."""
    import math
    import torch.nn as nn

    class ModulatedDeformConv(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1,
                     deformable_groups=1, im2col_step=64, bias=True):
            super().__init__()
            self.stride, self.padding = stride, padding
            self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, kernel_size, kernel_size))
            self.bias = nn.Parameter(torch.zeros(out_channels))
            if not bias:
                self.bias.requires_grad = False

        def forward(self, x, offset, mask):
            return obev.modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding)

    L._stub("pcdet.ops.basicblock.modules.modulated_deform_conv", ModulatedDeformConv=ModulatedDeformConv)


def g3_radar_distill():
    _install_dcn_bridge()
    mod = L.load("pcdet.models.backbones_2d.radar_distill_final")
    cfg = L.AttrDict(dict(BEV_CFG, VOXEL_SIZE=[0.2, 0.2, 8.0], POINT_CLOUD_RANGE=[-12.8, -12.8, -5, 12.8, 12.8, 3]))
    m = mod.Radar_Distill(cfg, input_channels=256)
    sd = m.state_dict(); seeded_fill_(sd, seed=13); m.load_state_dict(sd)
    x4, x5 = _bev_inputs(22, B=2)
    out = {"n_params": np.array(sum(p.numel() for p in m.parameters()))}
    keys = sorted(m.state_dict().keys())
    out["state_keys"] = np.array(keys)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        d = m({"radar_multi_scale_2d_features": {"x_conv4": x4.clone(), "x_conv5": x5.clone()}})
        ms = d["radar_multi_scale_2d_features"]
        out[f"{mode}_8x_2"] = ms["radar_spatial_features_8x_2"]
        out[f"{mode}_8x_1"] = ms["radar_spatial_features_8x_1"]
        out[f"{mode}_2d_8x"] = d["radar_spatial_features_2d_8x"]
        out[f"{mode}_2d"] = d["radar_spatial_features_2d"]
    # losses on fresh random maps (AFD / PFD), incl. the NaN edge case
    g = np.random.default_rng(23)
    lid = torch.from_numpy(g.normal(0.2, 1, size=(2, 256, 16, 16)).astype(np.float32)) * \
        torch.from_numpy((g.uniform(size=(2, 1, 16, 16)) < 0.5).astype(np.float32))
    rad = torch.from_numpy(g.normal(0.0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    f, ml = m.low_loss(lid, rad)
    out["afd_feature"], out["afd_mask"] = f, ml
    # every lidar cell active -> no (radar active, lidar inactive) cell -> ratio x/0 = inf, 0*inf -> NaN
    f2, ml2 = m.low_loss(lid.abs() + 1.0, rad)
    out["afd_feature_nan"], out["afd_mask_nan"] = f2, ml2
    hms = [torch.from_numpy(g.uniform(0, 1, size=(2, c, 16, 16)).astype(np.float32) ** 6) for c in (1, 2, 2, 1, 2, 2)]
    logits = [{"hm": torch.from_numpy(g.normal(-2.0, 1.5, size=(2, c, 16, 16)).astype(np.float32))} for c in (1, 2, 2, 1, 2, 2)]
    r1 = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    r2 = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    l1 = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    l2 = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    out["pfd"] = m.high_loss(r1, r2, l1, l2, hms, logits)
    bd = {"multi_scale_2d_features": {"x_conv4": lid},
          "radar_multi_scale_2d_features": {"radar_spatial_features_8x_2": rad, "radar_spatial_features_8x_1": r1},
          "radar_spatial_features_2d": r1, "spatial_features_2d": l1,
          "radar_spatial_features_2d_8x": r2, "spatial_features_2d_8x": l2,
          "radar_pred_dicts": logits, "target_dicts": {"heatmaps": hms}}
    total, tb = m.get_loss(bd)
    out["get_loss_total"] = total
    for k, v in tb.items():
        out["tb_" + k] = np.float32(v)
    save("g3_radar_distill.npz", **out)


from tests.golden.head_cfg import HEAD_CFG, CLASS_NAMES  # noqa: E402


def g4_center_head():
    # bridge the compiled iou3d extension with the oracle's C restatement (that arithmetic is then unpinned here)
    L._stub("pcdet.ops.iou3d_nms.iou3d_nms_utils", boxes_aligned_iou3d_gpu=ohead.boxes_aligned_iou3d)
    L.load("pcdet.models.model_utils.centernet_utils")
    sys.modules["pcdet.utils.box_utils"].bbox3d_overlaps_diou = sys.modules[
        "pcdet.models.model_utils.centernet_utils"].bbox3d_overlaps_diou
    mod = L.load("pcdet.models.dense_heads.radar_center_head")
    pc_range, voxel, grid = bench_geometry(128)
    m = mod.Radar_CenterHead(L.AttrDict(HEAD_CFG), input_channels=256, num_class=10, class_names=CLASS_NAMES,
                             grid_size=grid, point_cloud_range=pc_range, voxel_size=voxel,
                             predict_boxes_when_training=False)
    sd = m.state_dict(); seeded_fill_(sd, seed=14); m.load_state_dict(sd)
    batch = make_batch(batch_size=2, n_lidar=16, n_radar=16, n_boxes=12, grid=128, seed=4)
    gt = torch.from_numpy(batch["gt_boxes"]).clone()
    gt[1, -3:, :] = 0            # padded rows (class 0 = 'bg'), as collate_batch produces
    gt[0, 0, :2] = torch.tensor([pc_range[3] - 0.05, pc_range[1] + 0.05])   # near the border: clipped gaussian
    g = np.random.default_rng(24)
    feat = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    m.train()
    torch.set_grad_enabled(True)
    d = m({"radar_spatial_features_2d": feat, "gt_boxes": gt.clone(), "batch_size": 2})
    out = {"gt_boxes": gt}
    for h, pd in enumerate(d["radar_pred_dicts"]):
        for k, v in pd.items():
            out[f"pred_{h}_{k}"] = v.detach()
    td = d["target_dicts"]
    for h in range(6):
        out[f"hm_{h}"] = td["heatmaps"][h]; out[f"tb_{h}"] = td["target_boxes"][h]
        out[f"ind_{h}"] = td["inds"][h]; out[f"mask_{h}"] = td["masks"][h]; out[f"gtbox_{h}"] = td["gt_box"][h]
    loss, tb = m.get_loss()
    torch.set_grad_enabled(False)
    out["loss"] = loss.detach()
    for k, v in tb.items():
        out["tb_" + k] = np.float32(v)
    save("g4_center_head.npz", **out)


def g5_conv5():
    L._stub("pcdet.utils.spconv_utils", replace_feature=None,
            spconv=type("S", (), {"SparseModule": torch.nn.Module, "SparseSequential": torch.nn.Sequential}))
    mod = L.load("pcdet.models.backbones_3d.spconv_backbone_2d")
    from functools import partial
    norm = partial(torch.nn.BatchNorm2d, eps=1e-3, momentum=0.01)
    m = torch.nn.Sequential(mod.post_act_block_dense(256, 256, 3, norm_fn=norm, stride=2, padding=1),
                            mod.BasicBlock(256, 256, norm_fn=norm), mod.BasicBlock(256, 256, norm_fn=norm))
    sd = m.state_dict(); seeded_fill_(sd, seed=15); m.load_state_dict(sd)
    x4, _ = _bev_inputs(25)
    out = {}
    for mode in ("eval", "train"):
        m.train(mode == "train")
        out[f"{mode}_x_conv5"] = m(x4)
    save("g5_conv5.npz", **out)


def g6_decode():
    """Eval path of the reference head: generate_predicted_boxes -> decode_bbox_from_heatmap -> class_agnostic_nms, with the absent
    iou3d extension bridged by the oracle's C restatement of nms_gpu (so the rotated-overlap arithmetic is not pinned here)."""
    from oracle import post as opost
    L.install()
    L._stub("pcdet.ops.iou3d_nms.iou3d_nms_utils", boxes_aligned_iou3d_gpu=ohead.boxes_aligned_iou3d, nms_gpu=opost.nms_gpu)
    L.load("pcdet.models.model_utils.centernet_utils")
    sys.modules.pop("pcdet.models.model_utils.model_nms_utils", None)      # the loader's inert placeholder -> the real leaf file
    L.load("pcdet.models.model_utils.model_nms_utils")
    sys.modules["pcdet.utils.box_utils"].bbox3d_overlaps_diou = sys.modules[
        "pcdet.models.model_utils.centernet_utils"].bbox3d_overlaps_diou
    mod = L.load("pcdet.models.dense_heads.radar_center_head")
    pc_range, voxel, grid = bench_geometry(128)
    m = mod.Radar_CenterHead(L.AttrDict(HEAD_CFG), input_channels=256, num_class=10, class_names=CLASS_NAMES,
                             grid_size=grid, point_cloud_range=pc_range, voxel_size=voxel,
                             predict_boxes_when_training=False)
    sd = m.state_dict(); seeded_fill_(sd, seed=14); m.load_state_dict(sd)
    g = np.random.default_rng(26)
    feat = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    m.eval()
    d = m({"radar_spatial_features_2d": feat, "batch_size": 2})
    out = {}
    for h, pd in enumerate(m.forward_ret_dict["pred_dicts"]):
        for k, v in pd.items():
            out[f"pred_{h}_{k}"] = v.detach()
    for b, fd in enumerate(d["final_box_dicts"]):
        out[f"boxes_{b}"] = fd["pred_boxes"]; out[f"scores_{b}"] = fd["pred_scores"]; out[f"labels_{b}"] = fd["pred_labels"]
        print("sample", b, "detections", tuple(fd["pred_boxes"].shape))
    save("g6_decode.npz", **out)


def g7_pillar():
    """PillarVFE + PointPillarScatter of the reference on padded voxels produced by the oracle's voxeliser (spconv is absent)."""
    from oracle import voxel as ovox
    pv = L.load("pcdet.models.backbones_3d.vfe.pillar_vfe")
    L._ns("pcdet.models.backbones_2d.map_to_bev", os.path.join(L.REF_ROOT, "pcdet/models/backbones_2d/map_to_bev"))
    sc = L.load("pcdet.models.backbones_2d.map_to_bev.pointpillar_scatter")
    pc_range, voxel, grid = bench_geometry(128)
    voxel = [voxel[0], voxel[1], pc_range[5] - pc_range[2]]              # pillars: one cell in z
    batch = make_batch(batch_size=2, n_lidar=1500, n_radar=16, n_boxes=2, grid=128, seed=8)
    vox, coords, num = ovox.batch_points_to_voxels(batch["points"], 2, voxel, pc_range, max_points=8, max_voxels=700)
    out = {}
    for tag, use_abs, with_dist in (("a", True, False), ("b", False, True)):
        cfg = L.AttrDict(USE_NORM=True, WITH_DISTANCE=with_dist, USE_ABSLOTE_XYZ=use_abs, NUM_FILTERS=[64])
        m = pv.PillarVFE(cfg, num_point_features=5, voxel_size=voxel, point_cloud_range=pc_range)
        sd = m.state_dict(); seeded_fill_(sd, seed=31); m.load_state_dict(sd)
        for mode in ("eval", "train"):
            m.train(mode == "train")
            bd = {"voxels": torch.from_numpy(vox), "voxel_num_points": torch.from_numpy(num), "voxel_coords": torch.from_numpy(coords)}
            bd = m(bd)
            out[f"{tag}_{mode}_pillar_features"] = bd["pillar_features"]
            if mode == "train":
                out[f"{tag}_running_mean"] = m.pfn_layers[0].norm.running_mean.clone()
                out[f"{tag}_running_var"] = m.pfn_layers[0].norm.running_var.clone()
        if tag == "a":
            s = sc.PointPillarScatter(L.AttrDict(NUM_BEV_FEATURES=64), grid_size=[int(grid[0]), int(grid[1]), 1])
            m.eval()
            bd = m({"voxels": torch.from_numpy(vox), "voxel_num_points": torch.from_numpy(num), "voxel_coords": torch.from_numpy(coords)})
            out["spatial_features"] = s(bd)["spatial_features"]
    save("g7_pillar.npz", **out)


G9_CASES = (("a", True, False, True, [64]), ("c", True, False, True, [64, 64]), ("d", False, True, True, [32, 128]), ("e", True, False, False, [64]))


def g9_pillar_train():
    """PillarVFE of the reference in TRAINING use: forward + parameter gradients of loss = sum(features * go) for one PFN layer, two
    layers (the [x | max] concatenation), a 16-channel first layer and USE_NORM False (pillar_vfe.py:8-49,94-123)."""
    from oracle import voxel as ovox
    pv = L.load("pcdet.models.backbones_3d.vfe.pillar_vfe")
    pc_range, voxel, grid = bench_geometry(128)
    voxel = [voxel[0], voxel[1], pc_range[5] - pc_range[2]]
    batch = make_batch(batch_size=2, n_lidar=1500, n_radar=16, n_boxes=2, grid=128, seed=8)
    vox, coords, num = ovox.batch_points_to_voxels(batch["points"], 2, voxel, pc_range, max_points=8, max_voxels=700)
    out = {}
    torch.set_grad_enabled(True)
    for tag, use_abs, with_dist, use_norm, filters in G9_CASES:
        cfg = L.AttrDict(USE_NORM=use_norm, WITH_DISTANCE=with_dist, USE_ABSLOTE_XYZ=use_abs, NUM_FILTERS=filters)
        m = pv.PillarVFE(cfg, num_point_features=5, voxel_size=voxel, point_cloud_range=pc_range)
        sd = m.state_dict(); seeded_fill_(sd, seed=41); m.load_state_dict(sd)
        m.train()
        bd = m({"voxels": torch.from_numpy(vox), "voxel_num_points": torch.from_numpy(num), "voxel_coords": torch.from_numpy(coords)})
        feats = bd["pillar_features"]
        go = torch.from_numpy(np.random.default_rng(9).normal(size=tuple(feats.shape)).astype(np.float32))
        (feats * go).sum().backward()
        out[f"{tag}_features"] = feats.detach()             # `go` is regenerated by the tests from the same seed
        for k, p in m.named_parameters():
            out[f"{tag}_grad_{k}"] = p.grad.detach()
        for k, b in m.named_buffers():
            if "running" in k:
                out[f"{tag}_{k}"] = b.detach().clone()
        m.eval()
        with torch.no_grad():
            bd = m({"voxels": torch.from_numpy(vox), "voxel_num_points": torch.from_numpy(num), "voxel_coords": torch.from_numpy(coords)})
        out[f"{tag}_eval_features"] = bd["pillar_features"]
        print(tag, "features", tuple(feats.shape), "params", [k for k, _ in m.named_parameters()])
    torch.set_grad_enabled(False)
    save("g9_pillar_train.npz", **out)


def g8_optim():
    """A13: the reference's own build_optimizer / OptimWrapper / OneCycle + clip_grad_norm_ loop (tools/train_utils/optimization/
    __init__.py:19-54, fastai_optim.py:104-235, learning_schedules_fastai.py:44-77, train_utils.py:44-64) on the small problem
    of tests/golden/optim_case.py.  The package is pure torch and imports as shipped."""
    import importlib.util
    from torch.nn.utils import clip_grad_norm_
    from tests.golden import optim_case as OC
    pkg_dir = os.path.join(L.REF_ROOT, "tools/train_utils/optimization")
    spec = importlib.util.spec_from_file_location("ref_optimization", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    ref = importlib.util.module_from_spec(spec)
    sys.modules["ref_optimization"] = ref
    spec.loader.exec_module(ref)
    torch.manual_seed(0)
    model = OC.OptimCaseNet()
    sd = model.state_dict(); seeded_fill_(sd, seed=18); model.load_state_dict(sd)
    cfg = L.AttrDict(OC.OPTIM_CFG)
    opt = ref.build_optimizer(model, cfg)
    sched, _ = ref.build_scheduler(opt, total_iters_each_epoch=OC.TOTAL_ITERS_EACH_EPOCH, total_epochs=OC.TOTAL_EPOCHS, last_epoch=-1, optim_cfg=cfg)
    names = [n for n, _ in OC.trainable(model)]
    out = {"names": np.array(names), "lr0": np.float64(opt.lr), "mom0": np.float64(opt.mom)}
    lrs, moms, norms, snaps = [], [], [], []
    for it in range(OC.N_STEPS):
        sched.step(it)
        lrs.append(float(opt.lr)); moms.append(float(opt.mom))
        model.train()
        opt.zero_grad()
        OC.assign_grads(model, it)
        norms.append(float(clip_grad_norm_(model.parameters(), cfg.GRAD_NORM_CLIP)))
        opt.step()
        snaps.append(torch.cat([p.detach().reshape(-1) for _, p in OC.trainable(model)]).clone())
    out["lr"], out["mom"], out["total_norm"] = np.array(lrs), np.array(moms), np.array(norms, dtype=np.float32)
    out["params_after"] = torch.stack(snaps)
    # inner torch.optim.Adam state in the reference's own parameter numbering (two groups: non-BatchNorm leaves, BatchNorm leaves)
    osd = opt.state_dict()
    by_id = {id(p): n for n, p in model.named_parameters()}
    order = [by_id[id(p)] for g in opt.opt.param_groups for p in g["params"]]
    out["ref_param_order"] = np.array(order)
    out["ref_group_sizes"] = np.array([len(g["params"]) for g in osd["param_groups"]])
    for idx, name in enumerate(order):
        st = osd["state"].get(idx)
        out[f"has_state_{name}"] = np.array(st is not None)
        if st is not None:
            out[f"step_{name}"] = np.array(int(st["step"]))
            out[f"exp_avg_{name}"] = st["exp_avg"]; out[f"exp_avg_sq_{name}"] = st["exp_avg_sq"]
    save("g8_optim.npz", **out)


def g10_augment():
    """The `_distill` world transforms, the augmentor's tail and the nuScenes sweep assembly, produced by the reference's own functions
    (pcdet/datasets/augmentor/augmentor_utils.py:28-180, data_augmentor.py:239-262,396-425, nuscenes/nuscenes_dataset_distill.py:82-117,
    240-283) under seeded global numpy generators.  common_utils is the reference's real file here (rotate_points_along_z,
    limit_period); box_utils and the dataset base classes stay inert placeholders (not called by these paths)."""
    import tempfile
    import types
    from pathlib import Path
    from tests.golden import augment_case as AC
    L.install()
    del sys.modules["pcdet.utils.common_utils"]                      # the loader's reduced stand-in -> the real module
    L._ns("pcdet.datasets", f"{L.REF_ROOT}/pcdet/datasets")
    L._ns("pcdet.datasets.augmentor", f"{L.REF_ROOT}/pcdet/datasets/augmentor")
    L._ns("pcdet.datasets.nuscenes", f"{L.REF_ROOT}/pcdet/datasets/nuscenes")
    L._ns("pcdet.ops.roiaware_pool3d", f"{L.REF_ROOT}/pcdet/ops/roiaware_pool3d")
    L._stub("pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils")
    L._stub("pcdet.datasets.dataset_distill", DatasetTemplate_Distill=object)
    L._stub("pcdet.datasets.nuscenes.nuscenes_dataset", NuScenesDataset=object)
    L._stub("pyquaternion", Quaternion=None)
    L._stub("pcdet.utils.spconv_utils", spconv=None)                # common_utils imports it for an unrelated helper
    L._stub("nuscenes"); L._stub("nuscenes.utils"); L._stub("nuscenes.utils.data_classes", RadarPointCloud=None)
    CU = L.load("pcdet.utils.common_utils")
    AU = L.load("pcdet.datasets.augmentor.augmentor_utils")
    for n in ("database_sampler", "database_sampler_distill", "database_sampler_radar"):
        L._stub("pcdet.datasets.augmentor." + n)
    DA = L.load("pcdet.datasets.augmentor.data_augmentor").DataAugmentor
    out = {}
    for seed in AC.SEEDS:
        b, p, r = AC.scene(seed)
        np.random.seed(seed)
        for axis in ("x", "y"):
            b, p, r, en = getattr(AU, "random_flip_distill_along_%s" % axis)(b, p, r, return_flip=True)
            out[f"flip_{axis}_{seed}"] = np.array(bool(en))
        b, p, r, rot = AU.global_rotation_distill(b, p, r, rot_range=AC.ROT_RANGE, return_rot=True)
        b, p, r, sc = AU.global_scaling_distill(b, p, r, AC.SCALE_RANGE, return_scale=True)
        d = DA.random_world_translation_distill(None, {"gt_boxes": b, "points": p, "radar_points": r}, {"NOISE_TRANSLATE_STD": AC.TRANSLATE_STD})
        # the augmentor's tail (forward :407-425) on an empty queue: heading wrapped, class mask applied
        shell = types.SimpleNamespace(data_augmentor_queue=[])
        mask = np.arange(len(b)) % 3 != 1
        d.update(gt_names=np.array(["car", "bus", "truck"] * 3)[:len(b)], gt_boxes_mask=mask)
        d["gt_boxes"][:, 6] += 3.0 * (seed - 3)                     # headings far outside [-pi, pi)
        d = DA.forward(shell, d)
        out[f"rot_{seed}"], out[f"scale_{seed}"], out[f"translate_{seed}"] = np.array(rot), np.array(sc), d["noise_translate"]
        out[f"boxes_{seed}"], out[f"points_{seed}"], out[f"radar_{seed}"] = d["gt_boxes"], d["points"], d["radar_points"]
    ND = L.load("pcdet.datasets.nuscenes.nuscenes_dataset_distill").NuScenesDataset_Distill
    with tempfile.TemporaryDirectory() as tmp:
        info = AC.write_sample_files(os.path.join(tmp, "data"), seed=0)
        fake = types.SimpleNamespace(infos=[info], root_path=Path(tmp) / "data" / "v")          # the reference reads root_path.parent / path
        fake.get_sweep = types.MethodType(ND.get_sweep, fake)
        fake._load_points = lambda path: np.fromfile(os.path.join(tmp, "data", path), dtype=np.float32)
        np.random.seed(5)
        out["lidar_sweeps"] = ND.get_lidar_with_sweeps(fake, 0, max_sweeps=10)
        out["radar_sweeps"] = ND.get_radar_with_sweeps(fake, 0, max_sweeps=6)
        sw, tl = ND.get_sweep(fake, info["sweeps"][3])
        out["sweep3_points"], out["sweep3_times"] = sw, tl
    save("g10_augment.npz", **out)


def _fresh_reference_modules():
    """Every set starts from a clean slate: leaf files of the reference (and the placeholders / bridges an earlier set installed for
    them, e.g. g4's reduced iou3d_nms_utils) are dropped from sys.modules so the next set imports what IT needs."""
    for k in [k for k in sys.modules if k == "pcdet" or k.startswith("pcdet.") or k == "ref_optimization" or k.startswith("ref_optimization.")]:
        del sys.modules[k]


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10"]
    fns = {"g1": g1_vfe, "g2": g2_dense_enc, "g3": g3_radar_distill, "g4": g4_center_head, "g5": g5_conv5, "g6": g6_decode, "g7": g7_pillar,
           "g8": g8_optim, "g9": g9_pillar_train, "g10": g10_augment}
    for w in which:
        _fresh_reference_modules()
        torch.set_grad_enabled(False)
        fns[w]()
