"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the package imports, the
reference's yaml loads through the config surface, registries carry the reference's NAME strings."""
import os

import numpy as np
import pytest
import torch


def test_library_exports_every_header_symbol():
    from radardistill_amd import native
    native.build()
    L = native.lib()
    syms = native.header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), s
    assert set(syms) == set(native.SIGNATURES)
    assert L.rd_abi_version() == 3
    assert L.rd_rankgrid_bytes(32 * 1024) == (2 * 1024 + 1 + 1) * 4
    C = native.ctypes_lib()          # the same entry points through ctypes (what a reference maintainer's stub would use)
    for s in syms:
        assert hasattr(C, s), s
    assert C.rd_abi_version() == 3 and C.rd_rankgrid_bytes(32 * 1024) == L.rd_rankgrid_bytes(32 * 1024)
    with pytest.raises(TypeError):
        L.rd_rankgrid_bytes()          # the generated callers check their argument count


def test_generated_caller_marshals_like_ctypes():
    """csrc/_rdcall.so (generated from native.SIGNATURES): a struct pointer may be a ctypes Structure instance (its own storage),
    an address or None; integers and floats are parsed directly; a wrong argument count or type is a TypeError, not a crash.
    Host-only entry points, no GPU."""
    import ctypes
    from radardistill_amd import native
    native.build()
    L, C = native.lib(), native.ctypes_lib()
    cfg = native.CenterLossCfg()
    cfg.n_heads, cfg.B, cfg.K = 6, 8, 500
    want = C.rd_center_loss_ws_floats(ctypes.byref(cfg))
    assert want == 32 + 8 * 13 + 6 * 8 * 500 * 30
    assert L.rd_center_loss_ws_floats(cfg) == want                              # structure instance
    assert L.rd_center_loss_ws_floats(ctypes.addressof(cfg)) == want            # plain address
    assert L.rd_center_loss_ws_floats(None) == 0                                # NULL
    assert L.rd_weight_layout_split_items(256, 256, 9, 1) == C.rd_weight_layout_split_items(256, 256, 9, 1) == 16 * 4
    assert L.rd_weight_layout_split_items(100, 64, 16, 8) == 4 * 2 * 2          # A = Cin 64 -> 4 tiles, B = Cout 100 -> 2, taps 16 -> 2 chunks
    with pytest.raises(TypeError):
        L.rd_center_loss_ws_floats("not a pointer")
    with pytest.raises(TypeError):
        L.rd_weight_layout_split_items(1, 2, 3)
    msg = L.rd_last_error()
    assert msg is None or isinstance(msg, bytes)


def test_missing_extension_fails_loudly(monkeypatch):
    from radardistill_amd import native
    monkeypatch.setattr(native, "SO_PATH", "/nonexistent/librdamd.so")
    monkeypatch.setattr(native, "_LIB", None)
    monkeypatch.setattr(native, "_CTYPES", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.lib()


def test_cpu_tensor_is_rejected_not_computed():
    """The product has no CPU path: handing it CPU tensors is an error, never a silent fallback."""
    from radardistill_amd import kernels as K
    with pytest.raises(RuntimeError):
        K.bn_stats(torch.zeros(8, 32))


def test_package_imports_and_registries():
    import importlib
    for mod in ["radardistill_amd", "radardistill_amd.kernels", "radardistill_amd.autograd", "radardistill_amd.sparse",
                "radardistill_amd.pcdet", "radardistill_amd.pcdet.config", "radardistill_amd.pcdet.models",
                "radardistill_amd.pcdet.utils.spconv_utils", "radardistill_amd.lowp", "radardistill_amd.graphs", "radardistill_amd.ckpt"]:
        importlib.import_module(mod)
    from radardistill_amd.pcdet.models.backbones_3d import __all__ as B3
    from radardistill_amd.pcdet.models.backbones_3d.vfe import __all__ as VFE
    assert {"DynamicPillarVFESimple2D", "Radar_DynamicPillarVFESimple2D"} <= set(VFE)
    assert {"PillarRes18BackBone8x", "Radar_PillarRes18BackBone8x"} <= set(B3)
    m = B3["Radar_PillarRes18BackBone8x"](None, 32, np.array([128, 128, 40]))
    assert sum(p.numel() for p in m.parameters()) == 6479872            # radar SparseEnc parameters
    from radardistill_amd.pcdet.utils.spconv_utils import find_all_spconv_keys
    assert "conv2.0.0.weight" in find_all_spconv_keys(m) and len(find_all_spconv_keys(m)) == 19


@pytest.mark.skipif(not os.path.isdir("/root/reference/tools/cfgs"), reason="reference tree only exists in the build container")
def test_reference_yaml_loads_unchanged(monkeypatch):
    from radardistill_amd.pcdet.config import AttrDict, cfg_from_list, cfg_from_yaml_file
    monkeypatch.chdir("/root/reference/tools")
    c = cfg_from_yaml_file("cfgs/radar_distill/radar_distill_train.yaml", AttrDict())
    assert c.MODEL.NAME == "PillarNet" and c.MODEL.RADAR_BACKBONE_2D.NAME == "Radar_Distill"
    assert c.MODEL.FREEZE_PIPELINE == ["DynamicPillarVFESimple2D", "PillarRes18BackBone8x", "BaseBEVBackboneV2", "CenterHead"]
    assert c.DATA_CONFIG.POINT_CLOUD_RANGE == [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]      # overrides the _BASE_CONFIG_
    assert c.DATA_CONFIG.POINT_FEATURE_ENCODING.radar_used_feature_list == ['x', 'y', 'z', 'rcs', 'vx', 'vy']
    cfg_from_list(["OPTIMIZATION.LR", "0.002", "DATA_CONFIG.DATA_AUGMENTOR.DISABLE_AUG_LIST", "gt_sampling_distill,foo"], c)
    assert c.OPTIMIZATION.LR == 0.002 and c.DATA_CONFIG.DATA_AUGMENTOR.DISABLE_AUG_LIST == ["gt_sampling_distill", "foo"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/tools/cfgs"), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("yaml_name,modules,total,trainable", [
    ("radar_distill_train.yaml", ["DynamicPillarVFESimple2D", "Radar_DynamicPillarVFESimple2D", "PillarRes18BackBone8x", "Radar_PillarRes18BackBone8x",
                                  "BaseBEVBackboneV2", "Radar_Distill", "CenterHead", "Radar_CenterHead"], 41075593, 24910077),
    ("radar_distill_val.yaml", ["Radar_DynamicPillarVFESimple2D_Test", "Radar_PillarRes18BackBone8x", "Radar_Distill", "Radar_CenterHead"],
     24910845, 24910077)])
def test_build_network_on_the_unmodified_reference_yamls(monkeypatch, yaml_name, modules, total, trainable):
    """tools/train.py:143 / tools/test.py: cfg_from_yaml_file + build_network on the reference's own files, dataset facts taken from
    the yaml as the reference's dataset does (VOXEL_SIZE from DATA_PROCESSOR, feature counts from POINT_FEATURE_ENCODING).
    Parameter counts are the reference modules' (512 + 544 + 2 x 6 479 872 + 7 936 512 + 16 682 577 + 2 x 1 747 852;
    768 frozen DCN biases among the student's)."""
    from radardistill_amd.data import SyntheticDistillDataset
    from radardistill_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from radardistill_amd.pcdet.models import build_network
    monkeypatch.chdir("/root/reference/tools")
    c = cfg_from_yaml_file("cfgs/radar_distill/" + yaml_name, AttrDict())
    ds = SyntheticDistillDataset.from_cfg(c)
    assert list(ds.grid_size) == [1440, 1440, 40] and list(ds.voxel_size) == [0.075, 0.075, 0.2]
    m = build_network(model_cfg=c.MODEL, num_class=len(c.CLASS_NAMES), dataset=ds)
    assert [type(x).__name__ for x in m.module_list] == modules
    assert sum(p.numel() for p in m.parameters()) == total
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == trainable
    if "train" in yaml_name:
        # the student-init checkpoint of ckpt.py loads: every teacher tensor lands twice, except the radar VFE's 15-column Linear
        from radardistill_amd.ckpt import student_init_state
        teacher = {k: torch.full_like(v, 0.5) for k, v in m.state_dict().items() if not k.startswith("radar_") and k != "global_step"}
        init = student_init_state({"epoch": 1, "it": 2, "optimizer_state": None, "version": "x", "model_state": teacher})
        _, updated = m._load_state_dict(init["model_state"], strict=False)
        skipped = [k for k in init["model_state"] if k not in updated]
        assert skipped == ["radar_vfe.pfn_layers.0.linear.weight"]
        assert float(m.radar_backbone_3d.conv2[0][0].weight.mean()) == 0.5 and float(m.backbone_2d.blocks[0][1].weight.mean()) == 0.5


def test_batched_head_loss_equals_per_head_loss_and_oracle(monkeypatch):
    """Host logic only (CPU): the all-heads-at-once CenterHead loss == the reference-shaped per-head loop == the oracle.
    The rotated-overlap kernel is a GPU op; its oracle twin is patched in here because this test is about the batching."""
    from oracle import head as ohead
    from oracle.pillarnet import CLASS_NAMES, HEADS
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.dense_heads import __all__ as REG
    from radardistill_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils
    from radardistill_amd.synthetic import bench_geometry, make_batch
    from tests.golden.head_cfg import HEAD_CFG
    monkeypatch.setattr(iou3d_nms_utils, "boxes_aligned_iou3d_gpu", lambda a, b: ohead.boxes_aligned_iou3d(a.detach(), b.detach()))
    pc_range, voxel, gs = bench_geometry(128)
    m = REG["Radar_CenterHead"](AttrDict(HEAD_CFG), input_channels=256, num_class=10, class_names=CLASS_NAMES, grid_size=gs,
                                point_cloud_range=pc_range, voxel_size=voxel, predict_boxes_when_training=False)
    batch = make_batch(batch_size=2, n_lidar=16, n_radar=16, n_boxes=14, grid=128, seed=9)
    gt = torch.from_numpy(batch["gt_boxes"]).clone()
    gt[1, -4:, :] = 0                                     # padded rows; head 3 (barrier) may end up without positives in sample 1
    td = m.assign_targets(gt, feature_map_size=(16, 16), gt_boxes_host=gt.numpy())
    g = torch.Generator().manual_seed(0)
    preds = []
    for nc in (1, 2, 2, 1, 2, 2):
        d = {k: torch.randn(2, c, 16, 16, generator=g) * 0.5 for k, c in ohead.HEAD_OUT.items()}
        d["hm"] = torch.randn(2, nc, 16, 16, generator=g) - 2.0
        preds.append(d)
    for d in preds:
        for v in d.values():
            v.requires_grad_(True)
    m.forward_ret_dict = {"pred_dicts": preds, "target_dicts": td}
    loss_b, tb_b = m.get_loss()
    loss_p, tb_p = m.get_loss_per_head()
    oloss, otb = ohead.center_head_loss(preds, {k: v for k, v in td.items() if k != "_stacked"}, voxel, pc_range)
    assert abs(float(loss_b) - float(loss_p)) < 1e-4 * abs(float(loss_p)) and abs(float(loss_b) - float(oloss)) < 1e-4 * abs(float(oloss))
    for k in tb_p:
        assert abs(float(tb_b[k]) - float(tb_p[k])) <= 1e-4 * abs(float(tb_p[k])) + 1e-6, k
        assert abs(float(tb_b[k]) - float(otb[k])) <= 1e-4 * abs(float(otb[k])) + 1e-6, k
    # the stacked-output form produced by the batched branch plan: one (B, H, W, NO) tensor, columns [hm | center | ...], heads inner
    names = ["hm"] + list(ohead.HEAD_OUT.keys())
    groups, cols, c0 = {}, [], 0
    for n in names:
        widths = [p[n].shape[1] for p in preds]
        groups[n] = (c0, widths)
        cols += [p[n].detach() for p in preds]
        c0 += sum(widths)
    o4 = torch.cat(cols, dim=1).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    col = 0
    views = [dict() for _ in preds]
    for n in names:
        for h, p in enumerate(preds):
            views[h][n] = o4[..., col:col + p[n].shape[1]].permute(0, 3, 1, 2)
            col += p[n].shape[1]
    m.forward_ret_dict = {"pred_dicts": views, "target_dicts": td, "pred_stacked": (o4, groups)}
    loss_s, tb_s = m.get_loss()
    assert abs(float(loss_s) - float(loss_p)) < 1e-4 * abs(float(loss_p))
    for k in tb_p:
        assert abs(float(tb_s[k]) - float(tb_p[k])) <= 1e-4 * abs(float(tb_p[k])) + 1e-6, k
    (g_o4,) = torch.autograd.grad(loss_s.sum(), [o4])
    m.forward_ret_dict = {"pred_dicts": preds, "target_dicts": td}
    gb = torch.autograd.grad(loss_b.sum(), [preds[1]["hm"], preds[4]["dim"], preds[0]["iou"]])
    gp = torch.autograd.grad(loss_p.sum(), [preds[1]["hm"], preds[4]["dim"], preds[0]["iou"]])
    for a, b in zip(gb, gp):
        assert float((a - b).abs().max()) <= 1e-5 * (float(b.abs().max()) + 1e-6)
    c_hm1 = groups["hm"][0] + preds[0]["hm"].shape[1]
    c_dim4 = groups["dim"][0] + 4 * 3
    for a, b in ((g_o4[..., c_hm1:c_hm1 + preds[1]["hm"].shape[1]], gp[0]), (g_o4[..., c_dim4:c_dim4 + 3], gp[1])):
        assert float((a.permute(0, 3, 1, 2) - b).abs().max()) <= 1e-5 * (float(b.abs().max()) + 1e-6)


def test_collate_batch_and_dataloader():
    """collate_batch (dataset_distill.py:220-325): index column, gt padding, padded-voxel concatenation, DOUBLE_FLIP lists."""
    from radardistill_amd.data import SyntheticSweeps, collate_batch
    a = {"points": np.ones((3, 5), np.float32), "radar_points": np.full((2, 6), 2, np.float32), "gt_boxes": np.ones((4, 10), np.float32),
         "voxels": np.ones((5, 8, 5), np.float32), "voxel_coords": np.ones((5, 3), np.int32), "voxel_num_points": np.ones(5, np.int32), "frame_id": np.int64(7)}
    b = {"points": np.zeros((1, 5), np.float32), "radar_points": np.zeros((0, 6), np.float32), "gt_boxes": np.ones((2, 10), np.float32) * 3,
         "voxels": np.zeros((2, 8, 5), np.float32), "voxel_coords": np.zeros((2, 3), np.int32), "voxel_num_points": np.zeros(2, np.int32), "frame_id": np.int64(9)}
    r = collate_batch([a, b])
    assert r["batch_size"] == 2 and r["points"].shape == (4, 6) and r["points"][:, 0].tolist() == [0, 0, 0, 1]
    assert r["radar_points"].shape == (2, 7) and r["voxel_coords"].shape == (7, 4) and r["voxel_coords"][:, 0].tolist() == [0] * 5 + [1] * 2
    assert r["voxels"].shape == (7, 8, 5) and r["voxel_num_points"].shape == (7,)
    assert r["gt_boxes"].shape == (2, 4, 10) and float(r["gt_boxes"][1, 2:].sum()) == 0 and float(r["gt_boxes"][1, :2].mean()) == 3
    assert r["frame_id"].tolist() == [7, 9]
    flip = {"voxels": [np.ones((2, 8, 5), np.float32)] * 4, "voxel_coords": [np.ones((2, 3), np.int32)] * 4, "voxel_num_points": [np.ones(2, np.int32)] * 4}
    rf = collate_batch([flip, flip])
    assert rf["batch_size"] == 8 and rf["voxel_coords"][:, 0].tolist() == sum(([i] * 2 for i in range(8)), [])
    ds = SyntheticSweeps(4, grid=128, n_lidar=300, n_radar=50, n_boxes=5)
    dl = torch.utils.data.DataLoader(ds, batch_size=2, collate_fn=collate_batch, shuffle=False)
    batch = next(iter(dl))
    assert batch["batch_size"] == 2 and batch["points"].shape[1] == 6 and batch["radar_points"].shape[1] == 7 and batch["gt_boxes"].shape == (2, 5, 10)
    assert set(np.unique(batch["points"][:, 0]).tolist()) == {0.0, 1.0}


def test_checkpoint_round_trip_and_spconv1_layout(tmp_path):
    """checkpoint_state / save_checkpoint / load_params_from_file (train_utils.py:253-293, detector3d_template.py:411-470): reference
    dictionary layout, safe loading, and the spconv-1.x weight layout (k1, k2, Cin, Cout) converted on load."""
    import logging
    from radardistill_amd.train import checkpoint_state, save_checkpoint
    from radardistill_amd.pcdet.models.backbones_3d import __all__ as B3

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone_3d = B3["Radar_PillarRes18BackBone8x"](None, 32, np.array([128, 128, 40]))
            self.global_step = torch.zeros(1)

    from radardistill_amd.pcdet.models.detectors.detector3d_template import Detector3DTemplate
    net = Net()
    net._load_state_dict = Detector3DTemplate._load_state_dict.__get__(net)
    net.load_params_from_file = Detector3DTemplate.load_params_from_file.__get__(net)
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    state = checkpoint_state(net, None, epoch=3, it=77)
    assert set(state) == {"epoch", "it", "model_state", "optimizer_state", "version"}
    # an spconv-1.x checkpoint stores sparse conv kernels as (k1, k2, Cin, Cout)
    key = "backbone_3d.conv2.0.0.weight"
    assert state["model_state"][key].shape == (64, 3, 3, 32)
    state["model_state"][key] = state["model_state"][key].permute(1, 2, 3, 0).contiguous()
    save_checkpoint(state, str(tmp_path / "ckpt"))
    with torch.no_grad():
        for p in net.parameters():
            p.zero_()
    net.load_params_from_file(str(tmp_path / "ckpt.pth"), logging.getLogger("t"), to_cpu=True)
    for k, v in net.state_dict().items():
        assert torch.equal(v, ref[k]), k


def test_centernet_helpers_vs_golden_pinned_oracle(golden_dir):
    """radardistill_amd/pcdet/models/model_utils/centernet_utils.py (own formulation) against oracle/head.py + oracle/post.py, which
    fixtures g4 / g6 pin to the reference's functions: radii and splatted heat-maps bit-equal, DIoU bit-equal, decode of the g6
    network outputs equal to the oracle's candidate loop."""
    from oracle import head as ohead, post as opost
    from radardistill_amd.pcdet.models.model_utils import centernet_utils as CU
    g = torch.Generator().manual_seed(5)
    h = torch.rand(4000, generator=g) * 30 + 0.01
    w = torch.rand(4000, generator=g) * 12 + 0.01
    for ov in (0.1, 0.5, 0.7):
        assert torch.equal(CU.gaussian_radius(h, w, ov), ohead.gaussian_radius(h, w, ov))
    for r in (0, 1, 2, 5, 13):
        assert np.array_equal(CU.gaussian_patch(r).numpy(), ohead.gaussian2d(r).astype(np.float32))
    a, b = torch.zeros(20, 24), torch.zeros(20, 24)
    for (cx, cy, r) in [(0, 0, 2), (23, 19, 4), (5, 7, 3), (6, 7, 2), (12, 0, 6), (23, 3, 9), (11, 10, 30)]:
        CU.draw_gaussian_to_heatmap(a, (cx, cy), r)
        ohead.draw_gaussian(b, (cx, cy), r)
    assert torch.equal(a, b) and float(a.max()) == 1.0
    p = torch.randn(500, 7, generator=g); p[:, 3:6] = p[:, 3:6].abs() + 0.1
    q = p + 0.3 * torch.randn(500, 7, generator=g); q[:, 3:6] = q[:, 3:6].abs() + 0.1
    assert torch.equal(CU.bbox3d_overlaps_diou(p, q), ohead.diou(p, q))
    # decode: the reference head's raw outputs stored in g6
    gd = np.load(f"{golden_dir}/g6_decode.npz")
    pc_range, voxel, _ = __import__("radardistill_amd.synthetic", fromlist=["bench_geometry"]).bench_geometry(128)
    limit = torch.tensor([-61.2, -61.2, -10.0, 61.2, 61.2, 10.0])
    for head in range(6):
        pd = {k: torch.from_numpy(gd[f"pred_{head}_{k}"]) for k in ("hm", "center", "center_z", "dim", "rot", "vel", "iou")}
        hm, dim, iou = pd["hm"].sigmoid(), pd["dim"].exp(), (pd["iou"] + 1) * 0.5
        # channels-last storage, as the dense kernels hand the maps over
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)
        out = CU.decode_bbox_from_heatmap(heatmap=cl(hm), rot_cos=cl(pd["rot"][:, 0:1]), rot_sin=cl(pd["rot"][:, 1:2]), center=cl(pd["center"]),
                                          center_z=cl(pd["center_z"]), dim=cl(dim), vel=cl(pd["vel"]), iou=cl(iou), rectifier=0.5,
                                          point_cloud_range=pc_range, voxel_size=voxel, feature_map_stride=8, K=100, score_thresh=0.1,
                                          post_center_limit_range=limit)
        for bidx in range(hm.shape[0]):
            boxes, scores, labels = opost.decode_sample(hm[bidx], pd["center"][bidx], pd["center_z"][bidx], dim[bidx], pd["rot"][bidx], pd["vel"][bidx],
                                                        iou[bidx, 0], 100, 8, voxel, pc_range, 0.1, limit, 0.5)
            assert out[bidx]["pred_boxes"].shape == boxes.shape
            assert torch.equal(out[bidx]["pred_labels"].long(), labels)
            np.testing.assert_allclose(out[bidx]["pred_boxes"].numpy(), boxes.numpy(), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(out[bidx]["pred_scores"].numpy(), scores.numpy(), rtol=1e-6, atol=1e-7)


def test_config_set_overrides_and_logging():
    """pcdet/config.py:7-48 semantics: `--set` typing rules (scalar of the same type, `k:v,k:v` into a section, `a,b` into a list,
    unknown keys and type changes rejected) and the log format."""
    from radardistill_amd.pcdet.config import AttrDict, cfg_from_list, log_config_to_file
    c = AttrDict({"A": {"LR": 0.003, "STEPS": [35, 45], "SUB": {"X": 1, "NAME": "n"}}, "A2": {"A": 5}, "FLAG": True})
    cfg_from_list(["A.LR", "0.01", "A.STEPS", "10,20,30", "A.SUB", "X:7,NAME:q", "A2.A", "6", "FLAG", "False"], c)
    assert c.A.LR == 0.01 and c.A.STEPS == [10, 20, 30] and c.A.SUB.X == 7 and c.A.SUB.NAME == "q" and c.A2.A == 6 and c.FLAG is False
    with pytest.raises(AssertionError):
        cfg_from_list(["A.MISSING", "1"], c)
    with pytest.raises(AssertionError):
        cfg_from_list(["A.LR", "'text'"], c)
    with pytest.raises(AssertionError):
        cfg_from_list(["A.LR"], c)

    class Log:
        def __init__(self):
            self.lines = []

        def info(self, s):
            self.lines.append(s)

    lg = Log()
    log_config_to_file(c, logger=lg)
    assert lg.lines[0] == "----------- A -----------" and "cfg.A.LR: 0.01" in lg.lines and "cfg.A.SUB.X: 7" in lg.lines and "cfg.FLAG: False" in lg.lines
    assert "----------- SUB -----------" in lg.lines


def test_checkpoint_helpers_recall_counts_and_layout_fixups(tmp_path):
    """radardistill_amd/checkpoint.py: recall counting for all thresholds at once, trailing zero padding of gt_boxes, the `_optim` side
    file name, legacy sparse-kernel layouts (spconv 1.x (k, k, c_in, c_out) and its transposed variant) and name + shape fitting."""
    from radardistill_amd import checkpoint as CK
    iou = torch.tensor([[0.9, 0.2, 0.0], [0.4, 0.6, 0.0]])
    assert CK.count_recalled(iou, [0.3, 0.5, 0.7]) == [2, 2, 1] and CK.count_recalled(iou[:0], [0.3]) == [0]
    gt = torch.tensor([[1.0, 2, 3], [0, 0, 0], [4, 5, 6], [0, 0, 0], [0, 0, 0]])
    assert CK.strip_padding(gt).shape[0] == 3 and CK.strip_padding(gt[3:]).shape[0] == 0
    assert CK.optimizer_side_file("/a/b/checkpoint_epoch_3.pth") == "/a/b/checkpoint_epoch_3_optim.pth"
    with pytest.raises(ValueError):
        CK.optimizer_side_file("/a/b/ckpt.pt")
    with pytest.raises(FileNotFoundError):
        CK.read_checkpoint(str(tmp_path / "missing.pth"))

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.sp = torch.nn.Parameter(torch.zeros(8, 3, 3, 4))          # this build's sparse layout (c_out, k, k, c_in)
            self.lin = torch.nn.Linear(4, 2)
    net = Net()
    v1 = torch.arange(8 * 3 * 3 * 4, dtype=torch.float32).view(3, 3, 4, 8)          # spconv 1.x
    _, fitted = CK.fit_state_to_model(net, {"sp": v1, "lin.weight": torch.ones(2, 4), "lin.bias": torch.ones(3), "other": torch.ones(1)}, {"sp"})
    assert set(fitted) == {"sp", "lin.weight"} and torch.equal(fitted["sp"], v1.permute(3, 0, 1, 2))
    _, fitted = CK.fit_state_to_model(net, {"sp": torch.zeros(8, 3, 4, 3)}, {"sp"})          # last two axes swapped
    assert tuple(fitted["sp"].shape) == (8, 3, 3, 4)
    _, fitted = CK.fit_state_to_model(net, {"sp": v1}, set())          # not a sparse kernel: no re-layout is tried
    assert "sp" not in fitted
