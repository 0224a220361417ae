"""Diagnostic (GPU box): are the gradient differences HIP-vs-oracle(fp32) fp32 noise?  Compare both with an fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import sparse as osp, vfe as ovfe
from radardistill_amd.synthetic import make_batch
from tests.seeded import seeded_fill_
from tests.test_gpu_kernels import _vfe_module, _backbone, DEV

grid, B = 128, 2
vfe_m, pc_range, voxel, gs = _vfe_module("radar", 6, grid, seed=31)
bb = _backbone(True, grid, seed=32)
vfe_m.train(); bb.train()
batch = make_batch(batch_size=B, n_lidar=16, n_radar=1200, n_boxes=2, grid=grid, seed=11)
pts = torch.from_numpy(batch["radar_points"])
bd = bb(vfe_m({"radar_points": pts.to(DEV), "batch_size": B}))
ms = bd["radar_multi_scale_2d_features"]
g4 = torch.from_numpy(np.random.default_rng(1).normal(size=tuple(ms["x_conv4"].shape)).astype(np.float32))
g5 = torch.from_numpy(np.random.default_rng(2).normal(size=tuple(ms["x_conv5"].shape)).astype(np.float32))
((ms["x_conv4"] * g4.to(DEV)).sum() + (ms["x_conv5"] * g5.to(DEV)).sum()).backward()
named = dict(("radar_backbone_3d." + k, p) for k, p in bb.named_parameters())

def oracle(dtype):
    st = {("radar_vfe." + k): v.detach().cpu().clone() for k, v in vfe_m.state_dict().items()}
    st.update({("radar_backbone_3d." + k): v.detach().cpu().clone() for k, v in bb.state_dict().items()})
    seeded_fill_({k[len("radar_vfe."):]: v for k, v in st.items() if k.startswith("radar_vfe.")}, seed=31)
    seeded_fill_({k[len("radar_backbone_3d."):]: v for k, v in st.items() if k.startswith("radar_backbone_3d.")}, seed=32)
    ov = ovfe.dynamic_pillar_vfe(pts, st, "radar_vfe.", pc_range, voxel, gs, training=True)
    feats = ov["pillar_features"].detach().to(dtype)
    sb = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items() if k.startswith("radar_backbone_3d.")}
    params = [k for k in sb if sb[k].is_floating_point() and "running" not in k]
    for k in params:
        sb[k].requires_grad_(True)
    ob = osp.pillar_res18_backbone(feats, ov["pillar_coords"].numpy(), B, gs, sb, "radar_backbone_3d.", training=True)
    ((ob["x_conv4"] * g4.to(dtype)).sum() + (ob["x_conv5"] * g5.to(dtype)).sum()).backward()
    return {k: sb[k].grad for k in params}, ob

g32, ob32 = oracle(torch.float32)
g64, ob64 = oracle(torch.float64)
print("x_conv5 fwd: hip-vs-f64 %.3e   oracle32-vs-f64 %.3e" % (
    float((ms["x_conv5"].detach().cpu().double() - ob64["x_conv5"]).abs().max()), float((ob32["x_conv5"].double() - ob64["x_conv5"]).abs().max())))
gscale = max(float(v.abs().max()) for v in g64.values())
rows = []
for k in g64:
    ref = g64[k]
    den = float(ref.abs().max()) + 1e-4 * gscale
    e_hip = float((named[k].grad.detach().cpu().double() - ref).abs().max()) / den
    e_o32 = float((g32[k].double() - ref).abs().max()) / den
    e_pair = float((named[k].grad.detach().cpu().double() - g32[k].double()).abs().max()) / den
    rows.append((e_hip, e_o32, e_pair, k, den))
rows.sort(reverse=True)
print("err / (max|g| + 1e-4 gscale):  hip-vs-f64  oracle32-vs-f64  hip-vs-oracle32   param")
for e_hip, e_o32, e_pair, k, den in rows[:15]:
    print("  %.3e   %.3e   %.3e   %-50s den=%.3e" % (e_hip, e_o32, e_pair, k, den))
