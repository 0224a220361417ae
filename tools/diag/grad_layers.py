"""Diagnostic (GPU box): gradient w.r.t. intermediate sparse features, HIP vs fp64 oracle, to localise a backward bug."""
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
pf = bd["radar_pillar_features"]
for k in ("x_conv1", "x_conv2", "x_conv3"):
    ms[k].features.retain_grad()
ms["x_conv4"].retain_grad(); pf.retain_grad()
g4 = torch.from_numpy(np.random.default_rng(1).normal(size=tuple(ms["x_conv4"].shape)).astype(np.float32))
g5 = torch.from_numpy(np.random.default_rng(2).normal(size=tuple(ms["x_conv5"].shape)).astype(np.float32))
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
loss = 0
if mode in ("both", "x4"):
    loss = loss + (ms["x_conv4"] * g4.to(DEV)).sum()
if mode in ("both", "x5"):
    loss = loss + (ms["x_conv5"] * g5.to(DEV)).sum()
loss.backward()

st = {("radar_vfe." + k): v.detach().cpu().clone() for k, v in vfe_m.state_dict().items()}
st.update({("radar_backbone_3d." + k): v.detach().cpu().clone() for k, v in bb.state_dict().items()})
seeded_fill_({k[len("radar_vfe."):]: v for k, v in st.items() if k.startswith("radar_vfe.")}, seed=31)
seeded_fill_({k[len("radar_backbone_3d."):]: v for k, v in st.items() if k.startswith("radar_backbone_3d.")}, seed=32)
ov = ovfe.dynamic_pillar_vfe(pts, st, "radar_vfe.", pc_range, voxel, gs, training=True)
feats = ov["pillar_features"].detach().double().requires_grad_(True)
sb = {k: (v.double() if v.is_floating_point() else v) for k, v in st.items() if k.startswith("radar_backbone_3d.")}
ob = osp.pillar_res18_backbone(feats, ov["pillar_coords"].numpy(), B, gs, sb, "radar_backbone_3d.", training=True)
for k in ("x_conv1", "x_conv2", "x_conv3"):
    ob[k][0].retain_grad()
ob["x_conv4"].retain_grad()
lo = 0
if mode in ("both", "x4"):
    lo = lo + (ob["x_conv4"] * g4.double()).sum()
if mode in ("both", "x5"):
    lo = lo + (ob["x_conv5"] * g5.double()).sum()
lo.backward()

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().double()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30), float((a - b).norm() / (b.norm() + 1e-30))

print("mode", mode)
print("fwd  x_conv3 feats", rel(ms["x_conv3"].features, ob["x_conv3"][0]))
print("grad x_conv4 (dense)", rel(ms["x_conv4"].grad, ob["x_conv4"].grad))
for k in ("x_conv3", "x_conv2", "x_conv1"):
    print("grad", k, rel(ms[k].features.grad, ob[k][0].grad), "rows", ms[k].features.shape[0])
print("grad pillar_features", rel(pf.grad, feats.grad))
# where in x_conv2 is the error? rows with largest error
d = (ms["x_conv2"].features.grad.detach().cpu().double() - ob["x_conv2"][0].grad).abs().max(1)[0]
top = torch.topk(d, 8)
print("x_conv2 worst rows", top.indices.tolist(), [f"{v:.3e}" for v in top.values.tolist()], "max|g|", float(ob["x_conv2"][0].grad.abs().max()))
print("coords of worst rows", ms["x_conv2"].indices[top.indices.to(DEV)].cpu().tolist())

# ReLU sign disagreements between the fp32 HIP forward and the fp64 oracle forward at the block outputs
for k in ("x_conv1", "x_conv2", "x_conv3"):
    a = ms[k].features.detach().cpu().double(); b = ob[k][0].detach()
    dis = (a > 0) != (b > 0)
    print(k, "mask disagreements", int(dis.sum()), "of", a.numel(), "values there: hip", a[dis][:5].tolist(), "oracle", b[dis][:5].tolist())
