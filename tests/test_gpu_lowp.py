"""BASELINE configs[2] (bf16) and configs[4] (fp8 MFMA DenseEnc): the low-precision path of csrc/lowp.hip.

fp32 is the library's parity mode; these modes state their own tolerances:
  * conversions: bit-exact against torch's own bfloat16 / float8_e4m3fn casts (round-to-nearest-even, saturating at +-448),
  * one convolution on ALREADY-QUANTISED operands: 1e-4 of the output range against an fp64 convolution of the same quantised
    values -- the kernel adds nothing but fp32 accumulation order (this pins indexing, the halo, the transposed-conv scatter and
    the fp8 K = 64 MFMA operand layout),
  * the whole DenseEnc against the fp32 kernels / fixture g2: relative L2 error <= 1e-2 (bf16), <= 6e-2 (fp8 e4m3, per-tensor
    activation and per-channel weight scales) -- 8 / 3 mantissa bits through 13 layers."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from radardistill_amd import lowp as LP                        # noqa: E402
from tests.seeded import seeded_fill_                          # noqa: E402
from tests.test_gpu_kernels import DEV                          # noqa: E402
from tests.test_gpu_model import BEV_CFG, _bev_inputs, _cl      # noqa: E402


def _deq(t, dtype):
    """narrow tensor (CUDA) -> float64 CPU values."""
    if dtype == LP.BF16:
        return t.cpu().double()
    return t.cpu().view(torch.float8_e4m3fn).double()


@pytest.mark.parametrize("dtype", [LP.BF16, LP.FP8])
def test_cast_matches_torch_narrow_types(dtype):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(257, 64, generator=g) * 3.0
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-9, 447.9, 448.0, 460.0, 1e6, -1e6])      # subnormal range and saturation
    x[1, :4] = torch.tensor([2.0 ** -9, 2.0 ** -10, 0.0625 + 2.0 ** -8, -17.0])
    mul = 0.73
    got = LP.lp_cast(x.to(DEV), dtype, mul)
    want = (x * mul)
    if dtype == LP.BF16:
        assert torch.equal(got.cpu(), want.to(torch.bfloat16))
    else:
        want = want.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(got.cpu(), want)
    back = LP.lp_uncast(got, dtype, 2.0)
    assert torch.equal(back.cpu(), (_deq(got, dtype) * 2.0).float())
    # concat by placement
    cat = torch.zeros((257, 192), dtype=got.dtype, device=DEV)
    LP.lp_cast(x.to(DEV), dtype, mul, out=cat, out_col0=128)
    assert torch.equal(cat[:, 128:], got) and int(cat[:, :128].view(torch.uint8).max()) == 0
    assert float(LP.lp_amax(x.to(DEV).view(-1))) == float(x.abs().max())


@pytest.mark.parametrize("dtype", [LP.BF16, LP.FP8])
@pytest.mark.parametrize("B,H,W,Cin,Cout,ksize,deconv", [
    (2, 21, 19, 64, 96, 3, False),        # ragged map, 8x8 tiles, Cout not a tile multiple
    (8, 64, 48, 256, 256, 3, False),      # 8x16 x 128 tiles
    (1, 8, 16, 64, 33, 3, False),
    (2, 13, 11, 128, 64, 1, False),       # 1x1
    (2, 9, 7, 64, 128, 2, True),          # ConvTranspose2d(k2, s2)
    (8, 32, 32, 256, 256, 2, True)])
def test_lp_conv_on_quantised_operands(dtype, B, H, W, Cin, Cout, ksize, deconv):
    g = np.random.default_rng(B * 100 + H + Cin)
    x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
    taps = 4 if deconv else ksize * ksize
    w = torch.from_numpy((g.normal(size=(Cout, taps, Cin)) / np.sqrt(taps * Cin)).astype(np.float32))
    alpha = torch.from_numpy(g.uniform(0.5, 1.5, size=Cout).astype(np.float32))
    beta = torch.from_numpy(g.normal(size=Cout).astype(np.float32) * 0.2)
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV)
    xq = LP.lp_cast(rows, dtype, 4.0)                                      # spread over the fp8 range
    wq, ws = LP.lp_quant_weights(w.to(DEV), dtype)
    out = LP.lp_conv(xq, dtype, B, H, W, Cin, wq, ksize, alpha.to(DEV), beta.to(DEV), True, LP.F32, Cout, deconv=deconv)
    xd = _deq(xq, dtype).view(B, H, W, Cin).permute(0, 3, 1, 2)
    wd = _deq(wq, dtype).view(Cout, taps, Cin)
    if deconv:
        ref = F.conv_transpose2d(xd, wd.view(Cout, 2, 2, Cin).permute(3, 0, 1, 2).contiguous(), stride=2)
    else:
        ref = F.conv2d(xd, wd.view(Cout, ksize, ksize, Cin).permute(0, 3, 1, 2).contiguous(), padding=ksize // 2)
    ref = torch.relu(ref * alpha.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1))
    ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    err = float((out.cpu().double() - ref_rows).abs().max())
    assert err <= 1e-4 * float(ref_rows.abs().max()) + 1e-6, err
    # narrow outputs: the same values (brought into the fp8 range by the epilogue scale) rounded once
    k = 200.0 / float(ref_rows.abs().max())
    for odt in (LP.BF16, LP.FP8):
        o2 = LP.lp_conv(xq, dtype, B, H, W, Cin, wq, ksize, (alpha * k).to(DEV), (beta * k).to(DEV), True, odt, Cout, deconv=deconv)
        tol = (2.0 ** -8 if odt == LP.BF16 else 2.0 ** -3) * 200.0
        assert float((_deq(o2, odt) - ref_rows * k).abs().max()) <= tol
    if dtype == LP.FP8:
        assert torch.allclose(ws.cpu(), w.view(Cout, -1).abs().max(1)[0] / 448.0, rtol=1e-6)


def _dense_enc(seed=12):
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_2d import __all__ as REG
    m = REG["BaseBEVBackboneV2"](AttrDict(BEV_CFG), input_channels=256)
    sd = m.state_dict(); seeded_fill_(sd, seed=seed); m.load_state_dict(sd)
    return m.to(DEV).eval()


def _rel_l2(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype,tol", [(LP.BF16, 1e-2), (LP.FP8, 6e-2)])
def test_lowp_dense_enc_vs_reference_fixture(golden_dir, dtype, tol):
    """The reference module's own eval outputs (fixture g2) vs the low-precision DenseEnc on the same seeded weights / inputs."""
    g = np.load(f"{golden_dir}/g2_dense_enc.npz")
    m = _dense_enc()
    x4, x5 = _bev_inputs(21)
    eng = LP.LowpDenseEnc(m, dtype)
    eng.calibrate(_cl(x4), _cl(x5))
    up, feat = eng.forward(_cl(x4), _cl(x5))
    e_up, e_feat = _rel_l2(up, g["eval_2d_8x"]), _rel_l2(feat, g["eval_2d"])
    print(f"dtype {dtype}: relative L2 vs the reference fixture: 2d_8x {e_up:.3e}  2d {e_feat:.3e}")
    assert e_up <= tol and e_feat <= tol


@pytest.mark.parametrize("B", [1, 8])
@pytest.mark.parametrize("dtype,tol", [(LP.BF16, 1e-2), (LP.FP8, 6e-2)])
def test_lowp_dense_enc_full_size_1024_bev(dtype, tol, B):
    """BASELINE configs[4] shapes: G = 1024 -> x_conv4 (B, 256, 128, 128), x_conv5 (B, 256, 64, 64), B in {1, 8}, against the fp32
    kernels of the same module (exact-fp32 MFMA)."""
    m = _dense_enc(seed=19)
    g = np.random.default_rng(50 + B)
    x4 = torch.from_numpy(g.normal(0, 1, size=(B, 256, 128, 128)).astype(np.float32)) * \
        torch.from_numpy((g.uniform(size=(B, 1, 128, 128)) < 0.4).astype(np.float32))
    x5 = torch.from_numpy(g.normal(0, 1, size=(B, 256, 64, 64)).astype(np.float32))
    x4d, x5d = _cl(x4), _cl(x5)
    with torch.no_grad():
        up_ref, feat_ref = m.dense_enc(x4d, x5d)
    eng = LP.LowpDenseEnc(m, dtype)
    eng.calibrate(x4d, x5d)
    up, feat = eng.forward(x4d, x5d)
    torch.cuda.synchronize()
    e_up, e_feat = _rel_l2(up, up_ref.cpu()), _rel_l2(feat, feat_ref.cpu())
    # timing (3 warm, 10 timed): the number that goes next to the roofline in profiles/
    for _ in range(3):
        eng.forward(x4d, x5d)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        eng.forward(x4d, x5d)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 10
    flops = 2.0 * 9 * 256 * 256 * B * (128 * 128 * 5 + 64 * 64 * 6) + 2.0 * 9 * 512 * 256 * B * 128 * 128 + 2.0 * 4 * 256 * 256 * B * 64 * 64
    print(f"dtype {dtype} B {B}: rel L2 2d_8x {e_up:.3e} 2d {e_feat:.3e}; DenseEnc forward {ms:.3f} ms = {flops / ms / 1e9:.1f} TF/s algorithmic")
    assert e_up <= tol and e_feat <= tol


def test_dual_branch_forward_and_afd_in_bf16():
    """BASELINE configs[2] at full size (35k LiDAR + 2k radar points, 512 x 512 BEV, B = 8): both branches forward with the frozen
    teacher's DenseEnc on bf16 storage (MODEL.BACKBONE_2D.PRECISION: bf16) and the AFD loss evaluated from bf16-stored maps
    (rd_afd_fwd_bf16), against the same forward in the fp32 parity mode.  Stated tolerance: teacher maps 1e-2 relative L2 (8
    mantissa bits through 13 layers, measured 3.5e-3), AFD feature / mask terms 1e-2 relative."""
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import load_data_to_gpu
    from radardistill_amd.synthetic import make_batch
    from tests.test_gpu_model import _build_pillarnet
    grid, B = 512, 8
    model, cfg, pc_range, voxel, gs = _build_pillarnet(grid)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    model = model.to(DEV).eval()
    batch = make_batch(batch_size=B, n_lidar=35000, n_radar=2000, n_boxes=30, grid=grid, seed=3)

    def forward(prec):
        model.backbone_2d.model_cfg['PRECISION'] = prec
        bd = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
        load_data_to_gpu(bd)
        with torch.no_grad():
            for m in model.module_list:
                if m.__class__.__name__.endswith('CenterHead'):
                    continue                                        # configs[2] stops at the feature maps + AFD
                if hasattr(m, 'prepare'):
                    m.prepare(bd)
                bd = m(bd)
        return bd

    ref = forward('fp32')
    got = forward('bf16')
    for k in ('spatial_features_2d', 'spatial_features_2d_8x'):
        e = _rel_l2(got[k], ref[k].cpu())
        print(k, "bf16 vs fp32 relative L2", e)
        assert e <= 1e-2, (k, e)
    # the STUDENT branch in bf16 arithmetic as well (round 3): train.autocast = every MFMA product on bf16-rounded operands (one of
    # bf16x3's three terms), fp32 accumulate and storage -- VFE, sparse encoder, CMA (DCNv2 columns), DenseEnc.  Stated tolerance:
    # 2e-2 relative L2 on the three student maps the losses read (8 mantissa bits through ~30 layers; measured 4-8e-3).
    from radardistill_amd.train import autocast
    with autocast(enabled=True):
        amp = forward('bf16')
    for k in ('radar_spatial_features_2d',):
        e = _rel_l2(amp[k], ref[k].cpu())
        print(k, "student bf16-product arithmetic vs fp32 relative L2", e)
        assert e <= 2e-2, (k, e)
    for k in ('radar_spatial_features_8x_2', 'radar_spatial_features_8x_1'):
        e = _rel_l2(amp['radar_multi_scale_2d_features'][k], ref['radar_multi_scale_2d_features'][k].cpu())
        print(k, "student bf16-product arithmetic vs fp32 relative L2", e)
        assert e <= 2e-2, (k, e)
    assert K.get_mfma_terms() == 3 and K.get_conv_math() == "f32"          # autocast restored the arithmetic mode
    # AFD: fp32 kernel on fp32 maps vs the bf16-map kernel on bf16 copies of the same three maps
    lid = ref['multi_scale_2d_features']['x_conv4']
    ra = ref['radar_multi_scale_2d_features']['radar_spatial_features_8x_2']
    rb = ref['radar_multi_scale_2d_features']['radar_spatial_features_8x_1']
    rows = [t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous() for t in (lid, ra, rb)]
    out32, _, _ = K.afd_fwd(rows[0], rows[1], rows[2], B)
    out16 = K.afd_fwd_bf16(*[LP.lp_cast(r, LP.BF16) for r in rows], B)
    print("AFD fp32", out32.tolist(), "bf16", out16.tolist())
    assert bool(torch.isfinite(out32).all())
    np.testing.assert_allclose(out16.cpu().numpy(), out32.cpu().numpy(), rtol=1e-2)
