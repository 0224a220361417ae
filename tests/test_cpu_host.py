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
    assert L.rd_abi_version() == 1
    assert L.rd_rankgrid_bytes(32 * 1024) == (2 * 1024 + 1 + 1) * 4


def test_missing_extension_fails_loudly(monkeypatch):
    from radardistill_amd import native
    monkeypatch.setattr(native, "SO_PATH", "/nonexistent/librdamd.so")
    monkeypatch.setattr(native, "_LIB", None)
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
                "radardistill_amd.pcdet.utils.spconv_utils", "radardistill_amd.selfcheck"]:
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
