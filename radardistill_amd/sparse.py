"""spconv-compatible sparse tensor / layer API on top of the HIP kernels.

Mirrors the subset of spconv 2.x that the reference uses (pcdet/utils/spconv_utils.py:3-38,
pcdet/models/backbones_3d/spconv_backbone_2d.py:9-28,41-77,264-299): `SparseConvTensor(features, indices,
spatial_shape, batch_size)` with `.features/.indices/.dense()/.replace_feature()`, `SubMConv2d`, `SparseConv2d`,
`SparseSequential`, `SparseModule`, and `conv.SparseConvolution` for the isinstance check of
find_all_spconv_keys.  Weight layout is spconv 2.x's [Cout, kh, kw, Cin].

Differences by design (MI355X-first): no hash table -- active sites live in a rank grid (bitmap + popcount prefix,
index.hip), output rows of SparseConv2d are in canonical (b, y, x) order, a convolution is ONE gathered implicit-GEMM
kernel over a neighbour table instead of 9 gather/GEMM/scatter-add rounds, and `dense()` returns channels-last memory.
"""
import types

import torch
import torch.nn as nn

from . import autograd as A
from . import kernels as K

# rank grids produced by our own voxeliser, keyed by the pillar_coords tensor they belong to
_RANKGRID_REGISTRY = {}


def register_rankgrid(coords, rankgrid, xmajor, level=None):
    """Remember the rank grid (and optionally the whole prebuilt `_Level` pyramid) that belongs to a pillar_coords tensor."""
    while len(_RANKGRID_REGISTRY) >= 4:          # teacher + student of the current step (+ the previous step's); bounded
        _RANKGRID_REGISTRY.pop(next(iter(_RANKGRID_REGISTRY)))
    _RANKGRID_REGISTRY[(coords.data_ptr(), tuple(coords.shape))] = (rankgrid, xmajor, coords, level)


class _Level:
    """Active-site set at one resolution: coords (n,3) int32 (b,y,x), rank grid, lazily built neighbour tables."""

    def __init__(self, coords, rankgrid, xmajor, batch, H, W):
        self.coords, self.rg, self.xmajor = coords, rankgrid, xmajor
        self.batch, self.H, self.W = batch, H, W
        self.n = coords.shape[0]
        self._subm = None
        self._down = None

    def subm_spec(self):
        if self._subm is None:
            nbr = K.nbr_subm(self.coords, self.rg, self.batch, self.H, self.W, self.xmajor)
            fwd = K.conv_index_table(nbr, flip=False)
            bwd = K.conv_index_table(nbr, flip=True)       # SubM: i is tap t of j  <=>  j is tap 8-t of i
            self._subm = A.ConvSpec(9, self.n, self.n, fwd, bwd, 0, keep=(nbr,), fwd_nbr=nbr, bwd_nbr=nbr)
        return self._subm

    def down_begin(self):
        """First half of down(): the output rank grid and its (device) active-cell count."""
        Ho, Wo = (self.H + 2 - 3) // 2 + 1, (self.W + 2 - 3) // 2 + 1
        rg_o = K.rankgrid_downsample(self.coords, self.batch, Ho, Wo)
        return rg_o, K.rankgrid_count_tensor(rg_o, self.batch * Ho * Wo)

    def down_finish(self, rg_o, n_out):
        Ho, Wo = (self.H + 2 - 3) // 2 + 1, (self.W + 2 - 3) // 2 + 1
        coords_o = K.rankgrid_coords(rg_o, self.batch, Ho, Wo, False, n_out)
        nbr = K.nbr_strided(coords_o, self.rg, self.batch, self.H, self.W, self.xmajor)
        nbrT = K.nbr_strided_T(self.coords, rg_o, self.batch, Ho, Wo)
        spec = A.ConvSpec(9, self.n, n_out, K.conv_index_table(nbr), K.conv_index_table(nbrT), 0, keep=(nbr, nbrT),
                          fwd_nbr=nbr, bwd_nbr=nbrT)
        self._down = (_Level(coords_o, rg_o, False, self.batch, Ho, Wo), spec)
        return self._down

    def down(self):
        """SparseConv2d(k3, s2, p1) output level + conv spec."""
        if self._down is None:
            rg_o, cnt = self.down_begin()
            self.down_finish(rg_o, int(cnt.item()))                                        # device -> host sync
        return self._down

    def tensors(self):
        """Every device tensor of this level and the levels below it (for record_stream when built on another stream)."""
        yield self.coords
        yield self.rg
        if self._subm is not None:
            yield self._subm.fwd_nbr
        if self._down is not None:
            lvl, spec = self._down
            yield spec.fwd_nbr
            yield spec.bwd_nbr
            yield from lvl.tensors()


def mark_pyramid(rg, xmajor, batch, H, W, n_down):
    """Rank grids of `n_down` stride-2 levels below the grid `rg`, device only (no host read): [(rg_l, Ho, Wo, count tensor), ...]."""
    out = []
    for _ in range(n_down):
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        rg_o = K.rankgrid_downsample_grid(rg, batch, H, W, xmajor, Ho, Wo)
        out.append((rg_o, Ho, Wo, K.rankgrid_count_tensor(rg_o, batch * Ho * Wo)))
        rg, xmajor, H, W = rg_o, False, Ho, Wo
    return out


def finish_pyramid(level, marked, counts):
    """Attach the levels marked by mark_pyramid (sizes now known on the host) below `level`, with all neighbour tables."""
    level.subm_spec()
    for (rg_o, _, _, _), n_out in zip(marked, counts):
        if level._down is None:
            level.down_finish(rg_o, int(n_out))
        level = level._down[0]
        level.subm_spec()


def pyramid_from_tables(rgs, dims, batch, coords, subm, down, up):
    """The `_Level` chain of one branch from the tables rd_geometry_finish filled (kernels.geometry_finish): level 0 is the x-major
    pillar grid, the levels below are y-major.  Equal, table for table, to finish_pyramid() on the same rank grids."""
    levels = [_Level(c, rg, l == 0, batch, H, W) for l, (c, rg, (H, W)) in enumerate(zip(coords, rgs, dims))]
    for lvl, nbr in zip(levels, subm):
        lvl._subm = A.ConvSpec(9, lvl.n, lvl.n, K.conv_index_table(nbr, flip=False), K.conv_index_table(nbr, flip=True), 0, keep=(nbr,),
                               fwd_nbr=nbr, bwd_nbr=nbr)
    for l, (nbr, nbrT) in enumerate(zip(down, up)):
        spec = A.ConvSpec(9, levels[l].n, levels[l + 1].n, K.conv_index_table(nbr), K.conv_index_table(nbrT), 0, keep=(nbr, nbrT),
                          fwd_nbr=nbr, bwd_nbr=nbrT)
        levels[l]._down = (levels[l + 1], spec)
    return levels[0]


def build_pyramids(levels, n_down):
    """SubM tables + `n_down` stride-2 levels below each of `levels`, in lockstep: ONE device->host read per depth for all branches."""
    cur = list(levels)
    for lvl in cur:
        lvl.subm_spec()
    for _ in range(n_down):
        begun = [lvl.down_begin() if lvl._down is None else None for lvl in cur]
        pending = [b[1] for b in begun if b is not None]
        counts = iter(torch.stack(pending).tolist()) if pending else iter(())
        nxt = []
        for lvl, b in zip(cur, begun):
            if b is not None:
                lvl.down_finish(b[0], int(next(counts)))
            nxt.append(lvl._down[0])
        cur = nxt
        for lvl in cur:
            lvl.subm_spec()


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, _level=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        if _level is None:
            H, W = self.spatial_shape
            if indices.dtype != torch.int32 or not indices.is_contiguous():
                indices = indices.int().contiguous()
                self.indices = indices
            hit = _RANKGRID_REGISTRY.get((indices.data_ptr(), tuple(indices.shape)))
            lvl = hit[3] if hit is not None else None
            if lvl is not None and (lvl.batch, lvl.H, lvl.W) == (self.batch_size, H, W):
                _level = lvl                          # pyramid prebuilt by the geometry prelude (detectors/pillarnet.py)
            else:
                if hit is not None:
                    rg, xmajor = hit[0], hit[1]
                else:
                    rg, xmajor = self._rankgrid_from_user_indices(indices, H, W)
                _level = _Level(indices, rg, xmajor, self.batch_size, H, W)
        self._level = _level

    def _rankgrid_from_user_indices(self, idx, H, W):
        """Caller-supplied indices: validate range and ordering on the device (one sync), then build the rank grid."""
        if idx.shape[0] == 0:
            return K.rankgrid_from_coords(idx, self.batch_size, H, W, False), False
        b, y, x = idx[:, 0].long(), idx[:, 1].long(), idx[:, 2].long()
        ok = ((b >= 0) & (b < self.batch_size) & (y >= 0) & (y < H) & (x >= 0) & (x < W)).all()
        k_yx = (b * H + y) * W + x
        k_xy = (b * W + x) * H + y
        s_yx = (k_yx[1:] > k_yx[:-1]).all()
        s_xy = (k_xy[1:] > k_xy[:-1]).all()
        ok, s_yx, s_xy = [bool(v) for v in torch.stack([ok, s_yx, s_xy]).tolist()]
        if not ok:
            raise RuntimeError("SparseConvTensor: indices outside the spatial shape / batch")
        if not (s_yx or s_xy):
            raise RuntimeError("SparseConvTensor: indices must be unique and sorted by (b,y,x) or (b,x,y) "
                               "(the dynamic-pillar VFE emits (b,x,y) order)")
        xmajor = not s_yx
        return K.rankgrid_from_coords(idx, self.batch_size, H, W, xmajor), xmajor

    def replace_feature(self, new_features):
        t = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, _level=self._level)
        t.indice_dict = self.indice_dict
        return t

    def dense(self, channels_first=True):
        H, W = self.spatial_shape
        rows = A.rows_to_dense(self.features, self.indices, self.batch_size, H, W)
        out = A.rows_to_nchw(rows, self.batch_size, H, W)
        return out if channels_first else out.permute(0, 2, 3, 1)

    @property
    def spatial_size(self):
        return self.spatial_shape[0] * self.spatial_shape[1]


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    """Base of SubMConv2d / SparseConv2d (name kept for find_all_spconv_keys, pcdet/utils/spconv_utils.py:23)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, bias=True, indice_key=None, subm=False):
        super().__init__()
        if kernel_size != 3 or (subm and stride != 1) or (not subm and (stride != 2 or padding != 1)):
            raise NotImplementedError("only SubMConv2d(k3) and SparseConv2d(k3, s2, p1) are on the RadarDistill path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.subm, self.indice_key = kernel_size, stride, padding, subm, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if self.bias is not None:
            bound = 1.0 / (self.in_channels * 9) ** 0.5
            nn.init.uniform_(self.bias, -bound, bound)

    def _spec_and_level(self, x):
        if self.subm:
            return x._level.subm_spec(), x._level
        lvl, spec = x._level.down()
        return spec, lvl

    def forward(self, x, stats=None):
        spec, lvl = self._spec_and_level(x)
        feats = A.conv(x.features, self.weight, self.bias, spec, self.out_channels, stats)
        out = SparseConvTensor(feats, lvl.coords, [lvl.H, lvl.W], x.batch_size, _level=lvl)
        out.indice_dict = x.indice_dict
        return out


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, 1, padding, bias, indice_key, subm=True)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, indice_key, subm=False)


class SparseSequential(SparseModule):
    """nn.Sequential over sparse tensors: plain nn.Modules (BatchNorm1d, ReLU) act on `.features` (spconv semantics)."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def __len__(self):
        return len(self._modules)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                x = x.replace_feature(apply_rowwise(m, x.features))
            else:
                x = m(x)
        return x


def apply_rowwise(m, feats):
    """BatchNorm1d / ReLU on (rows, C) features through the HIP kernels."""
    if isinstance(m, (nn.BatchNorm1d, nn.SyncBatchNorm)):
        if m.training:
            return A.bn_act_train(feats, m, None, act=0)
        return A.bn_act_eval(feats, m, None, act=0)
    if isinstance(m, nn.ReLU):
        return K.affine_act(feats, None, None, None, 1) if not feats.requires_grad else torch.relu(feats)
    return m(feats)


# spconv namespace shim: `from ...utils.spconv_utils import spconv` then `spconv.SubMConv2d`, `spconv.conv.SparseConvolution`
conv = types.SimpleNamespace(SparseConvolution=SparseConvolution)
