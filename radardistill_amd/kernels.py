"""Thin Python callers of the C ABI (include/rdamd.h): torch is used only for device memory and streams.

Every function takes CUDA tensors, validates shapes/dtypes on the host (a kernel fault can reset the
whole GPU host), passes raw device pointers plus the current HIP stream, and raises RuntimeError on a
non-zero return code.  No function here has a CPU path.
"""
import ctypes
import os

import torch

from . import native
from .native import ConvIndex, check

f32, i32 = torch.float32, torch.int32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() costs ~9 us of Python per call
    (measured: 2.6 ms of a 30 ms step over ~300 launches in the forward alone); the two C entry points below cost ~0.3 us."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    """Device pointer as a plain int (None = NULL): every entry point has argtypes (native.SIGNATURES), so ctypes converts it to a
    64-bit void* itself -- building a c_void_p object per argument cost ~0.4 ms per step over ~1600 pointers."""
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name, dim=None):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected contiguous CUDA {dtype} tensor, got device={t.device} dtype={t.dtype} "
                           f"contiguous={t.is_contiguous()}")
    if dim is not None and t.dim() != dim:
        raise RuntimeError(f"{name}: expected {dim}-d tensor, got shape {tuple(t.shape)}")
    return t


# ------------------------------------------------------------------------------------------ index structures
def rankgrid_alloc(n_cells, device):
    nbytes = native.lib().rd_rankgrid_bytes(int(n_cells))
    return torch.empty(nbytes // 4, dtype=i32, device=device)


def rankgrid_count_tensor(rg, n_cells):
    """0-d int32 view of the active-cell count word (device)."""
    n_words = (int(n_cells) + 31) // 32
    return rg[2 * n_words]


def voxelize(points, batch, gx, gy, x0, y0, vx, vy):
    _chk(points, f32, "points", 2)
    n, nf = points.shape[0], points.shape[1] - 1
    rg = rankgrid_alloc(batch * gx * gy, points.device)
    point_row = torch.empty(n, dtype=i32, device=points.device)
    check(native.lib().rd_voxelize(_p(points), n, nf, batch, gx, gy, x0, y0, vx, vy, _p(rg), _p(point_row), _stream()), "rd_voxelize")
    return rg, point_row


def rankgrid_coords(rg, batch, H, W, xmajor, n_rows):
    coords = torch.empty((n_rows, 3), dtype=i32, device=rg.device)
    check(native.lib().rd_rankgrid_coords(_p(rg), batch, H, W, int(xmajor), _p(coords), n_rows, _stream()), "rd_rankgrid_coords")
    return coords


def rankgrid_from_coords(coords, batch, H, W, xmajor):
    _chk(coords, i32, "coords", 2)
    rg = rankgrid_alloc(batch * H * W, coords.device)
    check(native.lib().rd_rankgrid_from_coords(_p(coords), coords.shape[0], batch, H, W, int(xmajor), _p(rg), _stream()), "rd_rankgrid_from_coords")
    return rg


def rankgrid_downsample(in_coords, batch, Ho, Wo):
    _chk(in_coords, i32, "in_coords", 2)
    rg = rankgrid_alloc(batch * Ho * Wo, in_coords.device)
    check(native.lib().rd_rankgrid_downsample(_p(in_coords), in_coords.shape[0], batch, Ho, Wo, _p(rg), _stream()), "rd_rankgrid_downsample")
    return rg


def rankgrid_downsample_grid(in_rg, batch, H, W, in_xmajor, Ho, Wo):
    """SparseConv2d(k3, s2, p1) output rank grid from the INPUT rank grid (no coordinate list / row count needed)."""
    _chk(in_rg, i32, "in_rankgrid", 1)
    if in_rg.numel() * 4 != native.lib().rd_rankgrid_bytes(int(batch * H * W)):
        raise RuntimeError("rankgrid_downsample_grid: input rank grid does not match (batch, H, W)")
    rg = rankgrid_alloc(batch * Ho * Wo, in_rg.device)
    check(native.lib().rd_rankgrid_downsample_grid(_p(in_rg), batch, H, W, int(in_xmajor), Ho, Wo, _p(rg), _stream()), "rd_rankgrid_downsample_grid")
    return rg


def nbr_subm(coords, rg, batch, H, W, xmajor):
    n = coords.shape[0]
    nbr = torch.empty((n, 9), dtype=i32, device=coords.device)
    check(native.lib().rd_nbr_subm(_p(coords), n, _p(rg), batch, H, W, int(xmajor), _p(nbr), _stream()), "rd_nbr_subm")
    return nbr


def nbr_strided(out_coords, in_rg, batch, H, W, in_xmajor):
    n = out_coords.shape[0]
    nbr = torch.empty((n, 9), dtype=i32, device=out_coords.device)
    check(native.lib().rd_nbr_strided(_p(out_coords), n, _p(in_rg), batch, H, W, int(in_xmajor), _p(nbr), _stream()), "rd_nbr_strided")
    return nbr


def nbr_strided_T(in_coords, out_rg, batch, Ho, Wo):
    n = in_coords.shape[0]
    nbrT = torch.empty((n, 9), dtype=i32, device=in_coords.device)
    check(native.lib().rd_nbr_strided_T(_p(in_coords), n, _p(out_rg), batch, Ho, Wo, _p(nbrT), _stream()), "rd_nbr_strided_T")
    return nbrT


# ---- geometry prelude: one call before and one after the step's only device->host read (composite.hip)
_GEOMETRY_PLANS = {}


def _align(n, a=64):
    return (n + a - 1) // a * a


def geometry_plan(batch, gx, gy, n_down):
    """Static part of a branch's index allocation: [(H, W)] per level and the int32 offsets / sizes of its rank grids."""
    key = (batch, gx, gy, n_down)
    plan = _GEOMETRY_PLANS.get(key)
    if plan is None:
        dims = [(gy, gx)]
        for _ in range(n_down):
            H, W = dims[-1]
            dims.append(((H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1))
        sizes = [native.lib().rd_rankgrid_bytes(int(batch * H * W)) // 4 for H, W in dims]
        offs, o = [], 0
        for sz in sizes:
            offs.append(o)
            o += _align(sz)
        plan = _GEOMETRY_PLANS[key] = (dims, sizes, offs, o)
    return plan


def geometry_begin(points, batch, gx, gy, x0, y0, vx, vy, n_down, scalars):
    """rd_geometry_begin: points -> (rank grids of level 0 .. n_down, point_row), all views of ONE allocation; `scalars` (int32 device,
    2 + n_down) receives {pillars, in-range points, rows of each level below}."""
    _chk(points, f32, "points", 2)
    n, nf = points.shape[0], points.shape[1] - 1
    dims, sizes, offs, total = geometry_plan(batch, gx, gy, n_down)
    block = torch.empty(total + n, dtype=i32, device=points.device)
    rgs = [block[o:o + sz] for o, sz in zip(offs, sizes)]
    point_row = block[total:total + n]
    base = block.data_ptr()
    down = (ctypes.c_void_p * max(n_down, 1))(*[base + 4 * o for o in offs[1:]])
    check(native.lib().rd_geometry_begin(_p(points), n, nf, batch, gx, gy, x0, y0, vx, vy, base, base + 4 * total, n_down, down, _p(scalars),
                                         _stream()), "rd_geometry_begin")
    return rgs, point_row, dims


def geometry_finish(rgs, batch, gx, gy, rows):
    """rd_geometry_finish: with the rows of every level known -> (coords, SubM tables, strided tables, transposed strided tables),
    views of ONE allocation."""
    L = len(rgs)
    rows = [int(r) for r in rows]
    offs, o = [], 0
    for l in range(L):
        lv = [o, o + _align(3 * rows[l])]                         # coords, SubM table
        o = lv[1] + _align(9 * rows[l])
        if l + 1 < L:
            lv += [o, o + _align(9 * rows[l + 1])]                # strided table (rows of level l + 1), its transpose (rows of level l)
            o = lv[3] + _align(9 * rows[l])
        offs.append(lv)
    block = torch.empty(max(o, 1), dtype=i32, device=rgs[0].device)
    base = block.data_ptr()
    PA = ctypes.c_void_p * L
    coords = [block[lv[0]:lv[0] + 3 * r].view(r, 3) for lv, r in zip(offs, rows)]
    subm = [block[lv[1]:lv[1] + 9 * r].view(r, 9) for lv, r in zip(offs, rows)]
    down = [block[offs[l][2]:offs[l][2] + 9 * rows[l + 1]].view(rows[l + 1], 9) for l in range(L - 1)]
    up = [block[offs[l][3]:offs[l][3] + 9 * rows[l]].view(rows[l], 9) for l in range(L - 1)]
    check(native.lib().rd_geometry_finish(PA(*[rg.data_ptr() for rg in rgs]), batch, gx, gy, L - 1, (ctypes.c_int32 * L)(*rows),
                                          PA(*[base + 4 * lv[0] for lv in offs]), PA(*[base + 4 * lv[1] for lv in offs]),
                                          PA(*([base + 4 * offs[l][2] for l in range(L - 1)] + [0])),
                                          PA(*([base + 4 * offs[l][3] for l in range(L - 1)] + [0])), _stream()), "rd_geometry_finish")
    return coords, subm, down, up


# ------------------------------------------------------------------------------------------ pillar VFE
def vfe_pillar_mean(points, point_row, n_pillars):
    acc = torch.empty((n_pillars, 4), dtype=f32, device=points.device)
    check(native.lib().rd_vfe_pillar_mean(_p(points), points.shape[0], points.shape[1] - 1, _p(point_row), n_pillars, _p(acc), _stream()),
          "rd_vfe_pillar_mean")
    return acc


def vfe_linear_stats(points, point_row, coords, acc, weight, geom):
    stats = torch.empty(65, dtype=f32, device=points.device)
    _chk(weight, f32, "vfe weight", 2)
    if weight.shape != (32, 9 + points.shape[1] - 1):
        raise RuntimeError(f"vfe weight shape {tuple(weight.shape)} != (32, {9 + points.shape[1] - 1})")
    check(native.lib().rd_vfe_linear_stats(_p(points), points.shape[0], points.shape[1] - 1, _p(point_row), _p(coords), _p(acc), _p(weight),
                                           _p(geom), _p(stats), _stream()), "rd_vfe_linear_stats")
    return stats


def vfe_linear_bn_relu_max(points, point_row, coords, acc, weight, geom, scale, shift, n_pillars, want_argmax):
    if weight.shape != (32, 9 + points.shape[1] - 1):
        raise RuntimeError(f"vfe weight shape {tuple(weight.shape)} != (32, {9 + points.shape[1] - 1})")
    out = torch.empty((n_pillars, 32), dtype=f32, device=points.device)
    argmax = torch.empty((n_pillars, 32), dtype=i32, device=points.device) if want_argmax else None
    ws = torch.empty((n_pillars, 32), dtype=torch.int64, device=points.device)
    check(native.lib().rd_vfe_linear_bn_relu_max(_p(points), points.shape[0], points.shape[1] - 1, _p(point_row), _p(coords), _p(acc),
                                                 _p(_chk(weight, f32, "w")), _p(geom), _p(_chk(scale, f32, "scale")), _p(_chk(shift, f32, "shift")),
                                                 n_pillars, _p(out), _p(argmax), _p(ws), _stream()), "rd_vfe_linear_bn_relu_max")
    return out, argmax


def vfe_group(point_row, n_pillars):
    """Group the in-range points by pillar -> (offsets (P + 1,), order (n_points,)) int32 (vfe_seg.hip)."""
    _chk(point_row, i32, "point_row", 1)
    n = point_row.shape[0]
    dev = point_row.device
    offsets = torch.empty(n_pillars + 1, dtype=i32, device=dev)
    order = torch.empty(max(n, 1), dtype=i32, device=dev)
    nb = native.lib().rd_vfe_group_ws_bytes(int(n_pillars))
    ws = torch.empty(nb // 4, dtype=i32, device=dev)
    check(native.lib().rd_vfe_group(_p(point_row), n, int(n_pillars), _p(offsets), _p(order), _p(ws), nb, _stream()), "rd_vfe_group")
    return offsets, order


def _vfe_seg_chk(points, coords, weight, n_pillars):
    _chk(points, f32, "points", 2); _chk(coords, i32, "coords", 2); _chk(weight, f32, "vfe weight", 2)
    if weight.shape != (32, 9 + points.shape[1] - 1) or coords.shape != (n_pillars, 3):
        raise RuntimeError(f"vfe: weight {tuple(weight.shape)} / coords {tuple(coords.shape)} do not match {points.shape[1] - 1} point features, {n_pillars} pillars")


def vfe_seg_stats(points, order, offsets, coords, weight, geom, n_pillars):
    _vfe_seg_chk(points, coords, weight, n_pillars)
    stats = torch.empty(65, dtype=f32, device=points.device)
    check(native.lib().rd_vfe_seg_stats(_p(points), points.shape[1] - 1, _p(order), _p(offsets), _p(coords), _p(weight), _p(geom), int(n_pillars),
                                        _p(stats), _stream()), "rd_vfe_seg_stats")
    return stats


def vfe_seg_max(points, order, offsets, coords, weight, geom, scale, shift, n_pillars, want_argmax):
    """-> (out (P, 32), argmax (P, 32) or None, acc (P, 4) sum xyz + count)."""
    _vfe_seg_chk(points, coords, weight, n_pillars)
    dev = points.device
    out = torch.empty((n_pillars, 32), dtype=f32, device=dev)
    argmax = torch.empty((n_pillars, 32), dtype=i32, device=dev) if want_argmax else None
    acc = torch.empty((n_pillars, 4), dtype=f32, device=dev)
    check(native.lib().rd_vfe_seg_max(_p(points), points.shape[1] - 1, _p(order), _p(offsets), _p(coords), _p(weight), _p(geom),
                                      _p(_chk(scale, f32, "scale")), _p(_chk(shift, f32, "shift")), int(n_pillars), _p(out), _p(argmax), _p(acc),
                                      _stream()), "rd_vfe_seg_max")
    return out, argmax, acc


def vfe_backward(points, point_row, coords, acc, weight, geom, mean, rstd, gamma, beta, grad_out, argmax, n_valid, sync=None):
    """sync = (allreduce, count_dev) as in bn_bwd (SyncBatchNorm): two launches with the group's all-reduce between them."""
    n, nf = points.shape[0], points.shape[1] - 1
    P = grad_out.shape[0]
    gw = torch.empty_like(weight)
    g2 = torch.empty(64, dtype=f32, device=points.device)
    gg, gb = g2[:32], g2[32:]
    ws = torch.empty(max(n, 1) * 32 + 128, dtype=f32, device=points.device)
    if sync is not None:
        allreduce, count_dev = sync
        check(native.lib().rd_vfe_backward_reduce(_p(points), n, nf, _p(point_row), _p(coords), _p(acc), _p(weight), _p(geom), _p(mean), _p(rstd),
                                                  _p(gamma), _p(beta), _p(_chk(grad_out, f32, "grad_out", 2)), _p(argmax), P, _p(gg), _p(gb),
                                                  _p(ws), _stream()), "rd_vfe_backward_reduce")
        tot = g2.clone()
        allreduce(tot)
        check(native.lib().rd_vfe_backward_weight(_p(points), n, nf, _p(point_row), _p(coords), _p(acc), _p(weight), _p(geom), _p(mean), _p(rstd),
                                                  _p(gamma), _p(tot[:32]), _p(tot[32:]), _p(count_dev), P, _p(gw), _p(ws), _stream()),
              "rd_vfe_backward_weight")
        return gw, gg, gb
    check(native.lib().rd_vfe_backward(_p(points), n, nf, _p(point_row), _p(coords), _p(acc), _p(weight), _p(geom), _p(mean), _p(rstd),
                                       _p(gamma), _p(beta), _p(_chk(grad_out, f32, "grad_out", 2)), _p(argmax), P, int(n_valid),
                                       _p(gw), _p(gg), _p(gb), _p(ws), _stream()), "rd_vfe_backward")
    return gw, gg, gb


# ------------------------------------------------------------------------------------------ convolution
def conv_index_table(nbr, flip=False):
    ix = ConvIndex()
    ix.mode = 0
    ix.nbr = nbr.data_ptr() if nbr is not None and nbr.numel() else None
    ix.flip = int(flip)
    return ix


def conv_index_dense(B, Hin, Win, Hout, Wout, KH, KW, stride, pad, transposed=False):
    ix = ConvIndex()
    ix.mode = 2 if transposed else 1
    ix.nbr = None
    ix.B, ix.Hin, ix.Win, ix.Hout, ix.Wout = B, Hin, Win, Hout, Wout
    ix.KH, ix.KW, ix.stride, ix.pad, ix.flip = KH, KW, stride, pad, 0
    return ix


# Timing events for the hooks below come from a pool that bench.py fills (and records once, which is when HIP creates the handle)
# before the timed region: ~400 launches per step are bracketed, and creating their events inside the region cost ~0.5 ms per step.
_EVENT_POOL = []


def prefill_event_pool(n):
    while len(_EVENT_POOL) < n:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _EVENT_POOL.append(e)


def timing_event():
    if _EVENT_POOL:
        return _EVENT_POOL.pop()
    e = torch.cuda.Event(enable_timing=True)
    e.record()          # creates the HIP handle (the composite entry points record through the raw handle)
    return e


def _ev(e):
    """Raw hipEvent_t of a torch event as an int (0 / None = no event)."""
    h = e.cuda_event
    return h if isinstance(h, int) else getattr(h, "value", None)


# Optional per-launch timing hook used by bench.py for the roofline of the dominant kernel (k_conv_igemm<128,2,2,false>):
# a list to which (start_event, end_event, algorithmic_flops) of every such launch is appended.  None = off (no overhead).
CONV_PROFILE = None
PROFILE_ALL = os.environ.get("RD_BENCH_SHAPES") == "all"          # diagnostic: also time the narrow (<= 64-channel) launches
WGRAD_D3_MAX_CUS = int(os.environ.get("RD_WGRAD_D3_RES", "80"))          # mirrors launch_wgrad_d3_b3 (conv_wgrad_d3.hip): CUs its one round may occupy
PROFILE_TAGS = None          # None: every profiled launch gets its pair of events; a set: only launches of these instantiation tags


def split_bf16(x):
    """fp32 tensor -> the same bytes in split format ([4 x bf16 hi | 4 x bf16 lo] per 4 elements; kept in an fp32-typed tensor)."""
    _chk(x, f32, "split input")
    if x.numel() % 4:
        raise RuntimeError("split_bf16: element count must be a multiple of 4")
    out = torch.empty_like(x)
    check(native.lib().rd_split_bf16(_p(x), x.numel(), _p(out), _stream()), "rd_split_bf16")
    return out


def weight_layout_split(src, Cout, Cin, taps, kind, flip=False, out=None, frag=False):
    """Like weight_layout (operand kinds 0..3, 7, 8) with the destination written in split format (into `out` if given).
    frag (or kind | LAYOUT_FRAG): fragment-major split format (include/rdamd.h, RD_LAYOUT_FRAG)."""
    _chk(src, f32, "weight")
    if frag:
        kind |= LAYOUT_FRAG
    if src.numel() != Cout * Cin * taps:
        raise RuntimeError("weight_layout_split: element count mismatch")
    if kind & LAYOUT_FRAG:
        a_, b_ = (Cout, Cin) if (kind & ~LAYOUT_FRAG) in (0, 1, 3) else (Cin, Cout)
        if a_ % 32 or b_ % 16:
            raise RuntimeError(f"weight_layout_split: fragment-major needs slow axis % 32 == 0 and K axis % 16 == 0, got {a_} x {b_}")
    if out is None:
        out = torch.empty(Cout * Cin * taps, dtype=f32, device=src.device)
    elif _chk(out, f32, "split destination").numel() != Cout * Cin * taps:
        raise RuntimeError("weight_layout_split: destination size mismatch")
    check(native.lib().rd_weight_layout_split(_p(src), _p(out), Cout, Cin, taps, kind, int(flip), _stream()), "rd_weight_layout_split")
    return out


def weight_layout_split_multi(jobs_dev, chunk_job_dev, chunk_group_dev, n_chunks):
    """One launch for a table of rd_layout_job entries (device copies of the job array and the chunk tables; autograd._OperandCache)."""
    if not (jobs_dev.is_cuda and jobs_dev.dtype == torch.uint8 and jobs_dev.is_contiguous()):
        raise RuntimeError("weight_layout_split_multi: job table must be a contiguous CUDA uint8 tensor")
    _chk(chunk_job_dev, i32, "chunk_job", 1); _chk(chunk_group_dev, i32, "chunk_group", 1)
    if chunk_job_dev.numel() != n_chunks or chunk_group_dev.numel() != n_chunks:
        raise RuntimeError("weight_layout_split_multi: chunk tables must have n_chunks entries")
    check(native.lib().rd_weight_layout_split_multi(_p(jobs_dev), _p(chunk_job_dev), _p(chunk_group_dev), int(n_chunks), _stream()),
          "rd_weight_layout_split_multi")


SPARSEF = os.environ.get("RD_SPARSEF", "1") != "0"  # A/B switch: sparse (neighbour-table) convolutions on k_gemm_b3f<.., TABLE>
GEMMF = os.environ.get("RD_GEMMF", "1") != "0"      # A/B switch: 1-tap GEMMs on the fragment-major-weights kernel (conv_gemmf.hip)
D3F = os.environ.get("RD_D3F", "1") != "0"          # A/B switch: dense 3x3 convolutions on the fragment-major-weights kernel (conv_d3f.hip)
LAYOUT_FRAG = 16                                     # RD_LAYOUT_FRAG of include/rdamd.h


def wants_frag_weights(ix, in_rows, out_rows, Cin, Cout, taps):
    """True when rd_conv_fwd_split should get this convolution's weights in FRAGMENT-MAJOR split format (w_split = 2): bf16x3 mode, dense
    stride-1 3x3 geometry on same-size maps (forward: mode 1, data gradient: mode 2), Cin % 32 == 0, Cout % 32 == 0, and a map large
    enough for 8 x 16-pixel tiles to give every CU a workgroup (smaller maps keep the LDS-staged kernel's 8 x 8 tiles).  Mirrors
    conv_d3f_applies / launch_conv_d3f_b3 (conv_d3f.hip)."""
    if (GEMMF and taps == 1 and get_conv_math() == "bf16x3" and ix.mode in (1, 2) and ix.KH == 1 and ix.KW == 1 and ix.stride == 1 and ix.pad == 0
            and ix.Hin == ix.Hout and ix.Win == ix.Wout and in_rows == out_rows and Cin % 64 == 0 and Cout % 32 == 0 and Cout >= 64 and out_rows > 0):
        return True          # 1-tap GEMM (nn.Linear, 1x1 convolutions, the DCNv2 column GEMM, forward or data gradient): k_gemm_b3f (conv_gemmf.hip)
    if ix.mode == 0:          # neighbour table (SubMConv2d / SparseConv2d, forward or data gradient): the gathered form of k_gemm_b3f
        return (SPARSEF and get_conv_math() == "bf16x3" and 1 <= taps <= 9 and ix.nbr and Cin % 64 == 0 and Cout % 32 == 0 and Cout >= 64 and out_rows > 0
                and max(in_rows, 1) * Cin < 2 ** 31)
    if not (D3F and get_conv_math() == "bf16x3" and ix.mode in (1, 2) and taps == 9 and ix.KH == 3 and ix.KW == 3 and ix.stride == 1 and ix.pad == 1
            and ix.Hin == ix.Hout and ix.Win == ix.Wout and Cin % 32 == 0 and Cout % 32 == 0 and Cout >= 64 and in_rows == out_rows):
        return False
    hw = ix.Hout * ix.Wout
    if hw <= 0 or out_rows % hw:
        return False
    big_rows = (out_rows // hw) * ((ix.Hout + 7) // 8) * ((ix.Wout + 15) // 16)
    return big_rows * ((Cout + 63) // 64) >= 256


def _kernel_tag(ix, in_rows, out_rows, Cin, Cout, taps, in_split, tile, w_split=0):
    """Which instantiation rd_conv_fwd launches (mirrors the dispatch in conv.hip / conv_b3.hip; used by bench.py's roofline only):
    128 / 64 = gathered implicit-GEMM tile, "d3_128" / "d3_16x64" / "d3_64" = halo-staged dense 3x3 kernel (bf16x3 mode; pre-split weights assumed for the middle one)."""
    if w_split == 2 and ix.mode == 0:          # k_gemm_b3f<.., TABLE> (launch_gemm_b3f)
        return "sparsef_128" if ((out_rows + 127) // 128) * ((Cout + 127) // 128) >= 384 else "sparsef_64"
    if w_split == 2 and taps == 1:          # k_gemm_b3f (launch_gemm_b3f)
        return "gemmf_128" if ((out_rows + 127) // 128) * ((Cout + 127) // 128) >= 256 else "gemmf_64"
    if w_split == 2:          # fragment-major weights: k_conv_d3f_b3 (launch_conv_d3f_b3)
        big_rows = (out_rows // (ix.Hout * ix.Wout)) * ((ix.Hout + 7) // 8) * ((ix.Wout + 15) // 16)
        return "d3f_128" if (Cout >= 128 and big_rows * ((Cout + 127) // 128) >= 384) else "d3f_64"
    if (get_conv_math() == "bf16x3" and ix.mode in (1, 2) and taps == 9 and ix.KH == 3 and ix.KW == 3 and ix.stride == 1 and ix.pad == 1
            and ix.Hin == ix.Hout and ix.Win == ix.Wout and Cin % 32 == 0 and not in_split and in_rows == out_rows
            and os.environ.get("RD_D3", "1") != "0"):
        nb = out_rows // (ix.Hout * ix.Wout)
        big_rows = nb * ((ix.Hout + 7) // 8) * ((ix.Wout + 15) // 16)
        if big_rows * ((Cout + 127) // 128) >= 384:
            return "d3_128"
        small = os.environ.get("RD_D3_SMALL")
        wide = small == "1" if small is not None else big_rows * ((Cout + 63) // 64) >= 256
        return "d3_16x64" if wide else "d3_64"
    if (((tile == 64 and Cout >= 128) or Cout == 128) and ix.mode == 0 or (ix.mode == 2 and ix.stride == 2 and ix.Wout % 64 == 0 and Cout >= 128
                                                                      and os.environ.get("RD_TILE_TLINE", "1") != "0")) \
            and get_conv_math() == "bf16x3" and os.environ.get("RD_TILE_MID", "1") != "0":
        tile = "64x128"          # sparse mid-size layers and one-line transposed tiles: 64 rows x 128 channels
    elif Cout <= 64 and tile == 128 and ix.mode != 3 and get_conv_math() == "bf16x3" and os.environ.get("RD_TILE_NARROW", "1") != "0":
        tile = "128x64"
    if get_conv_math() == "bf16x3" and Cout > 32 and ix.mode != 3:
        # the geometry-specialised instantiations are different kernels (k_conv_igemm_b3<BM, BN, false, false, 2 | 3>): tag them apart
        return f"{tile}_{'table' if ix.mode == 0 else 'dense'}"
    return tile


def conv_fwd(x, weight_k, taps, bias, out_rows, Cout, ix, scale=None, shift=None, residual=None, relu=False, stats=None, nbr_keepalive=None,
             in_split=False, w_split=False):
    """x (in_rows, Cin); weight_k (Cout, taps, Cin) kernel layout -> (out_rows, Cout).  in_split / w_split: that operand is already in
    split format (bf16x3 mode); w_split = 2: the weights are in FRAGMENT-MAJOR split format (weight_layout_split(..., frag=True);
    only where wants_frag_weights() says so)."""
    _chk(x, f32, "conv input", 2)
    _chk(weight_k, f32, "conv weight")
    in_rows, Cin = x.shape
    if weight_k.numel() != Cout * taps * Cin:
        raise RuntimeError(f"conv weight has {weight_k.numel()} elements, expected {Cout}*{taps}*{Cin}")
    if ix.mode == 0 and nbr_keepalive is not None:
        if nbr_keepalive.shape != (out_rows, taps) or nbr_keepalive.dtype != i32 or not nbr_keepalive.is_contiguous():
            raise RuntimeError(f"neighbour table shape {tuple(nbr_keepalive.shape)} != ({out_rows}, {taps})")
    # (this wrapper runs ~165 times per step: the optional operands are checked inline, not through a loop over tuples)
    if bias is not None and _chk(bias, f32, "bias").numel() != Cout:
        raise RuntimeError(f"conv bias: {bias.numel()} elements, expected {Cout}")
    if scale is not None and _chk(scale, f32, "scale").numel() != Cout:
        raise RuntimeError(f"conv scale: {scale.numel()} elements, expected {Cout}")
    if shift is not None and _chk(shift, f32, "shift").numel() != Cout:
        raise RuntimeError(f"conv shift: {shift.numel()} elements, expected {Cout}")
    if stats is not None and _chk(stats, f32, "stats").numel() != 2 * Cout:
        raise RuntimeError(f"conv stats: {stats.numel()} elements, expected {2 * Cout}")
    if residual is not None and (_chk(residual, f32, "residual").shape != (out_rows, Cout)):
        raise RuntimeError("conv residual shape mismatch")
    out = torch.empty((out_rows, Cout), dtype=f32, device=x.device)
    # every launch of the Cout > 64, non-deform instantiations; `tile` mirrors the selection rule of rd_conv_fwd (conv.hip)
    prof = CONV_PROFILE is not None and (Cout > 64 or PROFILE_ALL) and ix.mode != 3
    if prof:
        tile = 128 if ((out_rows + 127) // 128) * ((Cout + 127) // 128) >= 384 else 64
        if taps == 1 and get_conv_math() == "bf16x3" and os.environ.get("RD_GEMM_TILE64", "1") != "0":
            tile = 64          # 1-tap layers: 64x64 tiles whatever the size (launch_conv_b3)
        tile = _kernel_tag(ix, in_rows, out_rows, Cin, Cout, taps, in_split, tile, int(w_split))
        prof = PROFILE_TAGS is None or tile in PROFILE_TAGS
    if prof:
        e0 = timing_event(); e1 = timing_event()
        e0.record()
    if in_split or w_split:
        check(native.lib().rd_conv_fwd_split(_p(x), int(in_split), in_rows, Cin, _p(weight_k), int(w_split), taps, _p(bias), _p(out), out_rows, Cout,
                                             ix, _p(scale), _p(shift), _p(residual), int(relu), _p(stats), _stream()),
              "rd_conv_fwd_split")
    else:
        check(native.lib().rd_conv_fwd(_p(x), in_rows, Cin, _p(weight_k), taps, _p(bias), _p(out), out_rows, Cout, ix,
                                       _p(scale), _p(shift), _p(residual), int(relu), _p(stats), _stream()), "rd_conv_fwd")
    if prof:
        e1.record()
        # algorithmic flops: dense 2*k^2*Cin*Cout*rows_out; sparse 2*pairs*Cin*Cout (pairs = valid table entries,
        # counted once per table on the device and read after the timed region)
        if ix.mode == 0:
            pairs = getattr(nbr_keepalive, "_rd_pairs", None) if nbr_keepalive is not None else None
            if pairs is None and nbr_keepalive is not None:
                pairs = (nbr_keepalive >= 0).sum()
                nbr_keepalive._rd_pairs = pairs
            CONV_PROFILE.append((e0, e1, pairs, 2.0 * Cin * Cout, (in_rows, Cin, Cout, taps, ix.mode, tile)))
        else:
            CONV_PROFILE.append((e0, e1, None, 2.0 * out_rows * taps * Cin * Cout, (in_rows, Cin, Cout, taps, ix.mode, tile)))
    return out


def _conv_flops_entry(ix, e0, e1, in_rows, out_rows, Cin, Cout, taps, tile, nbr_keepalive, mode_off=0):
    """CONV_PROFILE record of one launch (see conv_fwd)."""
    if ix.mode == 0:
        pairs = getattr(nbr_keepalive, "_rd_pairs", None) if nbr_keepalive is not None else None
        if pairs is None and nbr_keepalive is not None:
            pairs = (nbr_keepalive >= 0).sum()
            nbr_keepalive._rd_pairs = pairs
        return (e0, e1, pairs, 2.0 * Cin * Cout, (in_rows, Cin, Cout, taps, mode_off + ix.mode, tile))
    return (e0, e1, None, 2.0 * out_rows * taps * Cin * Cout, (in_rows, Cin, Cout, taps, mode_off + ix.mode, tile))


def _conv_tag(ix, in_rows, out_rows, Cin, Cout, taps, w_format):
    tile = 128 if ((out_rows + 127) // 128) * ((Cout + 127) // 128) >= 384 else 64
    if taps == 1 and get_conv_math() == "bf16x3" and os.environ.get("RD_GEMM_TILE64", "1") != "0":
        tile = 64
    return _kernel_tag(ix, in_rows, out_rows, Cin, Cout, taps, False, tile, w_format)


def conv_bn_act_fwd(x, weight, w_format, taps, bias, ix, out_rows, Cout, stats, gamma, beta, eps, momentum, running_mean, running_var,
                    residual, act, nbr_keepalive=None):
    """ONE library call (rd_conv_bn_act_fwd) for conv (+ bias) -> train-mode BatchNorm -> (+ residual) -> activation:
    -> (raw conv output, y, side (4, Cout)).  weight: w_format 0 = fp32 kernel layout, 1 = split, 2 = fragment-major split.
    stats: zero-filled (2 * Cout,) accumulator for the batch sums."""
    _chk(x, f32, "conv input", 2)
    in_rows, Cin = x.shape
    if weight.numel() != Cout * taps * Cin:
        raise RuntimeError(f"conv weight has {weight.numel()} elements, expected {Cout}*{taps}*{Cin}")
    if ix.mode == 0 and nbr_keepalive is not None and (nbr_keepalive.shape != (out_rows, taps) or nbr_keepalive.dtype != i32):
        raise RuntimeError(f"neighbour table shape {tuple(nbr_keepalive.shape)} != ({out_rows}, {taps})")
    if stats.numel() != 2 * Cout or gamma.numel() != Cout or beta.numel() != Cout:
        raise RuntimeError("conv_bn_act_fwd: stats / gamma / beta do not match Cout")
    if residual is not None and (residual.shape != (out_rows, Cout) or not residual.is_contiguous() or residual.dtype != f32):
        raise RuntimeError("conv_bn_act_fwd: residual must be a contiguous fp32 (out_rows, Cout) tensor")
    if bias is not None and bias.numel() != Cout:
        raise RuntimeError("conv_bn_act_fwd: bias size")
    dev = x.device
    raw = torch.empty((out_rows, Cout), dtype=f32, device=dev)
    y = torch.empty((out_rows, Cout), dtype=f32, device=dev)
    side = torch.empty((4, Cout), dtype=f32, device=dev)
    e0 = e1 = None
    if CONV_PROFILE is not None and (Cout > 64 or PROFILE_ALL) and ix.mode != 3:
        tag = _conv_tag(ix, in_rows, out_rows, Cin, Cout, taps, w_format)
        if PROFILE_TAGS is None or tag in PROFILE_TAGS:
            e0, e1 = timing_event(), timing_event()
    check(native.lib().rd_conv_bn_act_fwd(_p(x), in_rows, Cin, _p(weight), int(w_format), taps, _p(bias), _p(raw), out_rows, Cout, ix, _p(stats),
                                          _p(gamma), _p(beta), eps, momentum, _p(running_mean), _p(running_var), _p(residual), act, _p(y),
                                          _p(side), _ev(e0) if e0 is not None else None, _ev(e1) if e1 is not None else None, _stream()),
          "rd_conv_bn_act_fwd")
    if e0 is not None:
        CONV_PROFILE.append(_conv_flops_entry(ix, e0, e1, in_rows, out_rows, Cin, Cout, taps, tag, nbr_keepalive))
    return raw, y, side


def conv_bn_act_bwd(raw, y, grad_y, gamma, side, act, has_res, w_dgrad, w_format, taps, want_gx, in_rows, Cin, bwd_ix, x, fwd_ix, want_gw,
                    main_raw, side_raw, fwd_nbr=None, bwd_nbr=None):
    """ONE library call (rd_conv_bn_act_bwd): BatchNorm (+ act) backward, data gradient on the main stream, weight gradient on the
    side stream (side_raw None: main stream).  -> (grad_in or None, grad_residual or None, grad_gamma, grad_beta, grad_wk kernel layout or None)."""
    _chk(raw, f32, "raw", 2); _chk(grad_y, f32, "grad_y", 2)
    out_rows, Cout = raw.shape
    if grad_y.shape != raw.shape or (y is not None and y.shape != raw.shape) or side.shape != (4, Cout):
        raise RuntimeError("conv_bn_act_bwd: shape mismatch")
    if want_gx and w_dgrad.numel() != Cout * taps * Cin:
        raise RuntimeError("conv_bn_act_bwd: data-gradient weight operand size")
    if want_gw and (x.shape != (in_rows, Cin) or not x.is_contiguous()):
        raise RuntimeError("conv_bn_act_bwd: layer input shape")
    from . import autograd as _A
    dev = raw.device
    graw = torch.empty_like(raw)
    gres = torch.empty_like(raw) if has_res else None
    g2 = _A.zeros_accum(2 * Cout, dev)
    gx = torch.empty((in_rows, Cin), dtype=f32, device=dev) if want_gx else None
    gwk = _A.zeros_accum(Cout * taps * Cin, dev).view(Cout, taps, Cin) if want_gw else None
    if act == 1 and not has_res:
        y = None
    d0 = d1 = w0 = w1 = None
    if CONV_PROFILE is not None and want_gx and (Cin > 64 or PROFILE_ALL):
        dtag = _conv_tag(bwd_ix, out_rows, in_rows, Cout, Cin, taps, w_format) if w_format else (128 if ((in_rows + 127) // 128) * ((Cin + 127) // 128) >= 384 else 64)
        if PROFILE_TAGS is None or dtag in PROFILE_TAGS:
            d0, d1 = timing_event(), timing_event()
    if WGRAD_PROFILE is not None and want_gw:
        wtag = wgrad_tag(fwd_ix, in_rows, out_rows, Cin, Cout, taps)
        if PROFILE_TAGS is None or wtag in PROFILE_TAGS:
            w0, w1 = timing_event(), timing_event()
    check(native.lib().rd_conv_bn_act_bwd(_p(raw), _p(y), _p(grad_y), out_rows, Cout, _p(gamma), _p(side), act, int(has_res), _p(graw), _p(gres),
                                          _p(g2), _p(w_dgrad) if want_gx else None, int(w_format), taps, _p(gx), in_rows, Cin, bwd_ix,
                                          _p(x) if want_gw else None, fwd_ix, _p(gwk),
                                          _ev(d0) if d0 is not None else None, _ev(d1) if d1 is not None else None,
                                          _ev(w0) if w0 is not None else None, _ev(w1) if w1 is not None else None, main_raw, side_raw),
          "rd_conv_bn_act_bwd")
    if d0 is not None:
        CONV_PROFILE.append(_conv_flops_entry(bwd_ix, d0, d1, out_rows, in_rows, Cout, Cin, taps, dtag, bwd_nbr, mode_off=0 if w_format else 10))
    if w0 is not None:
        pairs = None
        if fwd_ix.mode == 0 and fwd_nbr is not None:
            pairs = getattr(fwd_nbr, "_rd_pairs", None)
            if pairs is None:
                pairs = (fwd_nbr >= 0).sum()
                fwd_nbr._rd_pairs = pairs
        WGRAD_PROFILE.append((w0, w1, pairs, 2.0 * Cin * Cout if pairs is not None else 2.0 * out_rows * taps * Cin * Cout,
                              (in_rows, Cin, Cout, taps, fwd_ix.mode, wtag)))
    return gx, gres, g2[:Cout], g2[Cout:], gwk, graw


def wgrad_tag(ix, in_rows, out_rows, Cin, Cout, taps, in_split=False, go_split=False):
    """Which instantiation conv_wgrad_impl (conv.hip) launches: the Cin tile is 128 for Cin >= 128 in bf16x3 mode; in exact fp32 only when
    that leaves >= 32 (tap, tile) pairs; dense stride-1 3x3 layers take the halo-staged kernel."""
    b3 = get_conv_math() == "bf16x3" and Cout >= 64 and Cin >= 64
    wide = Cin >= 128 if b3 else (Cin >= 128 and Cout >= 64 and taps * ((Cout + 127) // 128) * ((Cin + 127) // 128) >= 32)
    tag = ("wgrad_b3_" if b3 else "wgrad_f32_") + ("deform_" if ix.mode == 3 else "") + ("128" if wide else "64")
    if (get_conv_math() == "bf16x3" and ix.mode == 1 and taps == 9 and ix.KH == 3 and ix.KW == 3 and ix.stride == 1 and ix.pad == 1
            and ix.Hin == ix.Hout and ix.Win == ix.Wout and not in_split and not go_split and Cin % 4 == 0 and Cout % 4 == 0 and Cin >= 32
            and Cout >= 32 and in_rows == out_rows and os.environ.get("RD_WGRAD_D3", "1") != "0"):
        tag = "wgrad_d3"
    return tag


def conv_dgrad(grad_out, weight_k, taps, in_rows, Cin, ix_bwd, nbr_keepalive=None):
    """grad_out (out_rows, Cout), weight_k the FORWARD kernel layout (Cout, taps, Cin) -> grad_in (in_rows, Cin)."""
    _chk(grad_out, f32, "dgrad grad_out", 2)
    _chk(weight_k, f32, "dgrad weight")
    out_rows, Cout = grad_out.shape
    if weight_k.numel() != Cout * taps * Cin:
        raise RuntimeError(f"conv_dgrad: weight has {weight_k.numel()} elements, expected {Cout}*{taps}*{Cin}")
    if ix_bwd.mode == 0 and nbr_keepalive is not None:
        if nbr_keepalive.shape != (in_rows, taps) or nbr_keepalive.dtype != i32 or not nbr_keepalive.is_contiguous():
            raise RuntimeError(f"backward neighbour table shape {tuple(nbr_keepalive.shape)} != ({in_rows}, {taps})")
    gx = torch.empty((in_rows, Cin), dtype=f32, device=grad_out.device)
    tile = 128 if ((in_rows + 127) // 128) * ((Cin + 127) // 128) >= 384 else 64
    prof = CONV_PROFILE is not None and (Cin > 64 or PROFILE_ALL) and (PROFILE_TAGS is None or tile in PROFILE_TAGS)
    if prof:
        e0 = timing_event(); e1 = timing_event()
        e0.record()
    check(native.lib().rd_conv_dgrad(_p(grad_out), out_rows, Cout, _p(weight_k), taps, _p(gx), in_rows, Cin, ix_bwd, _stream()),
          "rd_conv_dgrad")
    if prof:
        e1.record()
        if ix_bwd.mode == 0:
            pairs = getattr(nbr_keepalive, "_rd_pairs", None) if nbr_keepalive is not None else None
            if pairs is None and nbr_keepalive is not None:
                pairs = (nbr_keepalive >= 0).sum()
                nbr_keepalive._rd_pairs = pairs
            CONV_PROFILE.append((e0, e1, pairs, 2.0 * Cin * Cout, (out_rows, Cout, Cin, taps, 10 + ix_bwd.mode, tile)))
        else:
            CONV_PROFILE.append((e0, e1, None, 2.0 * in_rows * taps * Cin * Cout, (out_rows, Cout, Cin, taps, 10 + ix_bwd.mode, tile)))
    return gx


# Optional timing hook for the weight-gradient kernel (bench.py RD_BENCH_SHAPES=1): (start, end, pairs, flops factor, shape)
WGRAD_PROFILE = None


def conv_wgrad(x, grad_out, taps, ix, nbr_keepalive=None, in_split=False, go_split=False):
    """-> grad weight in kernel layout (Cout, taps, Cin).  in_split / go_split: operand already in split format (bf16x3 mode)."""
    _chk(x, f32, "wgrad input", 2)
    _chk(grad_out, f32, "wgrad grad_out", 2)
    in_rows, Cin = x.shape
    out_rows, Cout = grad_out.shape
    from . import autograd as _A
    gw = _A.zeros_accum(Cout * taps * Cin, x.device).view(Cout, taps, Cin)          # zero-initialised accumulator (atomics)
    prof = WGRAD_PROFILE is not None
    if prof:
        tag = wgrad_tag(ix, in_rows, out_rows, Cin, Cout, taps, in_split, go_split)
        prof = PROFILE_TAGS is None or tag in PROFILE_TAGS
    if prof:
        e0 = timing_event(); e1 = timing_event()
        e0.record()
    if in_split or go_split:
        check(native.lib().rd_conv_wgrad_split(_p(x), int(in_split), in_rows, Cin, _p(grad_out), int(go_split), out_rows, Cout, taps,
                                               ix, _p(gw), _stream()), "rd_conv_wgrad_split")
    else:
        check(native.lib().rd_conv_wgrad(_p(x), in_rows, Cin, _p(grad_out), out_rows, Cout, taps, ix, _p(gw), _stream()),
              "rd_conv_wgrad")
    if prof:
        e1.record()
        pairs = None
        if ix.mode == 0 and nbr_keepalive is not None:
            pairs = getattr(nbr_keepalive, "_rd_pairs", None)
            if pairs is None:
                pairs = (nbr_keepalive >= 0).sum()
                nbr_keepalive._rd_pairs = pairs
        WGRAD_PROFILE.append((e0, e1, pairs, 2.0 * Cin * Cout if pairs is not None else 2.0 * out_rows * taps * Cin * Cout,
                              (in_rows, Cin, Cout, taps, ix.mode, tag)))
    return gw


def weight_layout(src, Cout, Cin, taps, kind, flip=False, out_shape=None):
    _chk(src, f32, "weight")
    if src.numel() != Cout * Cin * taps:
        raise RuntimeError("weight_layout: element count mismatch")
    dst = torch.empty(out_shape if out_shape is not None else (Cout * Cin * taps,), dtype=f32, device=src.device)
    check(native.lib().rd_weight_layout(_p(src), _p(dst), Cout, Cin, taps, kind, int(flip), _stream()), "rd_weight_layout")
    return dst


def weight_layout_multi(jobs, n):
    """jobs: ctypes array of native.LayoutJob (host); one launch for all of them."""
    check(native.lib().rd_weight_layout_multi(jobs, int(n), _stream()), "rd_weight_layout_multi")


def colsum(x):
    """Column sums (bias gradients): one launch, per-block partials combined with fp32 atomics into a zero-filled output."""
    _chk(x, f32, "colsum input", 2)
    rows, C = x.shape
    if C % 4 != 0:       # tiny heads (1..3 channels): not worth a kernel variant
        raise RuntimeError("colsum: C must be a multiple of 4")
    from . import autograd as _A
    out = _A.zeros_accum(C, x.device)
    check(native.lib().rd_colsum(_p(x), rows, C, _p(out), _stream()), "rd_colsum")
    return out


# ------------------------------------------------------------------------------------------ batch norm
def bn_stats(x, extra=0):
    """-> (2C + extra,) zero-filled, [0, 2C) = per-channel sum and sum of squares over the rows."""
    _chk(x, f32, "bn input", 2)
    rows, C = x.shape
    from . import autograd as _A
    stats = _A.zeros_stats(2 * C + extra, x.device)
    check(native.lib().rd_bn_stats(_p(x), rows, C, _p(stats), _stream()), "rd_bn_stats")
    return stats


# Optional timing hook for the HBM-bound BatchNorm forward pass (bench.py: achieved GB/s of a streaming kernel next to the MFMA
# roofline): (start event, end event, algorithmic bytes) per launch.  None = off.
BN_PROFILE = None


def bn_train_fwd(x, stats, gamma, beta, eps, momentum, running_mean, running_var, residual, act, sync=False):
    """finalize + affine + residual + activation in one launch -> y, side (4, C) = [mean | rstd | scale | shift] for the backward
    pass (one tensor: four row views cost four tensor objects per layer and step).
    sync: stats holds 2C + 1 values (sums and the row count) already summed over the process group (SyncBatchNorm)."""
    _chk(x, f32, "bn input", 2)
    rows, C = x.shape
    if _chk(stats, f32, "bn stats").numel() != 2 * C + (1 if sync else 0):
        raise RuntimeError("bn_train_fwd: stats must hold 2*C sums (+ the row count when synchronised)")
    if residual is not None and _chk(residual, f32, "residual", 2).shape != x.shape:
        raise RuntimeError("bn_train_fwd: residual shape mismatch")
    if gamma is not None and _chk(gamma, f32, "gamma").numel() != C:
        raise RuntimeError(f"bn_train_fwd: gamma must have {C} elements")
    if beta is not None and _chk(beta, f32, "beta").numel() != C:
        raise RuntimeError(f"bn_train_fwd: beta must have {C} elements")
    if running_mean is not None and _chk(running_mean, f32, "running_mean").numel() != C:
        raise RuntimeError(f"bn_train_fwd: running_mean must have {C} elements")
    if running_var is not None and _chk(running_var, f32, "running_var").numel() != C:
        raise RuntimeError(f"bn_train_fwd: running_var must have {C} elements")
    side = torch.empty((4, C), dtype=f32, device=x.device)
    y = torch.empty_like(x)
    prof = BN_PROFILE is not None and rows * C >= 2_000_000          # the bench only uses launches that move >= 16 MB
    if prof:
        e0 = timing_event(); e1 = timing_event()
        e0.record()
    fn = native.lib().rd_bn_train_fwd_sync if sync else native.lib().rd_bn_train_fwd
    sp = side.data_ptr()
    check(fn(_p(x), rows, C, _p(stats), _p(gamma), _p(beta), eps, momentum, _p(running_mean), _p(running_var),
             _p(residual), act, _p(y), sp, sp + 4 * C, sp + 8 * C, sp + 12 * C, _stream()), "rd_bn_train_fwd")
    if prof:
        e1.record()
        BN_PROFILE.append((e0, e1, float(rows) * C * 4 * (3 if residual is not None else 2), (rows, C)))   # read x (+ residual), write y
    return y, side


def bn_finalize(stats, rows, C, gamma, beta, eps, momentum, running_mean, running_var, sync=False):
    """sync: stats[2C] is the row count, the whole buffer already summed over the process group (`rows` is ignored)."""
    dev = stats.device
    mean = torch.empty(C, dtype=f32, device=dev)
    rstd = torch.empty(C, dtype=f32, device=dev)
    scale = torch.empty(C, dtype=f32, device=dev)
    shift = torch.empty(C, dtype=f32, device=dev)
    if sync:
        check(native.lib().rd_bn_finalize_sync(_p(stats), C, _p(gamma), _p(beta), eps, momentum, _p(running_mean), _p(running_var),
                                               _p(mean), _p(rstd), _p(scale), _p(shift), _stream()), "rd_bn_finalize_sync")
        return mean, rstd, scale, shift
    check(native.lib().rd_bn_finalize(_p(stats), rows, C, _p(gamma), _p(beta), eps, momentum, _p(running_mean), _p(running_var),
                                      _p(mean), _p(rstd), _p(scale), _p(shift), _stream()), "rd_bn_finalize")
    return mean, rstd, scale, shift


def affine_act(x, scale, shift, residual, act):
    _chk(x, f32, "affine input", 2)
    rows, C = x.shape
    y = torch.empty_like(x)
    if residual is not None:
        _chk(residual, f32, "residual", 2)
        if residual.shape != x.shape:
            raise RuntimeError("affine_act: residual shape mismatch")
    check(native.lib().rd_affine_act(_p(x), rows, C, _p(scale), _p(shift), _p(residual), act, _p(y), _stream()), "rd_affine_act")
    return y


def bn_bwd(x, y, grad_y, gamma, side, act, has_residual, sync=None):
    """side: (4, C) [mean | rstd | scale | shift] of bn_train_fwd.  sync = (allreduce, count_dev): SyncBatchNorm -- `allreduce(t)` sums a device tensor over the process group in place,
    count_dev is the group-wide row count; the returned parameter gradients are this rank's own sums (torch semantics)."""
    _chk(x, f32, "x", 2); _chk(grad_y, f32, "grad_y", 2)
    rows, C = x.shape
    if grad_y.shape != x.shape or (y is not None and y.shape != x.shape):
        raise RuntimeError("bn_bwd: shape mismatch")
    if _chk(side, f32, "bn side", 2).shape != (4, C):
        raise RuntimeError("bn_bwd: side must be the (4, C) tensor of bn_train_fwd")
    mean = side.data_ptr()
    rstd, scale, shift = mean + 4 * C, mean + 8 * C, mean + 12 * C
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if has_residual else None
    from . import autograd as _A
    g2 = _A.zeros_accum(2 * C, x.device)            # [grad_gamma | grad_beta], accumulated by the reduction pass
    gg, gb = g2[:C], g2[C:]
    if act == 1 and not has_residual:
        y = None                                     # ReLU mask re-derived from x*scale + shift: one tensor less to stream
    if sync is not None:
        allreduce, count_dev = sync
        check(native.lib().rd_bn_bwd_reduce(_p(x), _p(y), _p(grad_y), rows, C, mean, rstd, scale, shift, act, int(has_residual),
                                            _p(gg), _p(gb), _stream()), "rd_bn_bwd_reduce")
        tot = g2.clone()
        allreduce(tot)
        check(native.lib().rd_bn_bwd_apply(_p(x), _p(y), _p(grad_y), rows, C, _p(gamma), mean, rstd, scale, shift, act,
                                           int(has_residual), _p(tot[:C]), _p(tot[C:]), _p(count_dev), _p(gx), _p(gres), _stream()),
              "rd_bn_bwd_apply")
        return gx, gres, gg, gb
    check(native.lib().rd_bn_bwd(_p(x), _p(y), _p(grad_y), rows, C, _p(gamma), mean, rstd, scale, shift, act,
                                 int(has_residual), _p(gx), _p(gres), _p(gg), _p(gb), _stream()), "rd_bn_bwd")
    return gx, gres, gg, gb


def cat2_rows(a, b):
    _chk(a, f32, "cat a", 2); _chk(b, f32, "cat b", 2)
    if a.shape[0] != b.shape[0]:
        raise RuntimeError("cat2_rows: row counts differ")
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=f32, device=a.device)
    check(native.lib().rd_cat2_rows(_p(a), a.shape[1], _p(b), b.shape[1], a.shape[0], _p(out), _stream()), "rd_cat2_rows")
    return out


def split2_rows(g, Ca, Cb):
    _chk(g, f32, "cat grad", 2)
    if g.shape[1] != Ca + Cb:
        raise RuntimeError("split2_rows: channel counts do not add up")
    ga = torch.empty((g.shape[0], Ca), dtype=f32, device=g.device)
    gb = torch.empty((g.shape[0], Cb), dtype=f32, device=g.device)
    check(native.lib().rd_split2_rows(_p(g), g.shape[0], Ca, Cb, _p(ga), _p(gb), _stream()), "rd_split2_rows")
    return ga, gb


# ------------------------------------------------------------------------------------------ sparse <-> dense
def rows_to_dense(feats, coords, batch, H, W):
    _chk(feats, f32, "feats", 2); _chk(coords, i32, "coords", 2)
    n, C = feats.shape
    if coords.shape != (n, 3):
        raise RuntimeError("rows_to_dense: coords shape mismatch")
    dense = torch.empty((batch * H * W, C), dtype=f32, device=feats.device)
    check(native.lib().rd_rows_to_dense(_p(feats), _p(coords), n, C, batch, H, W, _p(dense), _stream()), "rd_rows_to_dense")
    return dense


def dense_to_rows(dense, coords, batch, H, W):
    _chk(dense, f32, "dense", 2); _chk(coords, i32, "coords", 2)
    C = dense.shape[1]
    n = coords.shape[0]
    if dense.shape[0] != batch * H * W:
        raise RuntimeError("dense_to_rows: dense shape mismatch")
    feats = torch.empty((n, C), dtype=f32, device=dense.device)
    check(native.lib().rd_dense_to_rows(_p(dense), _p(coords), n, C, batch, H, W, _p(feats), _stream()), "rd_dense_to_rows")
    return feats


# ------------------------------------------------------------------------------------------ DCNv2
def dcn_prep(offset_base, off_stride, mask_base, mask_stride, apply_sigmoid, B, H, W, Ho, Wo, k, stride, pad, dil=1):
    """offset_base / mask_base: tensors whose data_ptr is the first offset / mask channel of row 0."""
    taps = k * k
    rows = B * Ho * Wo
    dev = offset_base.device
    samp_idx = torch.empty((rows, taps, 4), dtype=i32, device=dev)
    samp_w = torch.empty((rows, taps, 4), dtype=f32, device=dev)
    check(native.lib().rd_dcn_prep(_p(offset_base), off_stride, _p(mask_base), mask_stride, int(apply_sigmoid), B, H, W, Ho, Wo, k, k,
                                   stride, pad, dil, _p(samp_idx), _p(samp_w), _stream()), "rd_dcn_prep")
    return samp_idx, samp_w


def dcn_columns(x_rows, samp_idx, samp_w):
    """x_rows (in_rows, C), sampling table (out_rows, taps, 4) -> deformed, mask-modulated columns (out_rows, taps * C)."""
    _chk(x_rows, f32, "dcn x", 2); _chk(samp_idx, i32, "dcn samp_idx", 3); _chk(samp_w, f32, "dcn samp_w", 3)
    rows, taps = samp_idx.shape[0], samp_idx.shape[1]
    C = x_rows.shape[1]
    if samp_idx.shape != (rows, taps, 4) or samp_w.shape != (rows, taps, 4):
        raise RuntimeError("dcn_columns: the sampling table must be (rows, taps, 4)")
    col = torch.empty((rows, taps * C), dtype=f32, device=x_rows.device)
    check(native.lib().rd_dcn_columns(_p(x_rows), x_rows.shape[0], C, _p(samp_idx), _p(samp_w), rows, taps, _p(col), _stream()), "rd_dcn_columns")
    return col


def conv_index_deform(samp_idx, samp_w):
    ix = ConvIndex()
    ix.mode = 3
    ix.nbr = None
    ix.samp_idx = samp_idx.data_ptr()
    ix.samp_w = samp_w.data_ptr()
    return ix


def dcn_bwd_data(x_rows, colgrad, offset_base, off_stride, mask_base, mask_stride, apply_sigmoid, B, H, W, Ho, Wo, k, stride, pad,
                 grad_offset_base, goff_stride, grad_mask_base, gmask_stride, dil=1):
    _chk(x_rows, f32, "dcn x", 2); _chk(colgrad, f32, "dcn colgrad")
    C = x_rows.shape[1]
    if x_rows.shape[0] != B * H * W or colgrad.numel() != B * Ho * Wo * k * k * C:
        raise RuntimeError("dcn_bwd_data: shape mismatch")
    gx = torch.empty_like(x_rows)
    check(native.lib().rd_dcn_bwd_data(_p(x_rows), C, _p(colgrad), _p(offset_base), off_stride, _p(mask_base), mask_stride,
                                       int(apply_sigmoid), B, H, W, Ho, Wo, k, k, stride, pad, dil, _p(gx), _p(grad_offset_base),
                                       goff_stride, _p(grad_mask_base), gmask_stride, _stream()), "rd_dcn_bwd_data")
    return gx


# ------------------------------------------------------------------------------------------ distillation losses
def afd_fwd(lidar, radar_a, radar_b, batch):
    for t in (lidar, radar_a, radar_b):
        _chk(t, f32, "afd map", 2)
    if not (lidar.shape == radar_a.shape == radar_b.shape):
        raise RuntimeError("afd_fwd: map shapes differ")
    rows, C = lidar.shape
    dev = lidar.device
    out = torch.empty(4, dtype=f32, device=dev)
    coef = torch.empty(6, dtype=f32, device=dev)
    rowinfo = torch.empty(2 * rows * 2, dtype=f32, device=dev)
    nb = native.lib().rd_afd_ws_bytes(rows)
    ws = torch.empty(nb // 4, dtype=f32, device=dev)
    check(native.lib().rd_afd_fwd(_p(lidar), _p(radar_a), _p(radar_b), rows, C, batch, _p(out), _p(coef), _p(rowinfo), _p(ws), nb, _stream()),
          "rd_afd_fwd")
    return out, coef, rowinfo


def afd_fwd_bf16(lidar, radar_a, radar_b, batch):
    """afd_fwd on maps stored in bf16 (BASELINE configs[2]) -> out[4] = (feature_a, mask_a, feature_b, mask_b)."""
    for t in (lidar, radar_a, radar_b):
        _chk(t, torch.bfloat16, "afd bf16 map", 2)
    if not (lidar.shape == radar_a.shape == radar_b.shape):
        raise RuntimeError("afd_fwd_bf16: map shapes differ")
    rows, C = lidar.shape
    dev = lidar.device
    out = torch.empty(4, dtype=f32, device=dev)
    coef = torch.empty(6, dtype=f32, device=dev)
    rowinfo = torch.empty(2 * rows * 2, dtype=f32, device=dev)
    nb = native.lib().rd_afd_ws_bytes(rows)
    ws = torch.empty(nb // 4, dtype=f32, device=dev)
    check(native.lib().rd_afd_fwd_bf16(_p(lidar), _p(radar_a), _p(radar_b), rows, C, batch, _p(out), _p(coef), _p(rowinfo), _p(ws), nb, _stream()),
          "rd_afd_fwd_bf16")
    return out


def afd_bwd(lidar, radar_a, radar_b, rowinfo, coef, gscale):
    rows, C = lidar.shape
    ga, gb = torch.empty_like(radar_a), torch.empty_like(radar_b)
    check(native.lib().rd_afd_bwd(_p(lidar), _p(radar_a), _p(radar_b), rows, C, _p(rowinfo), _p(coef), _p(_chk(gscale, f32, "gscale")),
                                  _p(ga), _p(gb), _stream()), "rd_afd_bwd")
    return ga, gb


def pfd_fwd(r1, l1, r2, l2, gt_hm, hm_logits):
    for t in (r1, l1, r2, l2, gt_hm, hm_logits):
        _chk(t, f32, "pfd map", 2)
    rows, C = r1.shape
    if not (l1.shape == r2.shape == l2.shape == r1.shape) or gt_hm.shape != hm_logits.shape or gt_hm.shape[0] != rows:
        raise RuntimeError("pfd_fwd: shape mismatch")
    dev = r1.device
    cls = torch.empty(rows, dtype=torch.int8, device=dev)
    counts = torch.empty(2, dtype=i32, device=dev)
    out = torch.empty(1, dtype=f32, device=dev)
    ws = torch.empty(1024, dtype=f32, device=dev)
    check(native.lib().rd_pfd_fwd(_p(r1), _p(l1), _p(r2), _p(l2), rows, C, _p(gt_hm), _p(hm_logits), gt_hm.shape[1], _p(cls), _p(counts),
                                  _p(out), _p(ws), 4096, _stream()), "rd_pfd_fwd")
    return out, cls, counts


def pfd_bwd(r1, l1, r2, l2, cls, counts, gscale):
    rows, C = r1.shape
    g1, g2 = torch.empty_like(r1), torch.empty_like(r2)
    check(native.lib().rd_pfd_bwd(_p(r1), _p(l1), _p(r2), _p(l2), rows, C, _p(cls), _p(counts), _p(_chk(gscale, f32, "gscale")),
                                  _p(g1), _p(g2), _stream()), "rd_pfd_bwd")
    return g1, g2


# ------------------------------------------------------------------------------------------ rotated overlap
def boxes_aligned_overlap_bev(boxes_a, boxes_b):
    _chk(boxes_a, f32, "boxes_a", 2); _chk(boxes_b, f32, "boxes_b", 2)
    if boxes_a.shape != boxes_b.shape or boxes_a.shape[1] != 7:
        raise RuntimeError("boxes_aligned_overlap_bev: expected two (N,7) tensors")
    n = boxes_a.shape[0]
    out = torch.zeros((n, 1), dtype=f32, device=boxes_a.device)
    check(native.lib().rd_boxes_aligned_overlap_bev(n, _p(boxes_a), _p(boxes_b), _p(out), _stream()), "rd_boxes_aligned_overlap_bev")
    return out


# ------------------------------------------------------------------------------------------ depthwise conv
def dwconv_fwd(x_rows, w_tc, bias, B, H, W, K, flip=False):
    _chk(x_rows, f32, "dwconv input", 2); _chk(w_tc, f32, "dwconv weight", 2)
    C = x_rows.shape[1]
    if x_rows.shape[0] != B * H * W or w_tc.shape != (K * K, C):
        raise RuntimeError("dwconv_fwd: shape mismatch")
    out = torch.empty_like(x_rows)
    check(native.lib().rd_dwconv_fwd(_p(x_rows), _p(w_tc), _p(bias), B, H, W, C, K, int(flip), _p(out), _stream()), "rd_dwconv_fwd")
    return out


def dwconv_wgrad(x_rows, go_rows, B, H, W, K):
    _chk(x_rows, f32, "dwconv input", 2); _chk(go_rows, f32, "dwconv grad", 2)
    C = x_rows.shape[1]
    if x_rows.shape != go_rows.shape or x_rows.shape[0] != B * H * W:
        raise RuntimeError("dwconv_wgrad: shape mismatch")
    gw = torch.empty((K * K, C), dtype=f32, device=x_rows.device)
    check(native.lib().rd_dwconv_wgrad(_p(x_rows), _p(go_rows), B, H, W, C, K, _p(gw), None, 0, _stream()), "rd_dwconv_wgrad")
    return gw


# ------------------------------------------------------------------------------------------ CenterHead targets
def center_targets(gt_boxes, cfg_struct):
    """gt_boxes (B, M, D) CUDA fp32 -> stacked targets dict (see rd_center_targets)."""
    _chk(gt_boxes, f32, "gt_boxes", 3)
    B, M, D = gt_boxes.shape
    c = cfg_struct
    dev = gt_boxes.device
    hm = torch.empty((B, c.n_channels, c.fy, c.fx), dtype=f32, device=dev)
    # the four per-slot outputs share ONE allocation in the order the library fills them (one memset instead of four); slots * D * 4
    # bytes is a multiple of 8 whenever slots is even -- otherwise pad so the int64 views stay aligned
    slots = c.n_heads * B * c.max_objs
    n_tb, n_i, n_gb = slots * D * 4, slots * 8, slots * 7 * 4
    if n_tb % 8 == 0:
        raw = torch.empty(n_tb + 2 * n_i + n_gb, dtype=torch.uint8, device=dev)
        tb = raw[:n_tb].view(f32).view(c.n_heads, B, c.max_objs, D)
        inds = raw[n_tb:n_tb + n_i].view(torch.int64).view(c.n_heads, B, c.max_objs)
        masks = raw[n_tb + n_i:n_tb + 2 * n_i].view(torch.int64).view(c.n_heads, B, c.max_objs)
        gb = raw[n_tb + 2 * n_i:].view(f32).view(c.n_heads, B, c.max_objs, 7)
    else:
        tb = torch.empty((c.n_heads, B, c.max_objs, D), dtype=f32, device=dev)
        inds = torch.empty((c.n_heads, B, c.max_objs), dtype=torch.int64, device=dev)
        masks = torch.empty((c.n_heads, B, c.max_objs), dtype=torch.int64, device=dev)
        gb = torch.empty((c.n_heads, B, c.max_objs, 7), dtype=f32, device=dev)
    check(native.lib().rd_center_targets(_p(gt_boxes), B, M, D, c, _p(hm), _p(tb), _p(inds), _p(masks), _p(gb), _stream()),
          "rd_center_targets")
    return {"heatmaps": hm, "target_boxes": tb, "inds": inds, "masks": masks, "gt_box": gb}


# ------------------------------------------------------------------------------------------ narrow convs of the head branches
class BranchTable:
    """Host-side (cin_off, col_off, n_out) tables of rd_nconv_* as ctypes int arrays (built once per head module)."""

    def __init__(self, cin_off, col_off, n_out):
        n = len(n_out)
        if not (len(cin_off) == len(col_off) == n and 1 <= n <= 64):
            raise RuntimeError("BranchTable: 1..64 branches, equal-length tables")
        arr = ctypes.c_int32 * n
        self.nb = n
        self.cin, self.col, self.n = arr(*cin_off), arr(*col_off), arr(*n_out)
        self.no = max(c + k for c, k in zip(col_off, n_out))
        self.cin_max = max(cin_off) + 64


def _nconv_chk(y, B, H, W, tab, who):
    _chk(y, f32, who, 2)
    if y.shape[0] != B * H * W or y.shape[1] < tab.cin_max or y.shape[1] % 4:
        raise RuntimeError(f"{who}: shape {tuple(y.shape)} does not hold {tab.nb} branches of 64 channels over {B}x{H}x{W} pixels")


def nconv_fwd(y, weight, bias, B, H, W, tab):
    """y (B*H*W, NB*64) -> (B*H*W, NO); weight (NO, 64, 3, 3) torch layout, bias (NO,) or None."""
    _nconv_chk(y, B, H, W, tab, "nconv input")
    _chk(weight, f32, "nconv weight")
    if tuple(weight.shape) != (tab.no, 64, 3, 3) or (bias is not None and _chk(bias, f32, "nconv bias").numel() != tab.no):
        raise RuntimeError(f"nconv: weight {tuple(weight.shape)} / bias do not match {tab.no} output columns")
    out = torch.empty((y.shape[0], tab.no), dtype=f32, device=y.device)
    check(native.lib().rd_nconv_fwd(_p(y), y.shape[1], _p(weight), _p(bias), B, H, W, tab.no, tab.nb, tab.cin, tab.col, tab.n, _p(out),
                                    _stream()), "rd_nconv_fwd")
    return out


def nconv_dgrad(grad_out, weight, B, H, W, tab, ldy):
    _chk(grad_out, f32, "nconv grad_out", 2)
    if grad_out.shape != (B * H * W, tab.no) or tuple(weight.shape) != (tab.no, 64, 3, 3) or ldy < tab.cin_max:
        raise RuntimeError("nconv_dgrad: shape mismatch")
    # every column of grad_y must be owned by a branch (the kernel writes only those): true when the branches tile the row
    if ldy != tab.nb * 64:
        gy = torch.zeros((grad_out.shape[0], ldy), dtype=f32, device=grad_out.device)
    else:
        gy = torch.empty((grad_out.shape[0], ldy), dtype=f32, device=grad_out.device)
    check(native.lib().rd_nconv_dgrad(_p(grad_out), _p(_chk(weight, f32, "nconv weight")), B, H, W, tab.no, tab.nb, tab.cin, tab.col, tab.n,
                                      _p(gy), ldy, _stream()), "rd_nconv_dgrad")
    return gy


def nconv_dgrad_bn(grad_out, weight, x, gamma, side, B, H, W, tab):
    """Data gradient of the narrow convolutions fused with the backward of the train-mode BatchNorm + ReLU in front of them.
    x (rows, NB*64): the BatchNorm's input; side: its (4, C) [mean | rstd | scale | shift] -> grad_x, grad_gamma, grad_beta."""
    _chk(grad_out, f32, "nconv grad_out", 2); _chk(x, f32, "bn input", 2)
    rows, C = x.shape
    if grad_out.shape != (B * H * W, tab.no) or tuple(weight.shape) != (tab.no, 64, 3, 3) or rows != B * H * W or C != tab.nb * 64:
        raise RuntimeError("nconv_dgrad_bn: shape mismatch (every channel of x must belong to a branch)")
    if _chk(side, f32, "bn side", 2).shape != (4, C) or (gamma is not None and _chk(gamma, f32, "gamma").numel() != C):
        raise RuntimeError("nconv_dgrad_bn: side must be (4, C), gamma (C,)")
    from . import autograd as _A
    g2 = _A.zeros_accum(2 * C, x.device)            # [grad_gamma | grad_beta]
    gx = torch.empty_like(x)
    sp = side.data_ptr()
    check(native.lib().rd_nconv_dgrad_bn(_p(grad_out), _p(_chk(weight, f32, "nconv weight")), _p(x), _p(gamma), sp, sp + 4 * C, sp + 8 * C, sp + 12 * C,
                                         B, H, W, tab.no, tab.nb, tab.cin, tab.col, tab.n, _p(gx), C, _p(g2), g2.data_ptr() + 4 * C, _stream()),
          "rd_nconv_dgrad_bn")
    return gx, g2[:C], g2[C:]


def nconv_wgrad(y, grad_out, B, H, W, tab):
    _nconv_chk(y, B, H, W, tab, "nconv input")
    _chk(grad_out, f32, "nconv grad_out", 2)
    if grad_out.shape != (B * H * W, tab.no):
        raise RuntimeError("nconv_wgrad: grad_out shape mismatch")
    from . import autograd as _A
    gw = _A.zeros_accum(tab.no * 576, y.device).view(tab.no, 64, 3, 3)
    check(native.lib().rd_nconv_wgrad(_p(y), y.shape[1], _p(grad_out), B, H, W, tab.no, tab.nb, tab.cin, tab.col, tab.n, _p(gw), _stream()),
          "rd_nconv_wgrad")
    return gw


# ------------------------------------------------------------------------------------------ ConvNeXt MLP tail
def gelu_grn_fwd(z, B, gamma, beta):
    """z (B*hw, C) -> (out, a = gelu(z), ssq (B, C)): GELU + GRN in two passes."""
    _chk(z, f32, "gelu_grn input", 2)
    rows, C = z.shape
    if rows % B or _chk(gamma, f32, "gamma").numel() != C or _chk(beta, f32, "beta").numel() != C:
        raise RuntimeError("gelu_grn_fwd: shape mismatch")
    a = torch.empty_like(z); out = torch.empty_like(z)
    ssq = torch.empty((B, C), dtype=f32, device=z.device)
    check(native.lib().rd_gelu_grn_fwd(_p(z), B, rows // B, C, _p(gamma), _p(beta), _p(a), _p(ssq), _p(out), _stream()), "rd_gelu_grn_fwd")
    return out, a, ssq


def gelu_grn_bwd(grad_out, a, z, ssq, B, gamma):
    _chk(grad_out, f32, "gelu_grn grad_out", 2)
    rows, C = grad_out.shape
    gz = torch.empty_like(z)
    buf = torch.empty((B + 2) * C, dtype=f32, device=z.device)          # [S scratch | grad_gamma | grad_beta]: zero-filled by ONE memset inside
    ws, gg, gb = buf[:B * C], buf[B * C:(B + 1) * C], buf[(B + 1) * C:]
    check(native.lib().rd_gelu_grn_bwd(_p(grad_out), _p(a), _p(z), _p(ssq), B, rows // B, C, _p(gamma), _p(ws), _p(gz), _p(gg), _p(gb), _stream()),
          "rd_gelu_grn_bwd")
    return gz, gg, gb


# ------------------------------------------------------------------------------------------ LayerNorm over channels-last rows
def layernorm_fwd(x, gamma, beta, eps):
    _chk(x, f32, "layernorm input", 2)
    rows, C = x.shape
    if _chk(gamma, f32, "gamma").numel() != C or _chk(beta, f32, "beta").numel() != C:
        raise RuntimeError("layernorm_fwd: gamma / beta must have C elements")
    y = torch.empty_like(x)
    stat = torch.empty((2, rows), dtype=f32, device=x.device)
    check(native.lib().rd_layernorm_fwd(_p(x), rows, C, _p(gamma), _p(beta), float(eps), _p(y), _p(stat[0]), _p(stat[1]), _stream()), "rd_layernorm_fwd")
    return y, stat


def layernorm_bwd(x, grad_y, gamma, stat):
    _chk(grad_y, f32, "layernorm grad", 2)
    rows, C = x.shape
    gx = torch.empty_like(x)
    g2 = torch.empty(2 * C, dtype=f32, device=x.device)
    check(native.lib().rd_layernorm_bwd(_p(x), _p(grad_y), rows, C, _p(gamma), _p(stat[0]), _p(stat[1]), _p(gx), _p(g2[:C]), _p(g2[C:]), _stream()),
          "rd_layernorm_bwd")
    return gx, g2[:C], g2[C:]


# ------------------------------------------------------------------------------------------ padded-voxel input format
def voxelize_hard(points, batch, grid_xyz, pc_range, voxel_size, max_points, max_voxels):
    """points (N, 1+C) [batch id, x, y, z, ...] sorted by batch id -> (voxels (M, max_points, C), coords (M, 4) int32 (b, z, y, x),
    num_points (M,) int32): the capacity-limited voxeliser of the padded-voxel format, bit-exact vs spconv's CPU algorithm
    (one host read: the voxel count M)."""
    _chk(points, f32, "points", 2)
    n, nf = points.shape[0], points.shape[1] - 1
    gx, gy, gz = [int(v) for v in grid_xyz]
    dev = points.device
    rows = batch * int(max_voxels)
    voxels = torch.empty((rows, int(max_points), nf), dtype=f32, device=dev)
    coords = torch.empty((rows, 4), dtype=i32, device=dev)
    num = torch.empty(rows, dtype=i32, device=dev)
    m = torch.zeros((), dtype=i32, device=dev)
    nb = native.lib().rd_voxelize_hard_ws_bytes(n, batch, gx, gy, gz)
    ws = torch.empty(nb // 4, dtype=i32, device=dev)
    check(native.lib().rd_voxelize_hard(_p(points), n, nf, batch, gx, gy, gz, float(pc_range[0]), float(pc_range[1]), float(pc_range[2]),
                                        float(voxel_size[0]), float(voxel_size[1]), float(voxel_size[2]), int(max_points), int(max_voxels),
                                        rows, _p(voxels), _p(coords), _p(num), _p(m), _p(ws), nb, _stream()), "rd_voxelize_hard")
    M = int(m.item())
    return voxels[:M], coords[:M], num[:M]


def _pvfe_args(voxels, num_points, coords, weight, use_abs, with_dist):
    _chk(voxels, f32, "voxels", 3); _chk(num_points, i32, "voxel_num_points", 1); _chk(coords, i32, "voxel_coords", 2); _chk(weight, f32, "weight", 2)
    M, P, C = voxels.shape
    if num_points.shape[0] != M or coords.shape != (M, 4):
        raise RuntimeError("pillar_vfe: voxel_num_points / voxel_coords do not match voxels")
    Cout, Cin = weight.shape
    return M, P, C, Cin, Cout, int(bool(use_abs)), int(bool(with_dist))


def pillar_vfe_stats(voxels, num_points, coords, weight, use_abs, with_dist, geom):
    """-> stats (2*Cout,): sum and sum of squares of the PFN Linear outputs over all real slots (BatchNorm1d: n = M*P)."""
    M, P, C, Cin, Cout, ua, wd = _pvfe_args(voxels, num_points, coords, weight, use_abs, with_dist)
    stats = torch.zeros(2 * Cout, dtype=f32, device=voxels.device)
    check(native.lib().rd_pillar_vfe_stats(_p(voxels), _p(num_points), _p(coords), M, P, C, _p(weight), Cin, Cout, ua, wd, *[float(v) for v in geom],
                                           _p(stats), _stream()), "rd_pillar_vfe_stats")
    return stats


def pillar_vfe_max(voxels, num_points, coords, weight, use_abs, with_dist, geom, scale, shift):
    """-> (M, Cout): max over the slots of relu(Linear * scale + shift)."""
    M, P, C, Cin, Cout, ua, wd = _pvfe_args(voxels, num_points, coords, weight, use_abs, with_dist)
    if _chk(scale, f32, "scale").numel() != Cout or _chk(shift, f32, "shift").numel() != Cout:
        raise RuntimeError("pillar_vfe_max: scale / shift must have Cout elements")
    out = torch.empty((M, Cout), dtype=f32, device=voxels.device)
    check(native.lib().rd_pillar_vfe_max(_p(voxels), _p(num_points), _p(coords), M, P, C, _p(weight), Cin, Cout, ua, wd, *[float(v) for v in geom],
                                         _p(scale), _p(shift), _p(out), _stream()), "rd_pillar_vfe_max")
    return out


def pillar_decorate(voxels, num_points, coords, Cin, use_abs, with_dist, geom, ld):
    """-> (M*P, ld) rows: the masked slot features of PillarVFE.forward (pillar_vfe.py:94-118), zero-padded to `ld` columns."""
    _chk(voxels, f32, "voxels", 3); _chk(num_points, i32, "voxel_num_points", 1); _chk(coords, i32, "voxel_coords", 2)
    M, P, C = voxels.shape
    if num_points.shape[0] != M or coords.shape != (M, 4):
        raise RuntimeError("pillar_decorate: voxel_num_points / voxel_coords do not match voxels")
    out = torch.empty((M * P, ld), dtype=f32, device=voxels.device)
    check(native.lib().rd_pillar_decorate(_p(voxels), _p(num_points), _p(coords), M, P, C, int(Cin), int(bool(use_abs)), int(bool(with_dist)),
                                          *[float(v) for v in geom], int(ld), _p(out), _stream()), "rd_pillar_decorate")
    return out


def pfn_pool_fwd(x, M, P, last):
    """x (M*P, C) -> (out, argmax): out (M, C) max over the slots (last) or (M*P, 2C) = [x | max repeated] (pillar_vfe.py:42-49)."""
    _chk(x, f32, "pfn rows", 2)
    C = x.shape[1]
    if x.shape[0] != M * P:
        raise RuntimeError("pfn_pool_fwd: rows != M*P")
    out = torch.empty((M, C) if last else (M * P, 2 * C), dtype=f32, device=x.device)
    argmax = torch.empty((M, C), dtype=i32, device=x.device)
    check(native.lib().rd_pfn_pool_fwd(_p(x), M, P, C, int(bool(last)), _p(out), _p(argmax), _stream()), "rd_pfn_pool_fwd")
    return out, argmax


def pfn_pool_bwd(grad_out, argmax, M, P, last):
    _chk(grad_out, f32, "grad_out", 2); _chk(argmax, i32, "argmax", 2)
    C = argmax.shape[1]
    if tuple(grad_out.shape) != ((M, C) if last else (M * P, 2 * C)):
        raise RuntimeError("pfn_pool_bwd: grad_out shape does not match the forward output")
    gx = torch.empty((M * P, C), dtype=f32, device=grad_out.device)
    check(native.lib().rd_pfn_pool_bwd(_p(grad_out), _p(argmax), M, P, C, int(bool(last)), _p(gx), _stream()), "rd_pfn_pool_bwd")
    return gx


# ------------------------------------------------------------------------------------------ inference post-processing
def nms_bev(boxes_sorted, thresh):
    """boxes (n, 7) sorted by descending score -> (keep (n,) int64 device, num_keep 0-d int32 device): rotated-BEV greedy NMS,
    no host round trip."""
    _chk(boxes_sorted, f32, "nms boxes", 2)
    n = boxes_sorted.shape[0]
    if boxes_sorted.shape[1] != 7:
        raise RuntimeError("nms_bev: boxes must be (n, 7)")
    keep = torch.empty(n, dtype=torch.int64, device=boxes_sorted.device)
    num = torch.zeros((), dtype=i32, device=boxes_sorted.device)
    nb = native.lib().rd_nms_ws_bytes(n)
    ws = torch.empty(max(nb // 8, 1), dtype=torch.int64, device=boxes_sorted.device)
    check(native.lib().rd_nms_bev(n, _p(boxes_sorted), float(thresh), _p(ws), nb, _p(keep), _p(num), _stream()), "rd_nms_bev")
    return keep, num


def boxes_overlap_bev(boxes_a, boxes_b):
    """(na, 7), (nb, 7) -> (na, nb) rotated BEV overlap areas."""
    _chk(boxes_a, f32, "boxes_a", 2); _chk(boxes_b, f32, "boxes_b", 2)
    if boxes_a.shape[1] != 7 or boxes_b.shape[1] != 7:
        raise RuntimeError("boxes_overlap_bev: boxes must be (n, 7)")
    out = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=f32, device=boxes_a.device)
    check(native.lib().rd_boxes_overlap_bev(boxes_a.shape[0], _p(boxes_a), boxes_b.shape[0], _p(boxes_b), _p(out), _stream()),
          "rd_boxes_overlap_bev")
    return out


# ------------------------------------------------------------------------------------------ test switch: ordered reductions
def set_deterministic(on):
    """rd_set_deterministic: every floating-point reduction of the library in one fixed order (slow; for tests that must tell a
    scheduling defect from summation-order noise)."""
    check(native.lib().rd_set_deterministic(int(bool(on))), "rd_set_deterministic")


def get_deterministic():
    return bool(native.lib().rd_get_deterministic())


# ------------------------------------------------------------------------------------------ arithmetic mode of the conv kernels
_CONV_MATH = ["f32"]


def set_conv_math(mode):
    """'f32': exact fp32 MFMA (default).  'bf16x3': split-bf16 MFMA, ~4e-6 relative error (conv_b3.hip)."""
    code = {"f32": 0, "bf16x3": 1}[mode]
    check(native.lib().rd_set_conv_math(code), "rd_set_conv_math")
    _CONV_MATH[0] = mode


def get_conv_math():
    return _CONV_MATH[0]


def probe_mfma_bf16(iters=4000, waves_per_simd=2, repeats=5):
    """Sustained rate (TFLOP/s) of a bare bf16 MFMA loop on random operands on this chip right now (rd_probe_mfma_bf16): best of
    `repeats` launches of ~1 ms each, HIP events on the current stream."""
    out = torch.zeros(1, dtype=f32, device="cuda")
    flops = ctypes.c_double(0.0)
    best = 0.0
    for _ in range(repeats + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(native.lib().rd_probe_mfma_bf16(int(iters), int(waves_per_simd), _p(out), flops, _stream()), "rd_probe_mfma_bf16")
        e1.record()
        e1.synchronize()
        best = max(best, flops.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def set_mfma_terms(terms):
    """3 (default): bf16x3 products.  1: plain bf16 products (hi * hi only), fp32 accumulate and storage -- the `--use_amp` arithmetic
    (only meaningful with set_conv_math('bf16x3'))."""
    check(native.lib().rd_set_mfma_terms(int(terms)), "rd_set_mfma_terms")


def get_mfma_terms():
    return native.lib().rd_get_mfma_terms()


# ------------------------------------------------------------------------------------------ CenterHead loss (all heads, fused)
def center_loss_fwd(cfg, maps, heatmaps, inds, masks, target_boxes, gt_box):
    """maps (B, H, W, NO); stacked targets of rd_center_targets.  -> (out (4*nh + 1,), scale (4*nh,), ws) -- see rd_center_loss_fwd."""
    _chk(maps, f32, "maps", 4); _chk(heatmaps, f32, "heatmaps", 4); _chk(target_boxes, f32, "target_boxes", 4); _chk(gt_box, f32, "gt_box", 4)
    i64 = torch.int64
    _chk(inds, i64, "inds", 3); _chk(masks, i64, "masks", 3)
    nh, B, Kk = cfg.n_heads, cfg.B, cfg.K
    if tuple(maps.shape) != (B, cfg.H, cfg.W, cfg.NO) or tuple(heatmaps.shape) != (B, cfg.n_ch, cfg.H, cfg.W):
        raise RuntimeError(f"center_loss: maps {tuple(maps.shape)} / heatmaps {tuple(heatmaps.shape)} do not match the configuration")
    if tuple(inds.shape) != (nh, B, Kk) or tuple(masks.shape) != (nh, B, Kk) or tuple(target_boxes.shape[:3]) != (nh, B, Kk) or tuple(gt_box.shape[:3]) != (nh, B, Kk):
        raise RuntimeError("center_loss: target tensors must be (n_heads, B, K, ...)")
    out = torch.empty(4 * nh + 1, dtype=f32, device=maps.device)
    scale = torch.empty(4 * nh, dtype=f32, device=maps.device)
    ws = torch.empty(int(native.lib().rd_center_loss_ws_floats(cfg)), dtype=f32, device=maps.device)
    check(native.lib().rd_center_loss_fwd(cfg, _p(maps), _p(heatmaps), _p(inds), _p(masks), _p(target_boxes), target_boxes.shape[3],
                                          _p(gt_box), gt_box.shape[3], _p(out), _p(scale), _p(ws), _stream()), "rd_center_loss_fwd")
    return out, scale, ws


def center_loss_bwd(cfg, maps, heatmaps, inds, masks, scale, ws, grad_loss):
    _chk(grad_loss, f32, "grad_loss")
    if grad_loss.numel() != 1:
        raise RuntimeError("center_loss_bwd: grad_loss must hold one value")
    grad = torch.empty_like(maps)
    check(native.lib().rd_center_loss_bwd(cfg, _p(maps), _p(heatmaps), _p(inds), _p(masks), _p(scale), _p(ws), _p(grad_loss), _p(grad),
                                          _stream()), "rd_center_loss_bwd")
    return grad
