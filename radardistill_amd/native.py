"""ctypes binding of librdamd.so (the C ABI declared in include/rdamd.h).

The library is built in-tree by `build()` (hipcc, gfx950 only) and loaded lazily by `lib()`.
There is NO fallback: if the shared object is missing or a call returns non-zero, a RuntimeError is
raised (the reference ops raise c10::Error, pcdet/ops/basicblock/src/cuda/modulated_deform_conv_cuda.cu:39-73).
"""
import ctypes
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "librdamd.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "rdamd.h")

c_int, c_i64, c_f32, c_f64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class ConvIndex(ctypes.Structure):
    """rd_conv_index of include/rdamd.h."""
    _fields_ = [("mode", c_int), ("nbr", c_vp), ("B", c_int), ("Hin", c_int), ("Win", c_int), ("Hout", c_int),
                ("Wout", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad", c_int), ("flip", c_int),
                ("samp_idx", c_vp), ("samp_w", c_vp)]


class LayoutJob(ctypes.Structure):
    """rd_layout_job of include/rdamd.h."""
    _fields_ = [("src", c_vp), ("dst", c_vp), ("Cout", c_int), ("Cin", c_int), ("taps", c_int), ("kind", c_int)]


class PackJob(ctypes.Structure):
    """rd_pack_job of include/rdamd.h."""
    _fields_ = [("src", c_vp), ("dst", c_vp), ("numel", c_i64)]


class CenterLossCfg(ctypes.Structure):
    """rd_center_loss_cfg of include/rdamd.h."""
    _fields_ = [("B", c_int), ("H", c_int), ("W", c_int), ("NO", c_int), ("n_heads", c_int), ("n_ch", c_int), ("K", c_int),
                ("hm_c0", c_int), ("c0_center", c_int), ("c0_z", c_int), ("c0_dim", c_int), ("c0_rot", c_int), ("c0_vel", c_int), ("c0_iou", c_int),
                ("head_of_ch", c_int * 16), ("code_w", c_f32 * 10), ("cls_w", c_f32), ("loc_w", c_f32),
                ("stride", c_f32), ("vs_x", c_f32), ("vs_y", c_f32), ("org_x", c_f32), ("org_y", c_f32)]


class TargetCfg(ctypes.Structure):
    """rd_target_cfg of include/rdamd.h."""
    _fields_ = [("n_classes", c_int), ("n_heads", c_int), ("n_channels", c_int), ("head_of_class", c_int * 16),
                ("local_of_class", c_int * 16), ("chan_off", c_int * 8), ("pcr0", c_f32), ("pcr1", c_f32), ("vs0", c_f32),
                ("vs1", c_f32), ("stride", c_int), ("fx", c_int), ("fy", c_int), ("max_objs", c_int), ("min_radius", c_int),
                ("overlap", c_f32)]


# name -> (restype, argtypes).  Must list EVERY symbol declared in include/rdamd.h (tests check this).
_P = c_vp
SIGNATURES = {
    "rd_last_error": (ctypes.c_char_p, []),
    "rd_abi_version": (c_int, []),
    "rd_device_ok": (c_int, []),
    "rd_stream_fork": (c_int, [_P, _P]),
    "rd_set_deterministic": (c_int, [c_int]),
    "rd_get_deterministic": (c_int, []),
    "rd_rankgrid_bytes": (c_i64, [c_i64]),
    "rd_voxelize": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32, _P, _P, _P]),
    "rd_rankgrid_coords": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "rd_rankgrid_from_coords": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_rankgrid_downsample": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_rankgrid_downsample_grid": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_nbr_subm": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_nbr_strided": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_nbr_strided_T": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "rd_vfe_pillar_mean": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P]),
    "rd_vfe_linear_stats": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "rd_vfe_linear_bn_relu_max": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "rd_vfe_backward": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "rd_vfe_group_ws_bytes": (c_i64, [c_int]),
    "rd_vfe_group": (c_int, [_P, c_int, c_int, _P, _P, _P, c_i64, _P]),
    "rd_vfe_seg_stats": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "rd_vfe_seg_max": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "rd_conv_fwd": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P, c_int, c_int, ctypes.POINTER(ConvIndex), _P, _P, _P, c_int, _P, _P]),
    "rd_conv_bn_act_fwd": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, _P, c_int, c_int, ctypes.POINTER(ConvIndex), _P, _P, _P, c_f32, c_f32, _P, _P,
                                   _P, c_int, _P, _P, _P, _P, _P]),
    "rd_conv_bn_act_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, c_int, c_int,
                                   ctypes.POINTER(ConvIndex), _P, ctypes.POINTER(ConvIndex), _P, _P, _P, _P, _P, _P, _P]),
    "rd_probe_mfma_bf16": (c_int, [c_int, c_int, _P, _P, _P]),
    "rd_geometry_begin": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32, _P, _P, c_int, _P, _P, _P]),
    "rd_geometry_finish": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_set_conv_math": (c_int, [c_int]),
    "rd_get_conv_math": (c_int, []),
    "rd_set_mfma_terms": (c_int, [c_int]),
    "rd_get_mfma_terms": (c_int, []),
    "rd_split_bf16": (c_int, [_P, c_i64, _P, _P]),
    "rd_weight_layout_split": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "rd_weight_layout_multi": (c_int, [_P, c_int, _P]),
    "rd_weight_layout_split_items": (c_int, [c_int, c_int, c_int, c_int]),
    "rd_weight_layout_split_multi": (c_int, [_P, _P, _P, c_int, _P]),
    "rd_center_loss_ws_floats": (c_i64, [_P]),
    "rd_center_loss_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, _P, _P, _P]),
    "rd_center_loss_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rd_conv_fwd_split": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, c_int, c_int, ctypes.POINTER(ConvIndex), _P, _P, _P, c_int, _P, _P]),
    "rd_conv_wgrad_split": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, ctypes.POINTER(ConvIndex), _P, _P]),
    "rd_conv_dgrad": (c_int, [_P, c_int, c_int, _P, c_int, _P, c_int, c_int, ctypes.POINTER(ConvIndex), _P]),
    "rd_conv_wgrad": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, ctypes.POINTER(ConvIndex), _P, _P]),
    "rd_weight_layout": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "rd_colsum": (c_int, [_P, c_i64, c_int, _P, _P]),
    "rd_bn_stats": (c_int, [_P, c_i64, c_int, _P, _P]),
    "rd_bn_train_fwd": (c_int, [_P, c_i64, c_int, _P, _P, _P, c_f32, c_f32, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_bn_finalize": (c_int, [_P, c_i64, c_int, _P, _P, c_f32, c_f32, _P, _P, _P, _P, _P, _P, _P]),
    "rd_affine_act": (c_int, [_P, c_i64, c_int, _P, _P, _P, c_int, _P, _P]),
    "rd_bn_bwd": (c_int, [_P, _P, _P, c_i64, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "rd_cat2_rows": (c_int, [_P, c_int, _P, c_int, c_i64, _P, _P]),
    "rd_split2_rows": (c_int, [_P, c_i64, c_int, c_int, _P, _P, _P]),
    "rd_bn_train_fwd_sync": (c_int, [_P, c_i64, c_int, _P, _P, _P, c_f32, c_f32, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_bn_finalize_sync": (c_int, [_P, c_int, _P, _P, c_f32, c_f32, _P, _P, _P, _P, _P, _P, _P]),
    "rd_bn_bwd_reduce": (c_int, [_P, _P, _P, c_i64, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    "rd_bn_bwd_apply": (c_int, [_P, _P, _P, c_i64, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_vfe_backward_reduce": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "rd_vfe_backward_weight": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "rd_rows_to_dense": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_dense_to_rows": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_dcn_prep": (c_int, [_P, c_int, _P, c_int, c_int] + [c_int] * 10 + [_P, _P, _P]),
    "rd_dcn_columns": (c_int, [_P, c_i64, c_int, _P, _P, c_i64, c_int, _P, _P]),
    "rd_dcn_bwd_data": (c_int, [_P, c_int, _P, _P, c_int, _P, c_int, c_int] + [c_int] * 10 + [_P, _P, c_int, _P, c_int, _P]),
    "rd_afd_ws_bytes": (c_i64, [c_i64]),
    "rd_afd_fwd": (c_int, [_P, _P, _P, c_i64, c_int, c_int, _P, _P, _P, _P, c_i64, _P]),
    "rd_afd_fwd_bf16": (c_int, [_P, _P, _P, c_i64, c_int, c_int, _P, _P, _P, _P, c_i64, _P]),
    "rd_afd_bwd": (c_int, [_P, _P, _P, c_i64, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_pfd_fwd": (c_int, [_P, _P, _P, _P, c_i64, c_int, _P, _P, c_int, _P, _P, _P, _P, c_i64, _P]),
    "rd_pfd_bwd": (c_int, [_P, _P, _P, _P, c_i64, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_boxes_aligned_overlap_bev": (c_int, [c_int, _P, _P, _P, _P]),
    "rd_opt_chunk_elems": (c_int, []),
    "rd_pack_grads_list": (c_int, [_P, c_int, _P]),
    "rd_pack_grads": (c_int, [_P, _P, c_int, _P, _P]),
    "rd_grad_norm": (c_int, [_P, _P, c_int, c_f32, _P, _P, c_i64, _P, c_f32, _P, _P, _P, _P]),
    "rd_grad_presence": (c_int, [_P, c_int, _P, _P]),
    "rd_adam_step": (c_int, [_P, _P, c_int, c_f64, c_f64, c_f64, c_f64, c_f64, c_int, _P, _P, _P, c_f32, _P, c_int, _P, _P, _P]),
    "rd_dwconv_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_dwconv_wgrad_ws_bytes": (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    "rd_dwconv_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_i64, _P]),
    "rd_center_targets": (c_int, [_P, c_int, c_int, c_int, ctypes.POINTER(TargetCfg), _P, _P, _P, _P, _P, _P]),
    "rd_voxelize_hard_ws_bytes": (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    "rd_voxelize_hard": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_int, c_int, c_i64,
                                 _P, _P, _P, _P, _P, c_i64, _P]),
    "rd_pillar_vfe_stats": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32,
                                    _P, _P]),
    "rd_pillar_vfe_max": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32,
                                  _P, _P, _P, _P]),
    "rd_pillar_decorate": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_int, _P, _P]),
    "rd_pfn_pool_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "rd_pfn_pool_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "rd_gelu_grn_fwd": (c_int, [_P, c_int, c_i64, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_gelu_grn_bwd": (c_int, [_P, _P, _P, _P, c_int, c_i64, c_int, _P, _P, _P, _P, _P, _P]),
    "rd_nms_ws_bytes": (c_i64, [c_int]),
    "rd_nms_bev": (c_int, [c_int, _P, c_f32, _P, c_i64, _P, _P, _P]),
    "rd_boxes_overlap_bev": (c_int, [c_int, _P, c_int, _P, _P, _P]),
    "rd_lp_cast": (c_int, [_P, c_i64, c_int, c_int, c_f32, _P, c_int, c_int, _P]),
    "rd_lp_uncast": (c_int, [_P, c_i64, c_int, c_f32, _P, _P]),
    "rd_lp_amax": (c_int, [_P, c_i64, _P, _P]),
    "rd_lp_quant_weights": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "rd_lp_conv": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "rd_layernorm_fwd": (c_int, [_P, c_i64, c_int, _P, _P, c_f32, _P, _P, _P, _P]),
    "rd_layernorm_bwd": (c_int, [_P, _P, c_i64, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "rd_nconv_fwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "rd_nconv_dgrad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "rd_nconv_dgrad_bn": (c_int, [_P] * 8 + [c_int] * 5 + [_P, _P, _P, _P, c_int, _P, _P, _P]),
    "rd_nconv_wgrad": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def csrc_sha():
    """Hash of the kernel sources (csrc/*.hip, *.hpp): profiles/ summaries record it so that a counter figure is never quoted for
    code it was not collected from (bench.py's roofline.traffic)."""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into csrc/librdamd.so (hipcc cross-compiles without a GPU)."""
    if not force and not _stale():
        build_fast()
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "obj"), exist_ok=True)
    for src in sources():
        obj = os.path.join(CSRC, "obj", os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        # an object depends on its source, the public header and EVERY csrc/*.hpp (common.hpp, conv_common.hpp, iou3d_dev.hpp ...):
        # cheaper than tracking includes per file, and an edited helper header can never relink a stale object
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                [os.path.getmtime(src), os.path.getmtime(HEADER)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.hpp"))]):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    build_fast()
    return SO_PATH


# ------------------------------------------------------------------------------------------ fast caller (CPython extension)
# ctypes spends ~1.8 us converting the 17 arguments of a convolution call; a training step makes ~730 library calls.  build() therefore
# also GENERATES csrc/_rdcall.c from SIGNATURES -- one METH_FASTCALL wrapper per entry point of include/rdamd.h: ints / floats parsed
# directly, a pointer argument may be None (NULL), a Python int (device or host address) or a ctypes Structure / array instance (its own
# storage, through the buffer protocol), the GIL released around the call -- and compiles it with gcc against librdamd.so.  lib()
# hands out these wrappers; the ctypes handle stays loaded for the symbol check of the tests.  Same library, same entry points, same
# error behaviour: only the argument marshalling moved from ctypes into 30 lines of C per function.
FAST_SO = os.path.join(CSRC, "_rdcall.so")
FAST_C = os.path.join(CSRC, "obj", "_rdcall.c")


def _c_kind(t):
    if t in (c_int,):
        return "int"
    if t in (c_i64,):
        return "i64"
    if t is c_f32:
        return "f32"
    if t is c_f64:
        return "f64"
    return "ptr"          # c_void_p and POINTER(struct)


# Two loops over the ~480 parameters that ran in the interpreter every step (train.FusedAdamOneCycle._fill_table / zero_grad:
# 0.6-0.9 ms + 0.4 ms of a 15.7 ms host-bound step): plain CPython C API, no library call.
_HOST_HELPERS = r"""
/* grad_ptrs(params: list, f32: torch.float32, out: writable int64 buffer of len(params)) -> number of entries of `out` that changed,
 * or -2 - i when parameter i's gradient is not a contiguous fp32 tensor (the caller converts it and calls again).
 * out[i] = params[i].grad.data_ptr(), 0 for a parameter without a gradient. */
static PyObject *w_grad_ptrs(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    static PyObject *s_grad, *s_dtype, *s_contig, *s_ptr;
    if (!s_grad) {
        s_grad = PyUnicode_InternFromString("grad"); s_dtype = PyUnicode_InternFromString("dtype");
        s_contig = PyUnicode_InternFromString("is_contiguous"); s_ptr = PyUnicode_InternFromString("data_ptr");
    }
    if (nargs != 3 || !PyList_Check(args[0])) { PyErr_SetString(PyExc_TypeError, "grad_ptrs(list, dtype, int64 buffer)"); return NULL; }
    const Py_ssize_t n = PyList_GET_SIZE(args[0]);
    Py_buffer b;
    if (PyObject_GetBuffer(args[2], &b, PyBUF_WRITABLE) != 0) return NULL;
    if (b.len < (Py_ssize_t)(n * sizeof(int64_t))) { PyBuffer_Release(&b); PyErr_SetString(PyExc_ValueError, "grad_ptrs: buffer too small"); return NULL; }
    int64_t *out = (int64_t *)b.buf;
    long changed = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *g = PyObject_GetAttr(PyList_GET_ITEM(args[0], i), s_grad);
        if (!g) { PyBuffer_Release(&b); return NULL; }
        int64_t v = 0;
        if (g != Py_None) {
            PyObject *dt = PyObject_GetAttr(g, s_dtype);
            if (!dt) { Py_DECREF(g); PyBuffer_Release(&b); return NULL; }
            const int dtype_ok = dt == args[1];
            Py_DECREF(dt);
            int contig = 0;
            if (dtype_ok) {
                PyObject *c = PyObject_CallMethodNoArgs(g, s_contig);
                if (!c) { Py_DECREF(g); PyBuffer_Release(&b); return NULL; }
                contig = PyObject_IsTrue(c);
                Py_DECREF(c);
            }
            if (!dtype_ok || contig != 1) { Py_DECREF(g); PyBuffer_Release(&b); return PyLong_FromLong(-2 - (long)i); }
            PyObject *pv = PyObject_CallMethodNoArgs(g, s_ptr);
            if (!pv) { Py_DECREF(g); PyBuffer_Release(&b); return NULL; }
            v = (int64_t)PyLong_AsLongLong(pv);
            Py_DECREF(pv);
            if (v == -1 && PyErr_Occurred()) { Py_DECREF(g); PyBuffer_Release(&b); return NULL; }
        }
        Py_DECREF(g);
        if (out[i] != v) { out[i] = v; ++changed; }
    }
    PyBuffer_Release(&b);
    return PyLong_FromLong(changed);
}

/* clear_grads(params: list): p.grad = None for every parameter */
static PyObject *w_clear_grads(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    static PyObject *s_grad;
    if (!s_grad) s_grad = PyUnicode_InternFromString("grad");
    if (nargs != 1 || !PyList_Check(args[0])) { PyErr_SetString(PyExc_TypeError, "clear_grads(list)"); return NULL; }
    const Py_ssize_t n = PyList_GET_SIZE(args[0]);
    for (Py_ssize_t i = 0; i < n; ++i)
        if (PyObject_SetAttr(PyList_GET_ITEM(args[0], i), s_grad, Py_None) != 0) return NULL;
    Py_RETURN_NONE;
}
"""


def _generate_fast_source():
    out = ["/* GENERATED by radardistill_amd/native.py from SIGNATURES -- do not edit. */", "#define PY_SSIZE_T_CLEAN", "#include <Python.h>",
           "#include <stdint.h>", '#include "rdamd.h"', "",
           "static int get_ptr(PyObject *o, void **out) {",
           "    if (o == Py_None) { *out = NULL; return 0; }",
           "    if (PyLong_Check(o)) { *out = PyLong_AsVoidPtr(o); return (*out == NULL && PyErr_Occurred()) ? -1 : 0; }",
           "    Py_buffer b;          /* ctypes Structure / array instance: its own storage */",
           "    if (PyObject_GetBuffer(o, &b, PyBUF_SIMPLE) == 0) { *out = b.buf; PyBuffer_Release(&b); return 0; }",
           "    return -1;", "}", ""]
    table = []
    for name, (res, args) in SIGNATURES.items():
        n = len(args)
        out.append(f"static PyObject *w_{name}(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {{")
        out.append(f'    if (nargs != {n}) {{ PyErr_Format(PyExc_TypeError, "{name}: expected {n} arguments, got %zd", nargs); return NULL; }}')
        call = []
        for i, t in enumerate(args):
            k = _c_kind(t)
            if k == "ptr":
                out.append(f'    void *a{i}; if (get_ptr(args[{i}], &a{i})) {{ if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "{name}: argument {i} is not a pointer (None, int or ctypes instance)"); return NULL; }}')
                call.append(f"a{i}")
            elif k == "int":
                out.append(f"    long a{i} = PyLong_AsLong(args[{i}]); if (a{i} == -1 && PyErr_Occurred()) return NULL;")
                call.append(f"(int)a{i}")
            elif k == "i64":
                out.append(f"    long long a{i} = PyLong_AsLongLong(args[{i}]); if (a{i} == -1 && PyErr_Occurred()) return NULL;")
                call.append(f"(int64_t)a{i}")
            else:
                out.append(f"    double a{i} = PyFloat_AsDouble(args[{i}]); if (a{i} == -1.0 && PyErr_Occurred()) return NULL;")
                call.append(f"({'float' if k == 'f32' else 'double'})a{i}")
        args_c = ", ".join(call)
        if res is ctypes.c_char_p:
            out.append(f"    const char *r = {name}({args_c});")
            out.append("    if (!r) Py_RETURN_NONE;")
            out.append("    return PyBytes_FromString(r);")
        else:
            ctype = "int64_t" if res is c_i64 else "int"
            out.append(f"    {ctype} r;")
            out.append("    Py_BEGIN_ALLOW_THREADS")
            out.append(f"    r = {name}({args_c});")
            out.append("    Py_END_ALLOW_THREADS")
            out.append("    return PyLong_FromLongLong((long long)r);")
        out.append("}")
        out.append("")
        table.append(f'    {{"{name}", (PyCFunction)(void (*)(void))w_{name}, METH_FASTCALL, NULL}},')
    out.append(_HOST_HELPERS)
    table.append('    {"grad_ptrs", (PyCFunction)(void (*)(void))w_grad_ptrs, METH_FASTCALL, NULL},')
    table.append('    {"clear_grads", (PyCFunction)(void (*)(void))w_clear_grads, METH_FASTCALL, NULL},')
    out.append("static PyMethodDef methods[] = {")
    out.extend(table)
    out.append("    {NULL, NULL, 0, NULL}};")
    out.append('static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_rdcall", "argument marshalling for librdamd.so", -1, methods};')
    out.append("PyMODINIT_FUNC PyInit__rdcall(void) { return PyModule_Create(&moddef); }")
    return "\n".join(out) + "\n"


def build_fast(force=False):
    """Generate and compile csrc/_rdcall.so (needs csrc/librdamd.so)."""
    import sysconfig
    src = _generate_fast_source()
    os.makedirs(os.path.dirname(FAST_C), exist_ok=True)
    old = open(FAST_C).read() if os.path.exists(FAST_C) else None
    if not force and old == src and os.path.exists(FAST_SO) and os.path.getmtime(FAST_SO) >= os.path.getmtime(SO_PATH):
        return FAST_SO
    with open(FAST_C, "w") as f:
        f.write(src)
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], "-I" + os.path.dirname(HEADER), FAST_C,
           "-o", FAST_SO, "-L" + CSRC, "-lrdamd", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"gcc failed on the generated caller:\n{r.stdout.decode(errors='replace')}")
    return FAST_SO


_LIB = None


_CTYPES = None


def ctypes_lib():
    """The ctypes handle of the library with typed entry points (tests: every symbol of the header is exported)."""
    global _CTYPES
    if _CTYPES is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here == header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _CTYPES = L
    return _CTYPES


def lib():
    """The library's entry points as fast callables (csrc/_rdcall.so, see build_fast).  Raises if either shared object has not been
    built: there is no CPU fallback and no slower second path."""
    global _LIB
    if _LIB is None:
        ctypes_lib()
        if not os.path.exists(FAST_SO):
            raise RuntimeError(f"{FAST_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        import importlib.machinery
        import importlib.util
        loader = importlib.machinery.ExtensionFileLoader("_rdcall", FAST_SO)
        spec = importlib.util.spec_from_loader("_rdcall", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        missing = [n for n in SIGNATURES if not hasattr(mod, n)]
        if missing:
            raise RuntimeError(f"{FAST_SO} is stale (no wrapper for {missing[:3]}...): rebuild")
        _LIB = mod
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().rd_last_error()
        raise RuntimeError(f"librdamd {what} failed (code {rc}): {msg.decode() if msg else ''}")


def header_symbols():
    """Function names declared in include/rdamd.h (used by the CPU test that the library exports them all)."""
    import re
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rd_[A-Za-z0-9_]+)\s*\(", text)))
