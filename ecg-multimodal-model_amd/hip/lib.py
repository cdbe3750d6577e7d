"""ctypes binding of libecgmm_hip.so (C ABI declared in include/ecgmm.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  Importing this module never
touches the GPU; the first call of :func:`lib` loads the shared object and fails loudly when it is
missing (build it with ``python __graft_entry__.py`` or ``make -C ecg-multimodal-model_amd/csrc``).
"""
import ctypes as C
import os

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
RESNET18_NPARAMS, RESNET18_NBUFFERS = 62, 60
RESNET1D_NPARAMS, RESNET1D_NBUFFERS = 52, 27

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ECGMM_LIB: A/B another build of the same library (kernel experiments); default = the in-tree build
LIB_PATH = os.environ.get("ECGMM_LIB") or os.path.join(_PKG_DIR, "libecgmm_hip.so")

vp, i32, i64, u64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("N", "H", "W", "Cin", "Cout", "R", "S", "stride", "pad_h", "pad_w")]


class ResNet18Desc(C.Structure):
    _fields_ = [("N", i32), ("H", i32), ("W", i32), ("out_dim", i32), ("dtype", i32), ("training", i32),
                ("bn_momentum", f32), ("bn_eps", f32)]


class ResNet1DDesc(C.Structure):
    _fields_ = [("N", i32), ("cin", i32), ("L", i32), ("num_classes", i32), ("dtype", i32), ("training", i32),
                ("bn_momentum", f32), ("bn_eps", f32), ("dropout_p", f32), ("seed", u64), ("offset", u64)]


class HeadDesc(C.Structure):
    _fields_ = [("B", i32), ("dim", i32 * 3), ("hidden", i32), ("num_classes", i32), ("training", i32),
                ("ln_eps", f32), ("dropout_p", f32), ("seed", u64), ("offset", u64)]


P = C.POINTER
# name -> (restype, argtypes).  Must list every function declared in include/ecgmm.h
# (tests/test_abi.py parses the header and checks the two agree).
SIGNATURES = {
    "ecgmm_version": (i32, []),
    "ecgmm_last_error": (C.c_char_p, []),
    "ecgmm_resnet18_fwd_workspace": (sz, [P(ResNet18Desc)]),
    "ecgmm_resnet18_bwd_workspace": (sz, [P(ResNet18Desc)]),
    "ecgmm_resnet18_forward": (i32, [P(ResNet18Desc), vp, P(vp), P(vp), vp, vp, sz, vp]),
    "ecgmm_resnet18_backward": (i32, [P(ResNet18Desc), vp, vp, P(vp), P(vp), vp, vp, sz, i32, i32, vp]),
    "ecgmm_head_fwd_workspace": (sz, [P(HeadDesc)]),
    "ecgmm_head_bwd_workspace": (sz, [P(HeadDesc)]),
    "ecgmm_head_forward": (i32, [P(HeadDesc), P(vp), P(vp), P(vp), vp, vp, vp, sz, vp]),
    "ecgmm_head_backward": (i32, [P(HeadDesc), P(vp), P(vp), P(vp), P(vp), vp, P(vp), vp, vp, sz, vp]),
    "ecgmm_conv_halo_enable": (i32, [i32]),
    "ecgmm_conv_halo_cus": (i32, [i32]),
    "ecgmm_conv_halo_w4": (i32, [i32]),
    "ecgmm_conv_halo_stagger": (i32, [i32]),
    "ecgmm_conv_halo_stream": (i32, [i32]),
    "ecgmm_conv_halo_pingpong": (i32, [i32]),
    "ecgmm_conv_wgrad_pingpong": (i32, [i32]),
    "ecgmm_conv_wgrad_ring_enable": (i32, [i32]),
    "ecgmm_side_wgrad": (i32, [i32]),
    "ecgmm_side_defer_join": (i32, [i32]),
    "ecgmm_resnet1d_side_wgrad": (i32, [i32]),
    "ecgmm_side_wait": (i32, [vp]),
    "ecgmm_side_stream": (vp, []),
    "ecgmm_side_fork": (i32, [vp]),
    "ecgmm_resnet1d_fwd_workspace": (sz, [P(ResNet1DDesc)]),
    "ecgmm_resnet1d_bwd_workspace": (sz, [P(ResNet1DDesc)]),
    "ecgmm_resnet1d_forward": (i32, [P(ResNet1DDesc), vp, P(vp), P(vp), vp, vp, sz, vp]),
    "ecgmm_resnet1d_backward": (i32, [P(ResNet1DDesc), vp, vp, P(vp), P(vp), vp, vp, sz, i32, i32, vp]),
    "ecgmm_nchw_to_nhwc": (i32, [i32, vp, vp, i32, i32, i64, vp]),
    "ecgmm_nhwc_to_nchw": (i32, [i32, vp, vp, i32, i32, i64, vp]),
    "ecgmm_cast": (i32, [i32, vp, vp, i64, vp]),
    "ecgmm_uncast": (i32, [i32, vp, vp, i64, vp]),
    "ecgmm_pack_conv_weight": (i32, [i32, vp, vp, vp, i32, i32, i32, vp]),
    "ecgmm_conv_stats_rows": (i32, [i64]),
    "ecgmm_conv_fwd": (i32, [i32, P(ConvDesc), vp, vp, vp, vp, vp, i32, vp]),
    "ecgmm_conv_fwd_wgrows": (i32, [i32, P(ConvDesc), vp, vp, vp, vp, vp, vp, i32, vp]),
    "ecgmm_conv_bwd_data": (i32, [i32, P(ConvDesc), vp, vp, vp, vp, vp]),
    "ecgmm_conv_bwd_weight_workspace": (sz, [i32, P(ConvDesc)]),
    "ecgmm_conv_bwd_weight": (i32, [i32, P(ConvDesc), vp, vp, vp, i32, vp, sz, vp]),
    "ecgmm_stem_packed_elems": (sz, [i32, i32]),
    "ecgmm_stem_stats_rows": (i32, [i32, i32, i32, i32, i32]),
    "ecgmm_stem_pack": (i32, [i32, vp, vp, i32, i32, vp]),
    "ecgmm_stem_fwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ecgmm_stem_wg_stats_rows": (i32, [i32, i32, i32, i32, i32]),
    "ecgmm_stem_fwd_wgrows": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ecgmm_stem_stats_only_rows": (i32, [i32, i32, i32, i32, i32]),
    "ecgmm_stem_stats_only": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ecgmm_stem_pool_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ecgmm_stem_pool_bwd_workspace": (sz, [i32, i32, i32, i32]),
    "ecgmm_stem_pool_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]),
    "ecgmm_stem_bwd_weight_workspace": (sz, [i32, i32, i32, i32, i32]),
    "ecgmm_stem_bwd_weight": (i32, [i32, vp, vp, vp, i32, vp, sz, i32, i32, i32, i32, i32, vp]),
    "ecgmm_col_stats_rows": (i32, [i32, i64, i32]),
    "ecgmm_col_stats": (i32, [i32, vp, i64, i32, vp, vp]),
    "ecgmm_bn_finalize": (i32, [vp, i32, i32, f64, vp, vp, vp, vp, vp, f32, f32, vp, vp]),
    "ecgmm_bn_eval_coef": (i32, [i32, vp, vp, vp, vp, f32, vp, vp]),
    "ecgmm_bn_act": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, vp, i64, i32, vp]),
    "ecgmm_bn_act_from_rows": (i32, [i32, vp, vp, i32, f64, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i32, i32, vp, i64, i32, vp]),
    "ecgmm_bn_bwd_scratch": (sz, [i32, i64, i32]),
    "ecgmm_bn_bwd": (i32, [i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp]),
    "ecgmm_conv_bwd_data_with_downsample": (i32, [i32, P(ConvDesc), vp, vp, vp, vp, vp, vp, vp]),
    "ecgmm_conv_bwd_data_bnred": (i32, [i32, P(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, P(i32), vp]),
    "ecgmm_conv_bwd_data_bnred_rows": (i32, [i32, P(ConvDesc)]),
    "ecgmm_bn_bwd_from_rows": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp, vp]),
    "ecgmm_bnrelu_maxpool": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ecgmm_maxpool_relu_bwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ecgmm_bn_fuse_min_pixels": (i32, [i64]),
    "ecgmm_bn_fold": (i32, [i32]),
    "ecgmm_tl_enable": (i32, [i32]),
    "ecgmm_tl_mark": (i32, [i32, vp]),
    "ecgmm_tl_collect": (i32, [i32, vp, vp]),
    "ecgmm_stem_recompute": (i32, [i32]),
    "ecgmm_pool_bn_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "ecgmm_avgpool": (i32, [i32, vp, vp, i32, i32, i32, vp, vp]),
    "ecgmm_bcast_rows": (i32, [i32, vp, vp, i32, i32, i32, f32, vp]),
    "ecgmm_se_gate_grad": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ecgmm_linear_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ecgmm_linear_bwd_scratch": (sz, [i32, i32, i32]),
    "ecgmm_linear_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "ecgmm_act_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "ecgmm_layernorm_fwd": (i32, [P(vp), P(i32), i32, vp, vp, vp, vp, vp, vp, i32, f32, vp]),
    "ecgmm_layernorm_bwd_scratch": (sz, [i32, i32]),
    "ecgmm_layernorm_bwd": (i32, [P(vp), P(i32), i32, vp, vp, vp, vp, P(vp), i32, vp, vp, vp, i32, vp, vp]),
    "ecgmm_varloss_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "ecgmm_varloss_bwd": (i32, [vp, i32, i32, vp, vp, i32, vp, i32, vp]),
    "ecgmm_ce_fwd": (i32, [vp, vp, i32, i32, i32, f32, f32, vp, vp, vp]),
    "ecgmm_ce_bwd": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
    "ecgmm_ce_plus_fwd": (i32, [vp, vp, i32, i32, vp, f32, vp, vp, vp]),
    "ecgmm_ce_plus_bwd": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, f32, vp]),
    "ecgmm_dropout_fwd": (i32, [vp, vp, vp, i64, f32, u64, u64, vp]),
    "ecgmm_dropout_bwd": (i32, [vp, vp, vp, i64, f32, vp]),
    "ecgmm_adam": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, f32, vp]),
    "ecgmm_axpby": (i32, [f32, vp, f32, vp, i64, vp]),
    "ecgmm_signal_preprocess_workspace": (sz, [i32, i32, i32]),
    "ecgmm_signal_preprocess": (i32, [vp, vp, i32, i32, vp, vp, i32, P(f64), P(f64), P(f64), i32, vp, sz, vp]),
    "ecgmm_glu_fwd": (i32, [vp, vp, i64, i32, vp]),
    "ecgmm_glu_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "ecgmm_sparsemax_fwd": (i32, [vp, vp, i64, i32, vp]),
    "ecgmm_sparsemax_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "ecgmm_ew": (i32, [i32, vp, vp, vp, i64, f32, vp]),
    "ecgmm_split_cols": (i32, [vp, vp, vp, i64, i32, i32, i32, vp]),
    "ecgmm_split_cols_bwd": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "ecgmm_bn_small_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp]),
    "ecgmm_bn_small_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ecgmm_entropy_fwd": (i32, [vp, vp, i64, i32, f32, vp]),
    "ecgmm_entropy_bwd": (i32, [vp, vp, vp, i64, i32, f32, vp]),
    "ecgmm_image_resize_tables_bytes": (sz, [i32, i32, i32, i32]),
    "ecgmm_image_resize_tables": (i32, [i32, i32, i32, i32, vp, sz]),
    "ecgmm_image_transform": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, sz, P(f32), P(f32), vp]),
    "ecgmm_prof_enable": (i32, [i32]),
    "ecgmm_prof_pause": (i32, [i32]),
    "ecgmm_prof_collect": (i32, [i32, P(f64), P(f64), P(f64), P(i64)]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libecgmm_hip.so not found at {LIB_PATH}: the HIP extension is the only compute path "
                "(no CPU fallback). Build it with `python __graft_entry__.py` or "
                "`make -C ecg-multimodal-model_amd/csrc`.")
        # torch ships its own libamdhip64: import it first so that this library binds to the SAME HIP
        # runtime instance (two runtimes in one process do not see each other's device context)
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.ecgmm_version() != 100:
            raise RuntimeError(f"libecgmm_hip.so version {h.ecgmm_version()} != 100 (stale build?)")
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ecgmm_last_error().decode(errors="replace")
        raise RuntimeError(f"ecgmm {what} failed (code {rc}): {msg}")
