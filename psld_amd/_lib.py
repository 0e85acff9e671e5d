"""ctypes binding of libpsld_hip.so (the C ABI declared in include/psld_hip.h).

The product path has NO fallback: if the shared object is missing or a symbol is absent the
import of the compute path raises.  (Contrast the reference, which JIT-builds its ops at import
time: op/upfirdn2d.py:10-16.)  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C psld_amd/csrc`` and travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PSLD_HIP_LIB: load another build of the same ABI (kernel experiments); the default is the in-tree library
LIB_PATH = os.environ.get("PSLD_HIP_LIB") or os.path.join(_HERE, "libpsld_hip.so")

c_f32p = C.c_void_p
c_f64p = C.c_void_p
c_stream = C.c_void_p
c_ll = C.c_longlong


class Epilogue(C.Structure):
    """struct psld_epilogue (include/psld_hip.h)."""
    _fields_ = [
        ("alpha", C.c_float),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p),
        ("ld_rowbias", C.c_int),
        ("rows_per_img", C.c_int),
        ("residual", C.c_void_p),
        ("ld_residual", C.c_int),
        ("residual_stride_batch", C.c_longlong),
        ("out_scale", C.c_float),
        ("accumulate", C.c_int),
        ("gn_part", C.c_void_p),
        ("gn_hw", C.c_int),
        ("gn_fine", C.c_int),
    ]


class SdeParams(C.Structure):
    """struct psld_sde_params."""
    _fields_ = [
        ("beta_0", C.c_double), ("beta_1", C.c_double), ("nu", C.c_double), ("gamma", C.c_double),
        ("m_inv", C.c_double), ("numerical_eps", C.c_double), ("decomp_lower", C.c_int),
    ]


class EmCoeffs(C.Structure):
    """struct psld_em_coeffs."""
    _fields_ = [
        ("beta", C.c_double), ("m_inv", C.c_double), ("gamma", C.c_double), ("nu", C.c_double),
        ("m", C.c_double),
        ("c11", C.c_float), ("c12", C.c_float), ("c21", C.c_float), ("c22", C.c_float),
        ("dt", C.c_double), ("score_mode", C.c_int), ("probability_flow", C.c_int),
    ]


class SscsCoeffs(C.Structure):
    """struct psld_sscs_coeffs."""
    _fields_ = [(n, C.c_double) for n in ("a_xx", "a_xm", "a_mx", "a_mm", "c11", "c12", "c21", "c22")]


I, F, D, LL, P = C.c_int, C.c_float, C.c_double, C.c_longlong, C.c_void_p
EP = C.POINTER(Epilogue)

ABI_VERSION = 14   # PSLD_ABI_VERSION of include/psld_hip.h these signatures were written against

# name -> (restype, argtypes): every symbol include/psld_hip.h declares
SIGNATURES = {
    "psld_version": (I, []),
    "psld_last_error": (C.c_char_p, []),
    "psld_set_math_mode": (I, [I]),
    "psld_get_math_mode": (I, []),
    "psld_set_gn_bwd_kernel": (I, [I]),
    "psld_get_gn_bwd_kernel": (I, []),
    "psld_gemm_f32": (I, [I, I, I, I, I, P, I, LL, P, I, LL, P, I, LL, I, EP, P]),
    "psld_gemm_tn_splitk_f32": (I, [I, I, I, P, I, P, I, P, I, P]),
    "psld_conv2d_nhwc_f32": (I, [P, I, P, I, I, I, I, P, I, I, I, I, I, I, I, I, P, I, EP, P]),
    "psld_conv2d_workspace_bytes": (LL, [I, I, I, I]),
    "psld_conv2d_nhwc_ws_f32": (I, [P, I, P, I, I, I, I, P, I, I, I, I, I, I, I, I, P, I, EP, P, LL, P]),
    "psld_conv2d_wgrad_nhwc_f32": (I, [P, I, I, P, I, I, I, I, I, I, I, I, I, I, P, I, I, I, P]),
    "psld_conv3x3_frag_bytes": (LL, [I, I]),
    "psld_conv3x3_split_supported": (I, [I, I, I, I, I, I]),
    "psld_pack_conv3x3_frag": (I, [P, P, I, I, I, P]),
    "psld_pack_frag_batch": (I, [P, I, LL, P]),
    "psld_conv3x3_split_f32": (I, [P, I, P, I, I, I, I, P, I, P, I, EP, P, LL, P]),
    "psld_attn_fwd_split_supported": (I, [I, I]),
    "psld_attn_fwd_split_f32": (I, [P, P, P, I, I, I, I, F, P, I, P, P]),
    "psld_conv3x3_wino_frag_bytes": (LL, [I, I]),
    "psld_conv3x3_wino_supported": (I, [I, I, I, I, I, I]),
    "psld_pack_conv3x3_wino": (I, [P, P, I, I, I, P]),
    "psld_pack_wino_batch": (I, [P, I, LL, P]),
    "psld_conv3x3_wino_f32": (I, [P, I, P, I, I, I, I, P, I, P, I, EP, P]),
    "psld_conv3x3_wino_ksplit": (I, [I, I, I, I, I, I]),
    "psld_conv3x3_wino_ws_bytes": (LL, [I, I, I, I, I, I]),
    "psld_conv3x3_wino_ws_f32": (I, [P, I, P, I, I, I, I, P, I, P, I, EP, P, LL, P]),
    "psld_conv3x3_wino_gn_supported": (I, [I, I, I, I, I, I]),
    "psld_conv3x3_wino_gn_f32": (I, [P, I, P, P, P, I, P, P, I, I, I, I, P, I, P, I, EP, P]),
    "psld_conv3x3_wino_gn_ws_f32": (I, [P, I, P, P, P, I, P, P, I, I, I, I, P, I, P, I, EP, P, LL, P]),
    "psld_gn_apply_limb_nhwc": (I, [P, P, P, P, I, I, I, I, F, C.c_ulonglong, P, P]),
    "psld_limb_bytes": (LL, [LL, I]),
    "psld_f32_to_limb": (I, [P, LL, I, P, P]),
    "psld_limb_to_f32": (I, [P, LL, I, P, P]),
    "psld_conv3x3_limb_f32": (I, [P, I, P, I, I, I, I, P, I, P, I, EP, P, LL, P]),
    "psld_gemm_frag_bytes": (LL, [I, I]),
    "psld_gemm_split_supported": (I, [I, I, I, I]),
    "psld_pack_gemm_frag": (I, [P, P, I, I, LL, LL, P]),
    "psld_gemm_split_f32": (I, [P, I, P, I, I, P, I, P, I, EP, P, LL, P]),
    "psld_conv3x3_wgrad_split_supported": (I, [I, I, I, I, I]),
    "psld_conv3x3_wgrad_split_cout_tile": (I, [I]),
    "psld_conv3x3_wgrad_split_f32": (I, [P, I, I, P, I, P, I, I, I, I, P, I, I, I, P]),
    "psld_conv3x3_wgrad_xlimb_f32": (I, [P, I, I, P, I, P, I, I, I, I, P, I, I, I, P]),
    "psld_conv3x3_wgrad_wino_supported": (I, [I, I, I, I, I, I]),
    "psld_conv3x3_wgrad_wino_nsplit": (I, [I, I, I, I, I]),
    "psld_conv3x3_wgrad_wino_ws_bytes": (LL, [I, I, I]),
    "psld_conv3x3_wgrad_wino_f32": (I, [P, I, I, P, I, P, I, I, I, I, P, I, P, I, F, P]),
    "psld_gemm_tn_split_supported": (I, [I, I, I]),
    "psld_gemm_tn_split_f32": (I, [I, I, I, P, I, P, I, P, I, I, P, I, I, P]),
    "psld_bgemm_split_supported": (I, [I, I, I, I, I]),
    "psld_bgemm_split_f32": (I, [I, I, I, I, I, P, I, LL, P, I, LL, P, I, LL, I, F, P]),
    "psld_reduce_slabs_f32": (I, [P, I, LL, P, I, I, I, I, F, P]),
    "psld_pack_oihw_to_ohwi_f32": (I, [P, P, I, I, I, P]),
    "psld_pack_oihw_to_dgrad_f32": (I, [P, P, I, I, I, P]),
    "psld_nchw_to_nhwc_f32": (I, [P, P, I, I, I, P]),
    "psld_nhwc_to_nchw_f32": (I, [P, P, I, I, I, P]),
    "psld_gn_workspace_bytes": (LL, [I, I, I, I]),
    "psld_gn_stats_nhwc_f32": (I, [P, I, I, I, I, F, P, P, P, P, P, P, P, P]),
    "psld_gn_stats_from_partials_f32": (I, [P, I, I, I, I, I, F, P, P, P, P, P, P, P]),
    "psld_gn_apply_nhwc_f32": (I, [P, P, P, P, I, I, I, I, F, C.c_ulonglong, P, P]),
    "psld_gn_bwd_nhwc_f32": (I, [P, P, P, P, P, P, I, I, I, I, I, F, C.c_ulonglong, P, P, I, P, F, P, P, I, P, P]),
    "psld_gn_bwd_colsum_supported": (I, [I, I, I, I]),
    "psld_gn_bwd_team_rows": (I, [I, I, I, I]),
    "psld_gn_bwd_team_sync_bytes": (LL, []),
    "psld_gn_bwd_team_f32": (I, [P, P, P, P, P, P, I, I, I, I, I, F, C.c_ulonglong, P, P, I, P, F, P, P, I, P, P]),
    "psld_param_reduce2_f32": (I, [P, P, I, I, I, P, P, F, P]),
    "psld_param_reduce_batch_f32": (I, [P, I, I, P]),
    "psld_reduce_slabs_batch_units": (I, [LL, I, I, I]),
    "psld_reduce_slabs_batch_f32": (I, [P, I, LL, P]),
    "psld_upfirdn2d_f32": (I, [P, P, I, I, I, I, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "psld_fused_bias_act_f32": (I, [P, P, P, LL, I, I, I, F, F, P]),
    "psld_fused_bias_act_grad_f32": (I, [P, P, P, P, LL, I, I, I, I, F, F, P]),
    "psld_axpby_f32": (I, [P, F, P, F, P, LL, I, P]),
    "psld_silu_f32": (I, [P, P, LL, P]),
    "psld_silu_bwd_f32": (I, [P, P, P, LL, P]),
    "psld_colsum_workspace_bytes": (LL, [I, I, I]),
    "psld_colsum_f32": (I, [P, I, I, I, I, P, F, P, P]),
    "psld_bias_grad_f32": (I, [P, I, I, I, I, P, I, P, F, P, P]),
    "psld_bias_grad_seg_f32": (I, [P, I, I, I, I, P, P, P, F, P, P]),
    "psld_copy_batch_f32": (I, [P, I, LL, P]),
    "psld_copy2d_f32": (I, [P, I, P, I, LL, I, I, P]),
    "psld_im2col3x3_small_f32": (I, [P, I, I, I, I, I, I, I, I, I, P, I, P]),
    "psld_im2col3x3_f32": (I, [P, I, I, I, I, I, I, I, I, P, P]),
    "psld_col2im3x3_f32": (I, [P, I, I, I, I, I, I, I, I, P, P]),
    "psld_conv3x3_fewout_supported": (I, [I, I]),
    "psld_conv3x3_fewout_f32": (I, [P, P, P, P, I, I, I, I, I, P]),
    "psld_scale_copy2d_f32": (I, [P, I, P, I, LL, I, F, P]),
    "psld_softmax_rows_f32": (I, [P, P, LL, I, P]),
    "psld_softmax_rows_bwd_f32": (I, [P, P, P, LL, I, P]),
    "psld_time_embed_f32": (I, [P, P, P, I, I, I, P]),
    "psld_perturb_coeffs_f64": (I, [P, I, C.POINTER(SdeParams), D, D, P, P, P]),
    "psld_perturb_f32": (I, [P, P, P, P, C.POINTER(SdeParams), I, I, I, P, P, P, P]),
    "psld_reduce_workspace_bytes": (LL, [LL]),
    "psld_sqerr_loss_f32": (I, [P, P, LL, I, P, P, F, P, P]),
    "psld_em_step_f64": (I, [P, P, P, C.POINTER(EmCoeffs), I, I, I, P, P]),
    "psld_reverse_sde_f64": (I, [P, P, C.POINTER(EmCoeffs), I, I, I, P, P, P]),
    "psld_vp_score_loss": (I, [P, P, P, D, D, I, LL, I, I, P, P, F, P, P]),
    "psld_reverse_sde_rows_f64": (I, [P, P, P, C.POINTER(SdeParams), D, D, I, I, I, I, I, P, P, P, P]),
    "psld_sscs_analytic_f64": (I, [P, P, C.POINTER(SscsCoeffs), I, I, I, P, P]),
    "psld_sscs_score_step_f64": (I, [P, P, C.POINTER(EmCoeffs), I, I, I, P]),
    "psld_samples_to_uint8": (I, [P, P, I, I, I, I, I, P]),
    "psld_uint8_to_images_f32": (I, [P, P, P, I, I, I, I, I, P]),
    "psld_lincomb_f64": (I, [P, P, P, P, I, LL, P, P]),
    "psld_scaled_norm_sq_f64": (I, [P, P, I, P, P, D, D, LL, P, P, P]),
    "psld_vp_perturb_f32": (I, [P, P, P, D, D, I, LL, P, P, P]),
    "psld_vp_reverse_f64": (I, [P, P, P, D, D, D, I, I, LL, P, P, P]),
    "psld_mask_combine_f64": (I, [P, P, P, I, I, I, P, P]),
    "psld_guide_f64": (I, [P, P, D, D, I, I, I, P, P]),
    "psld_softmax_xent_f32": (I, [P, P, I, I, F, F, P, P, P, P]),
    "psld_f64_to_f32": (I, [P, P, LL, P]),
    "psld_f32_to_f64": (I, [P, P, LL, P]),
    "psld_grad_norm_f32": (I, [P, LL, P, P, P]),
    "psld_adam_ema_f32": (I, [P, P, P, P, P, LL, P, D, D, D, D, D, D, I, D, I, P, P, P, P, P]),
    "psld_adam_step_scalars": (None, [D, D, D, I, P]),
    "psld_adam_step_scalars_dev": (I, [D, D, D, I, P, P]),
    "psld_ema_f32": (I, [P, P, LL, D, P, P]),
}


def is_launch(name: str) -> bool:
    """Entry points that enqueue work on a stream (status-returning, ``hipStream_t stream`` last).  Queries
    (``*_supported``, ``*_bytes``) and the math-mode switch are host-only.  tests/test_abi_cpu.py checks this rule
    against the header's parameter names."""
    res, args = SIGNATURES[name]
    return res is I and bool(args) and args[-1] is P

PSLD_ERR_NUMERIC = 3

_lib = None


class PsldHipError(RuntimeError):
    pass


_proxy = None     # a stand-in that forwards every call to the loaded library (tools/hbm_in_situ.py sizes the launches with one)


def set_proxy(proxy):
    global _proxy
    _proxy = proxy


def load():
    """The loaded library (or the stand-in set by ``set_proxy``)."""
    if _proxy is not None:
        return _proxy
    return _lib if _lib is not None else load_real()


def load_real() -> C.CDLL:
    """Load libpsld_hip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PsldHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C psld_amd/csrc`. "
            "There is no CPU / PyTorch fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PsldHipError(f"libpsld_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.psld_version() != ABI_VERSION:
        raise PsldHipError(f"{LIB_PATH} has ABI version {lib.psld_version()}, this package binds version {ABI_VERSION}: "
                           "rebuild it (`make -C psld_amd/csrc`)")
    _lib = lib
    return lib


def check(status: int, what: str = ""):
    """0 -> ok; PSLD_ERR_NUMERIC -> ValueError (mirrors psld.py:171); anything else -> RuntimeError
    (mirrors TORCH_CHECK in op/upfirdn2d.cpp:8,15-16)."""
    if status == 0:
        return
    msg = load().psld_last_error().decode("utf-8", "replace")
    if status == PSLD_ERR_NUMERIC:
        raise ValueError(msg or "Numerical precision error.")
    raise RuntimeError(f"libpsld_hip {what} failed (status {status}): {msg}")
