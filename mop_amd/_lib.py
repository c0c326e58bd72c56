"""ctypes binding of libmopk.so (include/mopk.h).  No torch types cross this boundary."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOPK_LIB") or os.path.join(HERE, "libmopk.so")   # MOPK_LIB: dev builds (tools/build_variant.py)

MOPK_F32, MOPK_BF16 = 0, 1
PREC_FP32, PREC_BF16 = 0, 1
PATH_AUTO, PATH_GENERIC, PATH_FUSED = 0, 1, 2


class View4(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sb", C.c_int64), ("sh", C.c_int64), ("sn", C.c_int64)]


class View5(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sv", C.c_int64), ("sb", C.c_int64), ("sh", C.c_int64),
                ("sn", C.c_int64)]


_fp = C.c_void_p  # device pointers travel as void*


MAX_LENS = 4          # MOPK_MAX_LENS
DENSE_HIDDEN = 16     # MOPK_DENSE_HIDDEN


class EdgewiseExt(C.Structure):
    """MopkEdgewiseExt: dense gate head / S lens bank (generic path)."""
    _fields_ = [
        ("gate_mode", C.c_int32), ("use_k3", C.c_int32), ("n_lens", C.c_int32), ("lens_dil", C.c_int32 * MAX_LENS),
        ("lens_w", _fp), ("W1", _fp), ("b1", _fp), ("W3", _fp), ("b3", _fp), ("W2", _fp), ("b2", _fp),
        ("dlens_w", _fp), ("dW1", _fp), ("db1", _fp), ("dW3", _fp), ("db3", _fp), ("dW2", _fp), ("db2", _fp),
        ("n_extra", C.c_int32), ("row_extra", _fp), ("col_extra", _fp), ("d_row_extra", _fp), ("d_col_extra", _fp),
    ]


class EdgewiseArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("N", C.c_int32), ("dk", C.c_int32),
        ("V", C.c_int32), ("r", C.c_int32), ("io_dtype", C.c_int32), ("precision", C.c_int32),
        ("path", C.c_int32), ("save_for_backward", C.c_int32), ("beta_not", C.c_float),
        ("q", View5), ("k", View5), ("v0", View4), ("vL", View4),
        ("sqk", _fp), ("vs0", _fp), ("vsL", _fp), ("Wr", _fp), ("br", _fp), ("Wc", _fp), ("bc", _fp),
        ("chain_logit", _fp),
        ("y", View4), ("saved", _fp), ("workspace", _fp),
        ("dy", View4), ("dq", View5), ("dk_", View5), ("dv0", View4), ("dvL", View4),
        ("dsqk_part", _fp), ("dvs0_part", _fp), ("dvsL_part", _fp),
        ("dWr", _fp), ("dbr", _fp), ("dWc", _fp), ("dbc", _fp), ("dlogit_part", _fp),
        ("ext", C.POINTER(EdgewiseExt)),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
        ("mask", _fp), ("mask_sb", C.c_int64), ("mask_sh", C.c_int64), ("mask_si", C.c_int64),
    ]


class DualPathArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("N", C.c_int32), ("dk", C.c_int32), ("hops", C.c_int32),
        ("io_dtype", C.c_int32), ("precision", C.c_int32), ("path", C.c_int32), ("causal", C.c_int32),
        ("g_and", C.c_float), ("g_or", C.c_float), ("g_not", C.c_float), ("g_chain", C.c_float),
        ("beta_not", C.c_float),
        ("q1", View4), ("k1", View4), ("v1", View4), ("q2", View4), ("k2", View4), ("v2", View4),
        ("mask", _fp), ("mask_sb", C.c_int64), ("mask_sh", C.c_int64), ("mask_si", C.c_int64),
        ("chain_logit", _fp), ("y", View4), ("saved", _fp), ("workspace", _fp),
        ("dy", View4), ("dq1", View4), ("dk1", View4), ("dv1", View4), ("dq2", View4), ("dk2", View4),
        ("dv2", View4), ("dlogit_part", _fp),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
    ]


class QuartetArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("T", C.c_int32), ("dh", C.c_int32),
        ("io_dtype", C.c_int32), ("precision", C.c_int32), ("path", C.c_int32),
        ("use_quartet", C.c_int32), ("eps", C.c_float),
        ("q", View4), ("k", View4), ("v", View4), ("q2", View4), ("k2", View4),
        ("mixture", _fp), ("quartet_scale", _fp),
        ("add_mask", _fp), ("am_sb", C.c_int64), ("am_sh", C.c_int64), ("am_si", C.c_int64),
        ("y", View4), ("attn", _fp), ("saved", _fp), ("workspace", _fp),
        ("dy", View4), ("dq", View4), ("dk_", View4), ("dv", View4), ("dq2", View4), ("dk2", View4),
        ("dmixture_part", _fp), ("dqscale_part", _fp),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
    ]


class SdpaArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("N", C.c_int32), ("dk", C.c_int32),
        ("io_dtype", C.c_int32), ("precision", C.c_int32), ("path", C.c_int32), ("causal", C.c_int32),
        ("q", View4), ("k", View4), ("v", View4),
        ("mask", _fp), ("mask_sb", C.c_int64), ("mask_sh", C.c_int64), ("mask_si", C.c_int64),
        ("bias", _fp), ("bias_sb", C.c_int64), ("bias_sh", C.c_int64), ("bias_si", C.c_int64),
        ("y", View4), ("saved", _fp), ("workspace", _fp),
        ("dy", View4), ("dq", View4), ("dk_", View4), ("dv", View4),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
    ]


# every symbol include/mopk.h declares: name -> (restype, argtypes)
class CrossViewArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("N", C.c_int32), ("dk", C.c_int32),
        ("io_dtype", C.c_int32), ("precision", C.c_int32), ("path", C.c_int32), ("causal", C.c_int32),
        ("use_prior", C.c_int32), ("anchor_mode", C.c_int32), ("fixed_k_star", C.c_int32),
        ("t1", C.c_float), ("t2", C.c_float), ("prior_weight", C.c_float),
        ("q1", View4), ("k1", View4), ("v1", View4), ("q2", View4), ("k2", View4),
        ("mix", _fp), ("mask", _fp), ("mask_sb", C.c_int64), ("mask_sh", C.c_int64), ("mask_si", C.c_int64),
        ("y", View4), ("saved", _fp), ("workspace", _fp), ("k_star", _fp),
        ("dy", View4), ("dq1", View4), ("dk1", View4), ("dv1", View4), ("dq2", View4), ("dk2", View4),
        ("dmix_part", _fp), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
    ]


class LayerNormArgs(C.Structure):
    _fields_ = [
        ("rows", C.c_int64), ("dim", C.c_int32), ("x_dtype", C.c_int32), ("y_dtype", C.c_int32), ("p_dtype", C.c_int32),
        ("eps", C.c_float), ("x_ld", C.c_int64), ("y_ld", C.c_int64),
        ("x", _fp), ("gamma", _fp), ("beta", _fp), ("y", _fp), ("mean", _fp), ("rstd", _fp),
        ("dy", _fp), ("dres", _fp), ("dx", _fp), ("dgamma", _fp), ("dbeta", _fp), ("workspace", _fp),
    ]


class LensMeansArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("N", C.c_int32), ("dk", C.c_int32), ("V", C.c_int32), ("L", C.c_int32),
        ("io_dtype", C.c_int32), ("dil", C.c_int32 * MAX_LENS),
        ("q", View4), ("k", View4), ("sqk", _fp), ("lens_w", _fp), ("row", _fp), ("col", _fp), ("d_row", _fp), ("d_col", _fp),
        ("dq", View4), ("dk_", View4), ("dsqk_part", _fp), ("dlens_part", _fp),
    ]


SYMBOLS = {
    "mopk_version": (C.c_int, []),
    "mopk_strerror": (C.c_char_p, [C.c_int]),
    "mopk_edgewise_fused_supported": (C.c_int, [C.POINTER(EdgewiseArgs)]),
    "mopk_edgewise_dominant_kernel": (C.c_char_p, [C.POINTER(EdgewiseArgs), C.c_int]),
    "mopk_edgewise_saved_bytes": (C.c_size_t, [C.POINTER(EdgewiseArgs)]),
    "mopk_edgewise_workspace_bytes": (C.c_size_t, [C.POINTER(EdgewiseArgs)]),
    "mopk_edgewise_lowrank_fwd": (C.c_int, [C.POINTER(EdgewiseArgs), C.c_void_p]),
    "mopk_edgewise_lowrank_bwd": (C.c_int, [C.POINTER(EdgewiseArgs), C.c_void_p]),
    "mopk_edgewise_reduce_parts": (C.c_int, [C.POINTER(EdgewiseArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mopk_edgewise_fwd": (C.c_int, [C.POINTER(EdgewiseArgs), C.c_void_p]),
    "mopk_edgewise_bwd": (C.c_int, [C.POINTER(EdgewiseArgs), C.c_void_p]),
    "mopk_dualpath_saved_bytes": (C.c_size_t, [C.POINTER(DualPathArgs)]),
    "mopk_dualpath_workspace_bytes": (C.c_size_t, [C.POINTER(DualPathArgs)]),
    "mopk_dualpath_fused_supported": (C.c_int, [C.POINTER(DualPathArgs)]),
    "mopk_dualpath_fwd": (C.c_int, [C.POINTER(DualPathArgs), C.c_void_p]),
    "mopk_dualpath_bwd": (C.c_int, [C.POINTER(DualPathArgs), C.c_void_p]),
    "mopk_quartet_saved_bytes": (C.c_size_t, [C.POINTER(QuartetArgs)]),
    "mopk_quartet_workspace_bytes": (C.c_size_t, [C.POINTER(QuartetArgs)]),
    "mopk_quartet_fused_supported": (C.c_int, [C.POINTER(QuartetArgs)]),
    "mopk_quartet_fwd": (C.c_int, [C.POINTER(QuartetArgs), C.c_void_p]),
    "mopk_quartet_bwd": (C.c_int, [C.POINTER(QuartetArgs), C.c_void_p]),
    "mopk_crossview_saved_bytes": (C.c_size_t, [C.POINTER(CrossViewArgs)]),
    "mopk_crossview_workspace_bytes": (C.c_size_t, [C.POINTER(CrossViewArgs)]),
    "mopk_crossview_fwd": (C.c_int, [C.POINTER(CrossViewArgs), C.c_void_p]),
    "mopk_crossview_bwd": (C.c_int, [C.POINTER(CrossViewArgs), C.c_void_p]),
    "mopk_sdpa_saved_bytes": (C.c_size_t, [C.POINTER(SdpaArgs)]),
    "mopk_sdpa_workspace_bytes": (C.c_size_t, [C.POINTER(SdpaArgs)]),
    "mopk_sdpa_fused_supported": (C.c_int, [C.POINTER(SdpaArgs)]),
    "mopk_sdpa_fwd": (C.c_int, [C.POINTER(SdpaArgs), C.c_void_p]),
    "mopk_sdpa_bwd": (C.c_int, [C.POINTER(SdpaArgs), C.c_void_p]),
    "mopk_dropout_keep": (C.c_int, [C.c_uint64, C.c_float, C.c_int64, C.c_int64, C.c_int64]),
    "mopk_layernorm_workspace_bytes": (C.c_size_t, [C.POINTER(LayerNormArgs)]),
    "mopk_layernorm_fwd": (C.c_int, [C.POINTER(LayerNormArgs), C.c_void_p]),
    "mopk_lens_means_supported": (C.c_int, [C.POINTER(LensMeansArgs), C.c_int]),
    "mopk_lens_means_fwd": (C.c_int, [C.POINTER(LensMeansArgs), C.c_void_p]),
    "mopk_lens_means_bwd": (C.c_int, [C.POINTER(LensMeansArgs), C.c_void_p]),
    "mopk_layernorm_bwd": (C.c_int, [C.POINTER(LayerNormArgs), C.c_void_p]),
}

_lib = None


def lib():
    """Load libmopk.so; fail loudly if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m mop_amd.build` "
                "(mop_amd has no CPU or PyTorch fallback for the attention cores)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI drifted
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().mopk_strerror(rc).decode()} (status {rc})")
