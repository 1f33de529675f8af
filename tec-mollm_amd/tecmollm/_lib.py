"""ctypes binding of libtecmollm_hip.so (C ABI: include/tecmollm.h).

The product path has no CPU fallback: if the HIP library is missing, `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TECM_LIB", os.path.join(_HERE, "libtecmollm_hip.so"))   # override for experiments
ABI_VERSION = 17

c_f32p = C.c_void_p


class TecmWin(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("N", C.c_int32), ("Lin", C.c_int32), ("Lout", C.c_int32),
                ("stride_t", C.c_int32), ("taps", C.c_int32), ("Cw", C.c_int32), ("pad", C.c_int32)]


class TecmDrop(C.Structure):
    _fields_ = [("p", C.c_float), ("_pad", C.c_int32), ("seed", C.c_uint64), ("ld", C.c_int64), ("seed_dev", C.c_void_p)]


class TecmGemm(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("A", c_f32p), ("lda", C.c_int64), ("a_layout", C.c_int32), ("io_bf16", C.c_int32),
        ("a_win", TecmWin), ("a_drop", TecmDrop),
        ("B", c_f32p), ("ldb", C.c_int64), ("b_layout", C.c_int32), ("_p1", C.c_int32),
        ("b_win", TecmWin), ("b_drop", TecmDrop),
        ("C", c_f32p), ("ldc", C.c_int64), ("c_win", TecmWin),
        ("alpha", C.c_float), ("act", C.c_int32),
        ("bias", c_f32p),
        ("rowbias", c_f32p), ("rb_ld", C.c_int64), ("rb_div", C.c_int32), ("rb_mod", C.c_int32),
        ("preact", c_f32p), ("ldp", C.c_int64),
        ("dact_src", c_f32p), ("ldd", C.c_int64),
        ("out_drop", TecmDrop),
        ("residual", c_f32p), ("ldr", C.c_int64),
        ("accumulate", C.c_int32), ("split_k", C.c_int32),
        ("workspace", c_f32p),
    ]


class TecmSpatial(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("L", C.c_int32), ("N", C.c_int32), ("Cin", C.c_int32), ("Demb", C.c_int32),
        ("H", C.c_int32), ("graphs_with_edges", C.c_int32),
        ("num_tiles", C.c_int32), ("tile_nodes", C.c_int32), ("win_max", C.c_int32),
        ("x", c_f32p),
        ("tf", c_f32p), ("tf_sb", C.c_int64), ("tf_sl", C.c_int64), ("tf_sn", C.c_int64), ("tf_sf", C.c_int64),
        ("node_tab", c_f32p), ("tod_tab", c_f32p), ("doy_tab", c_f32p), ("year_tab", c_f32p),
        ("season_tab", c_f32p),
        ("year_rows", C.c_int32), ("tile_edges_max", C.c_int32),
        ("Wl", c_f32p), ("bl", c_f32p), ("Wr", c_f32p), ("br", c_f32p),
        ("att", c_f32p), ("bias", c_f32p),
        ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("tile_lo", C.c_void_p), ("tile_hi", C.c_void_p),
        ("alpha_drop", TecmDrop),
        ("out", c_f32p),
        ("err_flag", C.c_void_p),
        ("flags", C.c_int32), ("out_ld", C.c_int32),
    ]


class TecmSpatialGrads(C.Structure):
    _fields_ = [
        ("dout", c_f32p), ("d_node_tab", c_f32p), ("d_tod_tab", c_f32p), ("d_doy_tab", c_f32p),
        ("d_year_tab", c_f32p), ("d_season_tab", c_f32p),
        ("partials", c_f32p), ("partial_ld", C.c_int64),
        ("t_chunk", C.c_int32), ("num_blocks", C.c_int32),
        ("src_ptr", C.c_void_p), ("src_col", C.c_void_p), ("src_ptr_off", C.c_void_p),
    ]


class TecmLnAdd(C.Structure):
    _fields_ = [("dy2", C.c_void_p), ("ld", C.c_int64), ("bf16", C.c_int32), ("_pad", C.c_int32), ("drop", TecmDrop)]


class TecmLnDyMap(C.Structure):
    _fields_ = [("T", C.c_int32), ("N", C.c_int32), ("drop", TecmDrop)]


class TecmAdamW(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("param", c_f32p), ("grad", c_f32p), ("exp_avg", c_f32p), ("exp_avg_sq", c_f32p),
        ("partials", C.c_void_p), ("total_norm_out", c_f32p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("weight_decay", C.c_float), ("max_norm", C.c_float), ("grad_scale", C.c_float),
        ("step", C.c_int32), ("zero_grad", C.c_int32), ("_pad", C.c_int32),
    ]


class TecmMetrics(C.Structure):
    _fields_ = [
        ("pred", c_f32p), ("p_stride_s", C.c_int64), ("p_stride_h", C.c_int64), ("p_stride_i", C.c_int64),
        ("target", c_f32p), ("t_stride_s", C.c_int64), ("t_stride_h", C.c_int64), ("t_stride_i", C.c_int64),
        ("S", C.c_int64), ("H", C.c_int32), ("_pad", C.c_int32), ("I", C.c_int64),
        ("mean", C.c_double), ("scale", C.c_double),
        ("clip_lo", C.c_float), ("clip_hi", C.c_float), ("clip", C.c_int32), ("_pad2", C.c_int32),
        ("stats", C.c_void_p),
    ]


class TecmWindowBatch(C.Structure):
    _fields_ = [
        ("X", c_f32p), ("TF", c_f32p), ("Y", c_f32p), ("starts", C.c_void_p), ("starts_host_check", C.c_void_p),
        ("T", C.c_int64), ("row", C.c_int64),
        ("N", C.c_int32), ("L_in", C.c_int32), ("L_out", C.c_int32), ("F_t", C.c_int32), ("B", C.c_int32),
        ("_pad", C.c_int32),
        ("x_out", c_f32p), ("tf_out", c_f32p), ("y_out", c_f32p),
    ]

class TecmConvDx(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("wpack", C.c_void_p), ("dinp", c_f32p),
                ("B", C.c_int32), ("Lc", C.c_int32), ("N", C.c_int32), ("Cout", C.c_int32), ("ld_in", C.c_int32),
                ("_pad", C.c_int32)]


class TecmConvDw(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("dy", C.c_void_p), ("workspace", c_f32p), ("dw3", c_f32p), ("dw5", c_f32p),
                ("dw7", c_f32p),
                ("B", C.c_int32), ("Lc", C.c_int32), ("N", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("ld_in", C.c_int32), ("num_blocks", C.c_int32), ("_pad", C.c_int32)]


class TecmConvFwd(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("wpack", C.c_void_p), ("bias", c_f32p), ("y", c_f32p),
                ("B", C.c_int32), ("Lc", C.c_int32), ("N", C.c_int32), ("Cout", C.c_int32), ("ld_in", C.c_int32),
                ("y_bf16", C.c_int32), ("stats", c_f32p), ("eps", C.c_float), ("_pad", C.c_int32)]


TECM_NORM_BLOCKS = 512
TECM_METRIC_STATS = 8


EXPORTS = {
    "tecm_abi_version": (C.c_int, []),
    "tecm_last_error": (C.c_char_p, []),
    "tecm_gemm_f32": (C.c_int, [C.POINTER(TecmGemm), C.c_void_p]),
    "tecm_gemm_bf16": (C.c_int, [C.POINTER(TecmGemm), C.c_void_p]),
    "tecm_gemm_bf16x3": (C.c_int, [C.POINTER(TecmGemm), C.c_void_p]),
    "tecm_gemm_bf16x6": (C.c_int, [C.POINTER(TecmGemm), C.c_void_p]),
    "tecm_spatial_fwd": (C.c_int, [C.POINTER(TecmSpatial), C.c_void_p]),
    "tecm_spatial_fwd2_ws_floats": (C.c_int64, [C.POINTER(TecmSpatial)]),
    "tecm_spatial_fwd2": (C.c_int, [C.POINTER(TecmSpatial), c_f32p, C.c_void_p]),
    "tecm_spatial_bwd": (C.c_int, [C.POINTER(TecmSpatial), C.POINTER(TecmSpatialGrads), C.c_void_p]),
    "tecm_spatial_bwd_blocks": (C.c_int, [C.POINTER(TecmSpatial)]),
    "tecm_spatial_bwd2_blocks": (C.c_int, [C.POINTER(TecmSpatial)]),
    "tecm_spatial_bwd2": (C.c_int, [C.POINTER(TecmSpatial), C.POINTER(TecmSpatialGrads), c_f32p, C.c_void_p]),
    "tecm_groupnorm_gelu_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_gn_y16_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "tecm_groupnorm_gelu_bwd": (C.c_int, [c_f32p, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                          C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, c_f32p, C.c_void_p]),
    "tecm_layernorm_fwd": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_int64, C.POINTER(TecmDrop), C.c_int32, C.c_int32,
                                     c_f32p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "tecm_layernorm_bwd": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                     C.c_int32, C.POINTER(TecmDrop), c_f32p, C.POINTER(C.c_int32), C.c_int64, C.c_int32,
                                     C.POINTER(TecmLnAdd), C.c_int32, C.POINTER(TecmLnDyMap), C.c_void_p]),
    "tecm_attention_fwd": (C.c_int, [c_f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.POINTER(TecmDrop), C.c_void_p]),
    "tecm_cast_bf16": (C.c_int, [c_f32p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "tecm_attention_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.POINTER(TecmDrop), C.c_void_p]),
    "tecm_colsum": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, c_f32p, C.c_int64,
                              C.c_int32, C.c_float, C.POINTER(TecmDrop), c_f32p, C.c_void_p]),
    "tecm_colsum_twin": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, c_f32p, C.c_int64,
                                   C.c_int32, C.c_float, C.POINTER(TecmDrop), c_f32p, C.c_void_p, C.c_int64, C.c_void_p]),
    "tecm_huber_fwd_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_float, c_f32p,
                                     C.c_void_p]),
    "tecm_huber_fwd_bwd_strided": (C.c_int, [c_f32p, C.POINTER(C.c_int64), c_f32p, C.POINTER(C.c_int64), c_f32p, c_f32p,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, c_f32p, C.c_void_p]),
    "tecm_lora_fold": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                 C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_pack_vectors": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, c_f32p, C.c_void_p]),
    "tecm_conv_weight_pack": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_conv_weight_unpack": (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_conv_dx_pack": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_conv_dx_bf16": (C.c_int, [C.POINTER(TecmConvDx), C.c_void_p]),
    "tecm_conv_fwd_pack": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_conv_fwd_bf16": (C.c_int, [C.POINTER(TecmConvFwd), C.c_void_p]),
    "tecm_conv_fwd_pack_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_conv_fwd_f32": (C.c_int, [C.POINTER(TecmConvFwd), C.c_void_p]),
    "tecm_conv_dw_workspace": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "tecm_conv_dw_bf16": (C.c_int, [C.POINTER(TecmConvDw), C.c_void_p]),
    "tecm_conv_dw_f32": (C.c_int, [C.POINTER(TecmConvDw), C.c_void_p]),
    "tecm_conv_dx_pack_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tecm_conv_dx_f32": (C.c_int, [C.POINTER(TecmConvDx), C.c_void_p]),
    "tecm_conv_fwd_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "tecm_gemm_tn_splits": (C.c_int32, [C.c_int64, C.c_int64, C.c_int64]),
    "tecm_p8_rows": (C.c_int, [C.c_int64, C.c_int64]),
    "tecm_conv_dx_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "tecm_transpose_scale": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                       C.c_void_p]),
    "tecm_weight_bf16": (C.c_int, [c_f32p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_void_p]),
    "tecm_dropout_apply": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_int32, C.POINTER(TecmDrop), C.c_void_p]),
    "tecm_adamw_clip_step": (C.c_int, [C.POINTER(TecmAdamW), C.c_void_p]),
    "tecm_checksum_tail": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "tecm_checksum_verify": (C.c_int, [c_f32p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "tecm_seed_advance": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "tecm_metrics_accumulate": (C.c_int, [C.POINTER(TecmMetrics), C.c_void_p]),
    "tecm_window_batch": (C.c_int, [C.POINTER(TecmWindowBatch), C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


class TecmError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP library once.  No fallback: a missing/incompatible library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TecmError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "The TEC-MoLLM MI355X path has no CPU/PyTorch fallback.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(handle, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if handle.tecm_abi_version() != ABI_VERSION:
        raise TecmError(f"ABI mismatch: library {handle.tecm_abi_version()} != binding {ABI_VERSION}")
    _lib = handle
    return handle


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().tecm_last_error().decode("utf-8", "replace")
        raise TecmError(f"{what or 'tecm call'} failed (code {rc}): {msg}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> None:
    if not t.is_cuda:
        raise TecmError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TecmError(f"{name} must be {dtype} (got {t.dtype})")
