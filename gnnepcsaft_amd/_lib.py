"""ctypes binding of ``libgnnepcsaft_hip.so`` (include/gnx.h).  No fallback: if the HIP library is missing or fails to
load, importing the compute path raises — the product never routes through a CPU implementation."""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Dict, Optional

import torch

_LIB_PATH = Path(__file__).resolve().parent / "libgnnepcsaft_hip.so"

ABI_VERSION = 3
GNX_OK, GNX_E_INVALID, GNX_E_HIP, GNX_E_RANGE, GNX_E_WORKSPACE = 0, -1, -2, -3, -4
# gnx_set_option ids (include/gnx.h)
OPT_GEMM_SPLIT, OPT_GEMM_WS, OPT_GEMM_VEC, OPT_WGRAD_VEC, OPT_WGRAD_WGS, OPT_AGG_BWD_RECOMPUTE, OPT_EMBED_BWD_MFMA, \
    OPT_STD_BWD_CENTERED, OPT_GEMM_PIPE, OPT_WGRAD_PIPE, OPT_EDGE_FUSED, OPT_SIDE_CUS, OPT_GEMM_AS, OPT_GEMM_WS_FAST, OPT_GEMM_TILE_ROWS, OPT_GEMM_MID, OPT_SPLIT_AHEAD = range(17)
GEMM_RELU, GEMM_ACCUMULATE, GEMM_B_TRANS, GEMM_SPLIT_ONLY, GEMM_PRESPLIT = 1, 2, 4, 8, 16
POOL_ADD, POOL_MEAN, POOL_MAX = 0, 1, 2
K_NONE, K_PNA_AGG_FWD, K_PNA_AGG_BWD, K_GEMM_WS, K_GEMM_WGRAD, K_GINE_AGG_FWD, K_GINE_AGG_BWD, K_EDGE_COMBINE_FWD, \
    K_EDGE_COMBINE_BWD, K_BN_FWD, K_BN_BWD, K_GEMM_TILED, K_GEMM_SMALL, K_GEMM_WGRAD_BATCHED, K_KEY_SEGMENT_SUM, \
    K_EMBED, K_PNA_EDGE_FWD, K_PNA_EDGE_BWD = range(18)
K_COUNT = 18
KERNEL_GROUPS = {K_PNA_AGG_FWD: "pna_aggregate_fwd", K_PNA_AGG_BWD: "pna_aggregate_bwd", K_GEMM_WS: "gemm_weights_stationary",
                 K_GEMM_WGRAD: "weight_gradient", K_GINE_AGG_FWD: "gine_aggregate_fwd", K_GINE_AGG_BWD: "gine_aggregate_bwd",
                 K_EDGE_COMBINE_FWD: "edge_combine_fwd", K_EDGE_COMBINE_BWD: "edge_combine_bwd", K_BN_FWD: "batchnorm_fwd",
                 K_BN_BWD: "batchnorm_bwd", K_GEMM_TILED: "gemm_tiled", K_GEMM_SMALL: "gemm_small",
                 K_GEMM_WGRAD_BATCHED: "weight_gradient_batched", K_KEY_SEGMENT_SUM: "key_segment_sum", K_EMBED: "embedding",
                 K_PNA_EDGE_FWD: "pna_edge_fused_fwd", K_PNA_EDGE_BWD: "pna_edge_fused_bwd"}


class GnxError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"gnx status {status}: {msg}")
        self.status = status


class GemmSeg(C.Structure):
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int64), ("rowscale", C.c_void_p), ("b", C.c_void_p),
                ("ldb", C.c_int64), ("k", C.c_int32), ("_pad", C.c_int32)]


class WgradProb(C.Structure):
    _fields_ = [("dC", C.c_void_p), ("lddc", C.c_int64), ("A", C.c_void_p), ("lda", C.c_int64),
                ("rowscale", C.c_void_p), ("dW", C.c_void_p), ("lddw", C.c_int64), ("dbias", C.c_void_p),
                ("M", C.c_int64), ("N", C.c_int32), ("K", C.c_int32)]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

PNA_MAX_LAYERS, PNA_MAX_TOWERS = 8, 8


class PnaFwdArgs(C.Structure):
    """gnx_pna_fwd_args of include/gnx.h (same field order)."""
    _fields_ = [("N", _i64), ("E", _i64), ("T", _i32), ("F", _i32), ("pre_layers", _i32), ("post_layers", _i32),
                ("D", _i32), ("merged", _i32),
                ("rowptr", _vp), ("src", _vp), ("dst", _vp), ("code", _vp), ("dperm", _vp), ("tiles", _vp), ("ntiles", _vp),
                ("max_tiles", _i64), ("x", _vp), ("Te", _vp), ("weff", _vp * PNA_MAX_TOWERS), ("Wm", _vp), ("bm", _vp),
                ("params", C.POINTER(_vp)), ("P", _vp), ("Q", _vp), ("A", _vp), ("hs", _vp * PNA_MAX_LAYERS),
                ("zs", _vp * PNA_MAX_LAYERS), ("ws", _vp), ("ws_bytes", _sz), ("out", _vp), ("etile_info", _vp),
                ("etile_w", _i32), ("tile_rows", _i32)]


class PnaBwdArgs(C.Structure):
    """gnx_pna_bwd_args of include/gnx.h (same field order)."""
    _fields_ = [("N", _i64), ("E", _i64), ("T", _i32), ("F", _i32), ("pre_layers", _i32), ("post_layers", _i32),
                ("R", _i32), ("D", _i32), ("avg_deg_log", _f32), ("merged", _i32), ("acc_first", _i32),
                ("use_side_streams", _i32), ("n_h", _i32), ("n_z", _i32),
                ("rowptr", _vp), ("colptr", _vp), ("cpos", _vp), ("code", _vp), ("code_pos", _vp), ("dperm", _vp),
                ("tiles", _vp), ("ntiles", _vp), ("chunks", _vp), ("nchunks", _vp),
                ("max_tiles", _i64), ("max_chunks", _i64),
                ("x", _vp), ("BE", _vp), ("EE", _vp), ("A", _vp),
                ("hs", _vp * PNA_MAX_LAYERS), ("zs", _vp * PNA_MAX_LAYERS), ("weff", _vp * PNA_MAX_TOWERS), ("Wm", _vp),
                ("params", C.POINTER(_vp)), ("grads", C.POINTER(_vp)), ("dout", _vp),
                ("gbuf", _vp * PNA_MAX_LAYERS), ("dA", _vp), ("gebuf", _vp * PNA_MAX_LAYERS), ("dP", _vp), ("dQ", _vp),
                ("dTe", _vp), ("dEE", _vp), ("dWm", _vp), ("dbm", _vp), ("dWeff", _vp), ("ws", _vp), ("ws_bytes", _sz),
                ("acc_buf", _vp), ("dx", _vp), ("defer_small", _i32), ("etile_w", _i32), ("etile_info", _vp)]


class PnaFinishArgs(C.Structure):
    """gnx_pna_finish_args of include/gnx.h (same field order)."""
    _fields_ = [("L", _i32), ("T", _i32), ("F", _i32), ("pre_layers", _i32), ("post_layers", _i32), ("R", _i32), ("D", _i32),
                ("merged", _i32), ("use_side_streams", _i32), ("_pad", _i32),
                ("BE", _vp), ("acc_buf", _vp), ("ones", _vp), ("avg_deg_log", C.POINTER(_f32)),
                ("params", C.POINTER(_vp)), ("grads", C.POINTER(_vp)), ("EE", C.POINTER(_vp)), ("dTe", C.POINTER(_vp)),
                ("dEE", C.POINTER(_vp)), ("dWm", C.POINTER(_vp)), ("dbm", C.POINTER(_vp)), ("dWeff", C.POINTER(_vp))]


class SmallProb(C.Structure):
    """gnx_small_prob of include/gnx.h."""
    _fields_ = [("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("bias", _vp), ("C", _vp), ("ldc", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("flags", _i32)]


SB_A_TRANS, SB_B_TRANS, SB_ACCUMULATE, SB_ATOMIC, SB_RELU = 1, 2, 4, 8, 16

# every symbol include/gnx.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "gnx_create": (_i32, [C.POINTER(_vp), _i32]),
    "gnx_destroy": (_i32, [_vp]),
    "gnx_set_stream": (_i32, [_vp, _vp]),
    "gnx_last_error": (C.c_char_p, []),
    "gnx_abi_version": (_i32, []),
    "gnx_set_option": (_i32, [_vp, _i32, _i32]),
    "gnx_get_option": (_i32, [_vp, _i32, C.POINTER(_i32)]),
    "gnx_prof_begin": (_i32, [_vp, C.c_uint32]),
    "gnx_prof_read": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                             C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gnx_prof_end": (_i32, [_vp]),
    "gnx_pack_csr_workspace_bytes": (_sz, [_i64, _i64]),
    "gnx_pack_csr": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "gnx_feature_code": (_i32, [_vp, _vp, _i64, _i32, C.POINTER(_i32), _vp, _vp, _vp, _sz]),
    "gnx_graph_ptr": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _sz]),
    "gnx_degree_scalers": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp]),
    "gnx_embed_sum_fwd": (_i32, [_vp, _vp, _i64, _i32, C.POINTER(_i32), _vp, _i32, _vp]),
    "gnx_table_scatter_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "gnx_embed_sum_bwd": (_i32, [_vp, _vp, _i64, _i32, C.POINTER(_i32), _i32, _vp, _i32, _vp, _vp, _sz]),
    "gnx_check_range": (_i32, [_vp]),
    "gnx_gemm_workspace_bytes": (_sz, [_vp, _i32, C.POINTER(GemmSeg), C.POINTER(_i64), _i32, _i64, _i32, _vp, _i32,
                                       _i32]),
    "gnx_gemm": (_i32, [_vp, _i32, C.POINTER(GemmSeg), _i64, _i32, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _sz]),
    "gnx_gemm_wgrad": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "gnx_gemm_wgrad_batched": (_i32, [_vp, _i32, C.POINTER(WgradProb)]),
    "gnx_degree_max": (_i32, [_vp, _vp, _i64, C.POINTER(_i32)]),
    "gnx_degree_classes_workspace_bytes": (_sz, [_i64, _i32]),
    "gnx_degree_classes": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _sz]),
    "gnx_class_tiles": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "gnx_gemm_grouped": (_i32, [_vp, _i32, C.POINTER(GemmSeg), C.POINTER(_i64), _i32, _i64, _i32, _vp, _vp, _i64, _vp,
                                _i64, _i32, _vp, _vp, _vp, _i64, _vp, _sz]),
    "gnx_gemm_tile_rows": (_i32, [_vp, _i64, _i32]),
    "gnx_gemm_grouped_rows": (_i32, [_vp, _i32, C.POINTER(GemmSeg), C.POINTER(_i64), _i32, _i64, _i32, _vp, _vp, _i64, _vp,
                                     _i64, _i32, _vp, _vp, _vp, _i64, _vp, _sz, _i32]),
    "gnx_gemm_wgrad_grouped": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp,
                                      _i64]),
    "gnx_pna_weff": (_i32, [_vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "gnx_pna_weff_bwd": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _i64]),
    "gnx_group_by_small_key": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _sz]),
    "gnx_key_segment_sum": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "gnx_edge_combine_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gnx_edge_combine_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz]),
    "gnx_pna_aggregate_fwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "gnx_pna_aggregate_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "gnx_edge_tiles_count": (_i32, [_i64, _i32]),
    "gnx_edge_tiles": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp]),
    "gnx_pna_edge_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, C.POINTER(_vp),
                                C.POINTER(_vp), _vp, _vp, _vp]),
    "gnx_pna_edge_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, C.POINTER(_vp), _vp, _vp,
                                _vp]),
    "gnx_gine_aggregate_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp]),
    "gnx_gine_aggregate_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32,
                                      _vp, _vp]),
    "gnx_gine_dle": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gnx_segment_pool_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gnx_segment_pool_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gnx_batchnorm_workspace_bytes": (_sz, [_i64, _i32]),
    "gnx_batchnorm_fwd": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _i32, _vp, _vp, _vp,
                                 _vp, _sz]),
    "gnx_batchnorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _sz]),
    "gnx_dropout": (_i32, [_vp, _vp, _i64, _f32, C.c_uint64, C.c_uint64, _vp]),
    "gnx_huber_ape": (_i32, [_vp, _vp, _vp, _i64, _f32, _vp, _vp]),
    "gnx_adamw_amsgrad": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64]),
    "gnx_sgd": (_i32, [_vp, _vp, _vp, _i64, _f32]),
    "gnx_pna_weight_only": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, C.POINTER(_vp), _i32, _vp, _vp,
                                   C.POINTER(_vp), _vp, _vp]),
    "gnx_pna_conv_fwd": (_i32, [_vp, C.POINTER(PnaFwdArgs)]),
    "gnx_pna_weight_only_all": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_vp),
                                       _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "gnx_pna_stack_finish": (_i32, [_vp, C.POINTER(PnaFinishArgs)]),
    "gnx_gemm_small_batched": (_i32, [_vp, _i32, C.POINTER(SmallProb)]),
    "gnx_pna_weff_batched": (_i32, [_vp, _i32, C.POINTER(_vp), _i64, _i32, _i32, C.POINTER(_f32), C.POINTER(_vp)]),
    "gnx_pna_weff_bwd_batched": (_i32, [_vp, _i32, C.POINTER(_vp), _i32, _i32, C.POINTER(_f32), C.POINTER(_vp), _i64]),
    "gnx_pna_conv_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "gnx_pna_conv_bwd": (_i32, [_vp, C.POINTER(PnaBwdArgs)]),
    "gnx_fill": (_i32, [_vp, _vp, _i64, _f32]),
    "gnx_scale": (_i32, [_vp, _vp, _i64, _f32]),
    "gnx_axpy": (_i32, [_vp, _vp, _vp, _i64, _f32]),
    "gnx_side_stream": (_i32, [_vp, C.POINTER(_vp)]),
    "gnx_side_begin": (_i32, [_vp]),
    "gnx_side_end": (_i32, [_vp]),
    "gnx_side_join": (_i32, [_vp]),
    "gnx_side_stream_n": (_i32, [_vp, _i32, C.POINTER(_vp)]),
    "gnx_side_begin_n": (_i32, [_vp, _i32]),
    "gnx_side_join_n": (_i32, [_vp, _i32]),
    "gnx_clip_rows": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None
_handles: Dict[int, int] = {}


def lib_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise ImportError(f"{_LIB_PATH} is missing: build it with `python -m gnnepcsaft_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gnx_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI version mismatch: library {lib.gnx_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    return load().gnx_last_error().decode(errors="replace")


def check(status: int) -> None:
    if status != GNX_OK:
        raise GnxError(status, last_error())


_bound_stream: dict = {}
try:
    _raw_stream = torch._C._cuda_getCurrentRawStream  # pylint: disable=protected-access
except AttributeError:  # pragma: no cover - older torch
    def _raw_stream(idx: int) -> int:
        return torch.cuda.current_stream(idx).cuda_stream


def handle(device: torch.device) -> int:
    """Opaque gnx_handle* for a CUDA(HIP) device, bound to torch's current stream on that device."""
    if device.type != "cuda":
        raise GnxError(GNX_E_INVALID, f"gnnepcsaft_amd kernels need a HIP device tensor, got {device}; there is no CPU "
                                      "fallback")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    lib = load()
    h = _handles.get(idx)
    if h is None:
        out = _vp()
        check(lib.gnx_create(C.byref(out), idx))
        h = out.value
        _handles[idx] = h
    # raw stream pointer without building a torch.cuda.Stream object (this runs once per launch); the library call is
    # skipped while the stream has not changed
    stream = _raw_stream(idx)
    if _bound_stream.get(idx) != stream:
        check(lib.gnx_set_stream(h, stream))
        _bound_stream[idx] = stream
    return h
