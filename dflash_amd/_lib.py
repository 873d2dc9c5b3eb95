"""ctypes binding of libdflash_hip.so (include/dflash_hip.h).  No fallback: if the
library is missing or a call is rejected, this raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first, so the kernels share its HIP runtime)

HERE = os.path.dirname(os.path.abspath(__file__))
# DFL_LIB_PATH: another build of the SAME library (a diagnostic -D build, an older revision for a same-box A/B)
LIB_PATH = os.environ.get("DFL_LIB_PATH") or os.path.join(HERE, "lib", "libdflash_hip.so")

_p, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class Rows(C.Structure):
    """dfl_rows of include/dflash_hip.h."""
    _fields_ = [("frag", C.c_void_p), ("rows", C.c_void_p), ("ld", C.c_int64), ("ss", C.c_void_p),
                ("nss", C.c_int32), ("norm_w", C.c_void_p), ("eps", C.c_float), ("valid_word", C.c_int32),
                ("mode", C.c_int32)]


_r = C.POINTER(Rows)


class RowsBatch(C.Structure):
    """dfl_rows_batch of include/dflash_hip.h."""
    _fields_ = [("r0", Rows), ("frag_stride", C.c_int64), ("rows_stride", C.c_int64), ("ss_stride", C.c_int64)]


_rb = C.POINTER(RowsBatch)

# name -> (restype, argtypes); mirrors include/dflash_hip.h one to one
SIGNATURES = {
    "dfl_version": (_i, []),
    "dfl_last_error": (C.c_char_p, []),
    "dfl_pack_weight": (_i, [_p, _p, _i, _i, _p]),
    "dfl_pack_weight_gateup": (_i, [_p, _p, _p, _i, _i, _p]),
    "dfl_set_dyn": (_i, [_p, _i, _i, _i, _i, _p]),
    "dfl_set_dyn2": (_i, [_p, _i, _i, _i, _i, _p]),
    "dfl_pack_rows": (_i, [_p, _i64, _i, _i, _p, _p, _i, _p]),
    "dfl_gemm_f32": (_i, [_p, _r, _r, _i, _i, _i, _i, _p, _p, _p]),
    "dfl_gemm_silu_mul": (_i, [_p, _r, _i, _i, _p, _p, _p]),
    "dfl_gemm_resid": (_i, [_p, _r, _i, _i, _p, _i64, _i, _p, _i64, _p, _p, _p]),
    "dfl_embed_rows": (_i, [_p, _p, _p, _i, _p, _p, _i, _p]),
    "dfl_argmax_ws_bytes": (_i64, []),
    "dfl_gemm_argmax": (_i, [_p, _r, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _p, _p]),
    "dfl_gemm_argmax_timed": (_i, [_p, _r, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p]),
    "dfl_norm_pack": (_i, [_p, _i, _i64, _i, _i, _p, _p, _p, _p, _p, _i64, _p, _f, _p, _i, _p, _i, _p]),
    "dfl_qknorm_rope_append": (_i, [_p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _p, _i, _p,
                                    _i, _i, _p]),
    "dfl_attn_ws_bytes": (_i64, [_i, _i]),
    "dfl_block_attn": (_i, [_p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p, _i, _p, _p]),
    "dfl_attn_fused_ws_bytes": (_i64, [_i, _i, _i]),
    "dfl_attn_fused": (_i, [_p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _f, _i, _p,
                            _i, _p, _i, _p, _p]),
    "dfl_attn_head_ws_bytes": (_i64, [_i, _i, _i]),
    "dfl_attn_head": (_i, [_p, _i64, _i, _i, _i, _p, _i64, _i, _i, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _f, _i,
                           _p, _i, _i, _i, _i, _i, _p, _i, _p, _i64, _p]),
    "dfl_attn_head_oproj": (_i, [_p, _i64, _i, _i, _i, _p, _i64, _i, _i, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _f, _i,
                                 _p, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _i64, _p, _p, _p]),
    "dfl_attn_head_cand": (_i, [_p, _i64, _i, _i, _i, _i, _i64, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _f, _i, _i,
                                _p, _i, _p, _i64, _p, _p, _i64, _i, _p]),
    "dfl_attn_head_cand_t": (_i, [_p, _i64, _i, _i, _i, _i, _i64, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _f, _i, _i,
                                  _p, _i, _p, _i64, _i64, _i, _p, _p, _i64, _i, _p]),
    "dfl_attn_head_batch": (_i, [_p, _i64, _i, _i, _i, _i, _i64, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _i64, _f, _i,
                                 _p, _i, _p, _i, _p, _i64, _p]),
    "dfl_attn_head_batch_f32": (_i, [_p, _i, _i64, _i64, _i, _i, _i, _i, _i64, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _i64, _f,
                                     _i, _p, _i, _p, _i, _p, _i64, _p]),
    "dfl_attn_head_batch_t": (_i, [_p, _i64, _i, _i, _i, _i, _i64, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i, _i64, _f, _i,
                                   _p, _i, _p, _i, _p, _i64, _i64, _i, _p]),
    "dfl_topk_rows": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p]),
    "dfl_candidate_select": (_i, [_p, _i64, _p, _i64, _p, _i, _i, _p, _i64, _p, _p, _i, _p, _p]),
    "dfl_moe_route": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    "dfl_moe_router": (_i, [_p, _i64, _p, _f, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _i, _p, _p]),
    "dfl_gemm_silu_mul_experts": (_i, [_p, _i64, _r, _i, _i, _i, _p, _i64, _p, _p, _p, _p]),
    "dfl_moe_gate_up": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    "dfl_moe_down": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "dfl_argmax": (_i, [_p, _i, _i, _i64, _p, _p]),
    "dfl_accept_commit": (_i, [_p, _p, _i, _p, _i64, _p, _p, _i, _p, _p]),
    "dfl_accept_commit_rearm": (_i, [_p, _p, _i, _p, _i64, _p, _p, _i, _p, _p, _i, _i64, _p]),
    "dfl_accept_commit_rearm_t": (_i, [_p, _p, _i, _p, _i64, _p, _p, _i, _p, _p, _i, _i64, _p, _p]),
    # ---- target prefill
    "dfl_prefill_rows_padded": (_i64, [_i]),
    "dfl_prefill_gemm_rows": (_i, [_p, _p, _i, _i, _i, _p, _i64, _p]),
    "dfl_prefill_gemm_resid": (_i, [_p, _p, _i, _i, _i, _p, _i64, _p, _i64, _p]),
    "dfl_prefill_gemm_silu": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "dfl_prefill_norm_pack": (_i, [_p, _i64, _i, _i, _p, _f, _p, _p]),
    "dfl_prefill_pack_rows": (_i, [_p, _i64, _i, _i, _p, _p]),
    "dfl_prefill_qk_rope": (_i, [_p, _i64, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p, _p, _i, _i, _p, _p, _i, _i, _p]),
    "dfl_prefill_attn": (_i, [_p, _i64, _i, _p, _p, _i, _i, _i, _i, _f, _p, _p]),
    "dfl_prefill_moe_max_tiles": (_i64, [_i, _i, _i]),
    "dfl_prefill_moe_max_items": (_i64, [_i, _i, _i]),
    "dfl_prefill_moe_route": (_i, [_p, _i64, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "dfl_prefill_moe_gather": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "dfl_prefill_moe_gemm_silu": (_i, [_p, _i64, _p, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p]),
    "dfl_prefill_moe_gemm_down": (_i, [_p, _i64, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p]),
    "dfl_prefill_moe_combine": (_i, [_p, _p, _i, _i, _i, _p, _i64, _p, _i64, _p, _p]),
    # ---- ragged batch of requests
    "dfl_batch_tiles": (_i, [_i]),
    "dfl_batch_ksplit": (_i, [_i]),
    "dfl_gemm_batch_ws_bytes": (_i64, [_i, _i]),
    "dfl_gemm_f32_batch": (_i, [_p, _rb, _i, _i, _i, _p, _p, _p]),
    "dfl_gemm_silu_mul_batch": (_i, [_p, _rb, _i, _i, _i, _p, _i64, _p, _p, _p]),
    "dfl_gemm_resid_batch": (_i, [_p, _rb, _i, _i, _i, _p, _i64, _i64, _i, _p, _i64, _i64, _p, _i64, _p, _p, _p]),
    "dfl_gemm_argmax_batch": (_i, [_p, _rb, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i64, _i, _p, _i64, _p]),
    "dfl_embed_rows_batch": (_i, [_p, _p, _i64, _i, _p, _i64, _i, _p, _i64, _p, _i, _p]),
    "dfl_norm_frag_batch": (_i, [_p, _i64, _i64, _i, _p, _i, _i64, _i, _p, _i64, _i64, _p, _f, _p, _i64, _i, _p, _i, _p]),
    "dfl_kv_append_batch": (_i, [_p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i64, _f, _p, _p, _i, _p, _p, _i,
                                 _i64, _i64, _p, _p]),
    "dfl_kv_append_batch_t": (_i, [_p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i64, _f, _p, _p, _i, _p, _p, _i,
                                   _i64, _i64, _p, _i, _p]),
    "dfl_attn_fused_batch_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "dfl_attn_fused_batch": (_i, [_p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p, _p, _i, _p, _p, _i,
                                  _i64, _f, _i, _p, _i, _p, _i, _p, _i64, _p]),
    "dfl_accept_commit_batch": (_i, [_p, _i64, _p, _i64, _i, _p, _i64, _i64, _p, _p, _p, _i, _p, _p, _i64, _p]),
    "dfl_accept_commit_batch_t": (_i, [_p, _i64, _p, _i64, _i, _p, _i64, _i64, _p, _p, _p, _i, _p, _p, _i64, _i, _p, _p, _p]),
}

_lib = None


class DFlashHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DFlashHipError(
                f"{LIB_PATH} not found: build it with `python -m dflash_amd.build` (hipcc, gfx950). "
                "dflash_amd has no CPU or PyTorch fallback for its kernels.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        if handle.dfl_version() != 1:
            raise DFlashHipError(f"ABI version mismatch: library {handle.dfl_version()} != binding 1")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DFlashHipError(f"{what} failed (rc={rc}): {lib().dfl_last_error().decode()}")
