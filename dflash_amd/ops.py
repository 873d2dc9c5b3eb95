"""torch-tensor wrappers over the C-ABI (one function per entry point).

torch owns device memory and the stream; these wrappers only validate and pass
raw pointers.  Everything raises on a non-CUDA tensor — there is no host path.
"""
from __future__ import annotations

from typing import Optional

import torch

from ._lib import Rows, check, lib

BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64
DYN_S, DYN_TAU, DYN_BS, DYN_POS0, DYN_START, DYN_STOP, DYN_CYCLE = range(7)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], dtype=None, name="tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"dflash_amd: {name} must be a GPU tensor (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"dflash_amd: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"dflash_amd: {name} must be contiguous")
    return t.data_ptr()


# ---- one-time packing -----------------------------------------------------
def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] bf16 row-major -> packed weight (flat bf16 tensor of N*K)."""
    n, k = w.shape
    out = torch.empty(n * k, dtype=BF16, device=w.device)
    check(lib().dfl_pack_weight(_p(w, BF16, "w"), _p(out), n, k, _stream()), "dfl_pack_weight")
    return out


def pack_weight_gateup(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    i, k = gate.shape
    assert up.shape == gate.shape
    out = torch.empty(2 * i * k, dtype=BF16, device=gate.device)
    check(lib().dfl_pack_weight_gateup(_p(gate, BF16, "gate"), _p(up, BF16, "up"), _p(out), i, k, _stream()),
          "dfl_pack_weight_gateup")
    return out


# ---- per-cycle ------------------------------------------------------------------
def set_dyn(dyn: torch.Tensor, S: int, tau: int, bs: int, pos0: int) -> None:
    check(lib().dfl_set_dyn(_p(dyn, I32, "dyn"), S, tau, bs, pos0, _stream()), "dfl_set_dyn")


def set_dyn2(dyn: torch.Tensor, S: int, tau: int, bs: int, pos0: int) -> None:
    """dyn int32 [>= 16]: one record per 16-row tile of a block of up to 32 rows."""
    assert dyn.numel() >= 16
    check(lib().dfl_set_dyn2(_p(dyn, I32, "dyn"), S, tau, bs, pos0, _stream()), "dfl_set_dyn2")


def pack_rows(x: torch.Tensor, rows: int, out_frag: torch.Tensor, dyn=None, dyn_word=0) -> None:
    """x: [rows(+), K] bf16 with unit inner stride -> frag16 (out_frag has 16*K elements)."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == BF16 and x.is_cuda
    k = x.shape[1]
    assert out_frag.numel() >= 16 * k
    check(lib().dfl_pack_rows(x.data_ptr(), x.stride(0), rows, k, _p(out_frag, BF16, "out_frag"),
                              _p(dyn, I32, "dyn"), dyn_word, _stream()), "dfl_pack_rows")


# ---- row sources (dfl_rows): where a GEMM's 16-row activation tile comes from ---------
import ctypes as _C


class RowSource:
    """Keeps the ctypes struct and the tensors it points into alive together."""

    def __init__(self, struct: Rows, *keep):
        self.struct, self.keep = struct, keep

    @property
    def ref(self):
        return _C.byref(self.struct)


def rows_frag(frag: torch.Tensor) -> RowSource:
    """frag16 fragments written by a producer kernel."""
    return RowSource(Rows(_p(frag, BF16, "frag"), None, 0, None, 0, None, 0.0, -1, 0), frag)


def _readable16(rows: torch.Tensor) -> torch.Tensor:
    """The kernels load all 16 rows of a tile before they know how many are valid (the lengths
    arrive by a scalar load they do not wait for): the storage behind `rows` must cover 16
    rows.  A view of a 16-row buffer does; a shorter tensor is copied into a padded one."""
    need = (15 * rows.stride(0) + rows.shape[1]) * rows.element_size()
    have = rows.untyped_storage().nbytes() - rows.storage_offset() * rows.element_size()
    if rows.shape[0] >= 16 or have >= need:
        return rows
    pad = torch.zeros(16, rows.shape[1], dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    return pad


def rows_plain(rows: torch.Tensor, valid_word: int = -1) -> RowSource:
    """bf16 rows [<=16, K] (unit inner stride); rows >= dyn[valid_word] count as zero."""
    assert rows.is_cuda and rows.dtype == BF16 and rows.dim() == 2 and rows.stride(1) == 1
    rows = _readable16(rows)
    return RowSource(Rows(None, rows.data_ptr(), rows.stride(0), None, 0, None, 0.0, valid_word, 1), rows)


def rows_normed(rows: torch.Tensor, ss: torch.Tensor, nss: int, norm_w: torch.Tensor, eps: float,
                valid_word: int = -1) -> RowSource:
    """the residual stream + partial sums of squares: the GEMM applies the RMSNorm itself."""
    assert rows.is_cuda and rows.dtype == BF16 and rows.dim() == 2 and rows.stride(1) == 1
    assert ss.numel() >= nss * 16
    rows = _readable16(rows)
    return RowSource(Rows(None, rows.data_ptr(), rows.stride(0), _p(ss, F32, "ss"), nss, _p(norm_w, BF16, "norm_w"),
                          eps, valid_word, 2), rows, ss, norm_w)


def _src(x) -> RowSource:
    return x if isinstance(x, RowSource) else rows_frag(x)


def gemm_f32(wp, x0, x1, mt: int, N: int, K: int, ksplit: int, out: torch.Tensor, dyn=None) -> None:
    """x0/x1: RowSource, or a frag16 tensor."""
    assert out.numel() >= ksplit * mt * 16 * N
    s0, s1 = _src(x0), (_src(x1) if x1 is not None else None)
    check(lib().dfl_gemm_f32(_p(wp, BF16, "wp"), s0.ref, s1.ref if s1 is not None else None, mt, N, K, ksplit,
                             _p(out, F32, "out"), _p(dyn, I32, "dyn"), _stream()), "dfl_gemm_f32")


def min_ksplit(K: int, mt: int) -> int:
    per = 16 * (8 if mt == 1 else 4) * 32
    return (K + per - 1) // per


def pick_ksplit(N: int, K: int, mt: int) -> int:
    """K split for dfl_gemm_f32 that fills the 256 CUs evenly: work units = column tiles x
    K chunks, dealt to <= 256 workgroups of equal load, preferring >= 2 tiles per workgroup
    (the next tile's weights are prefetched under the current one).  Measured on MI355X
    (scripts/bench_gemm.py): every shape runs ~3.6 us + bytes / 6.5 TB/s for any split
    near this choice, so the picker only has to avoid the unbalanced ones."""
    ntiles, ks_steps = N // 16, K // 32
    best, best_score = None, None
    for ks in range(min_ksplit(K, mt), 17):
        nfr = -(-ks_steps // (16 * ks))
        if nfr < 2 and ks > min_ksplit(K, mt):
            break
        gx_max = max(1, 256 // ks)
        per_wg = -(-ntiles // gx_max)
        fill = (ntiles * ks) / (256.0 * per_wg)           # busy fraction of the chip over the launch
        score = (round(min(fill, 1.0), 3), min(per_wg, 2), -ks)
        if best_score is None or score > best_score:
            best, best_score = ks, score
    return best


def gemm_silu_mul(wp_gu, x, I: int, K: int, act_frag: torch.Tensor, dyn=None) -> None:
    assert act_frag.numel() >= 16 * I
    check(lib().dfl_gemm_silu_mul(_p(wp_gu, BF16, "wp_gu"), _src(x).ref, I, K, _p(act_frag, BF16, "act"),
                                  _p(dyn, I32, "dyn"), _stream()), "dfl_gemm_silu_mul")


def argmax_ws(device) -> torch.Tensor:
    return torch.empty(lib().dfl_argmax_ws_bytes(), dtype=torch.uint8, device=device)


def gemm_argmax(wp, x, V: int, K: int, row0: int, nrows: int, ws, out_ids: torch.Tensor, out_off: int = 0,
                dyn=None, nrows_dyn_word: int = -1, logits: Optional[torch.Tensor] = None,
                margins: Optional[torch.Tensor] = None, events=None) -> None:
    """margins: optional fp32 tensor indexed like out_ids: top-1 minus top-2 logit per row.
    events: (start, end) torch.cuda.Event pair (enable_timing, already recorded once so that their handles exist),
    recorded right around the GEMM launch itself (dfl_gemm_argmax_timed)."""
    if logits is not None:
        assert logits.numel() >= 16 * V
    if margins is not None:
        assert margins.numel() >= out_off + nrows
    args = (_p(wp, BF16, "wp"), _src(x).ref, V, K, row0, nrows, _p(dyn, I32, "dyn"), nrows_dyn_word, _p(ws),
            _p(out_ids, I64, "out_ids"), out_off, _p(logits, BF16, "logits"), _p(margins, F32, "margins"))
    if events is not None:
        check(lib().dfl_gemm_argmax_timed(*args, events[0].cuda_event, events[1].cuda_event, _stream()),
              "dfl_gemm_argmax_timed")
        return
    check(lib().dfl_gemm_argmax(*args, _stream()), "dfl_gemm_argmax")


def gemm_resid(wp, x, N: int, K: int, h_io: torch.Tensor, *, add_residual: bool, ss_out=None, tap=None,
               dyn=None) -> None:
    """h_io [16, >=N] bf16 (unit inner stride) updated in place; tap: optional [16, *] view
    receiving the same rows; ss_out: fp32 [N/16 * 16] partial sums of squares."""
    assert h_io.is_cuda and h_io.dtype == BF16 and h_io.stride(1) == 1
    tp, ldt = None, 0
    if tap is not None:
        assert tap.is_cuda and tap.dtype == BF16 and tap.stride(1) == 1
        tp, ldt = tap.data_ptr(), tap.stride(0)
    if ss_out is not None:
        assert ss_out.numel() >= N
    check(lib().dfl_gemm_resid(_p(wp, BF16, "wp"), _src(x).ref, N, K, h_io.data_ptr(), h_io.stride(0),
                               int(add_residual), tp, ldt, _p(ss_out, F32, "ss_out"), _p(dyn, I32, "dyn"), _stream()),
          "dfl_gemm_resid")


def embed_rows(embed, ids, h_out, H: int, ss_out, dyn=None, dyn_word: int = 0) -> None:
    check(lib().dfl_embed_rows(_p(embed, BF16, "embed"), _p(ids, I64, "ids"), _p(h_out, BF16, "h_out"), H,
                               _p(ss_out, F32, "ss_out"), _p(dyn, I32, "dyn"), dyn_word, _stream()), "dfl_embed_rows")


def norm_pack(*, norm_w, frag, H: int, eps: float, part=None, nsplit=0, part_split=0, ldp=0, row_off=0,
              resid_in=None, embed=None, ids=None, h_out=None, h_out2=None, ld2=0, dyn=None, dyn_word=0) -> None:
    # h_out2 may be a column slice of a wider row-major buffer: only its base pointer and ld2 are used
    p2 = None
    if h_out2 is not None:
        assert h_out2.is_cuda and h_out2.dtype == BF16 and h_out2.stride(-1) == 1
        p2 = h_out2.data_ptr()
    check(lib().dfl_norm_pack(_p(part, F32, "part"), nsplit, part_split, ldp, row_off, _p(resid_in, BF16, "resid_in"),
                              _p(embed, BF16, "embed"), _p(ids, I64, "ids"), _p(h_out, BF16, "h_out"), p2, ld2,
                              _p(norm_w, BF16, "norm_w"), eps, _p(frag, BF16, "frag"), H, _p(dyn, I32, "dyn"),
                              dyn_word, _stream()), "dfl_norm_pack")


def qknorm_rope_append(*, qkv, nsplit, split_stride, ld, q_col, k_col, v_col, ctx_row0, blk_row0, n_q, n_kv,
                       q_norm_w, k_norm_w, eps, cos_tab, sin_tab, q_out, kcache, vcache, dyn,
                       ctx_rows_override=-1, row_base=0) -> None:
    assert kcache.shape == vcache.shape and kcache.dim() == 3 and kcache.shape[2] == 128
    check(lib().dfl_qknorm_rope_append(
        _p(qkv, F32, "qkv"), nsplit, split_stride, ld, q_col, k_col, v_col, ctx_row0, blk_row0, n_q, n_kv,
        _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
        _p(sin_tab, BF16, "sin"), cos_tab.shape[0], _p(q_out, BF16, "q_out"), _p(kcache, BF16, "kcache"),
        _p(vcache, BF16, "vcache"), kcache.shape[1], _p(dyn, I32, "dyn"), ctx_rows_override, row_base, _stream()),
        "dfl_qknorm_rope_append")


def attn_ws(n_q: int, max_splits: int, device) -> torch.Tensor:
    return torch.empty(lib().dfl_attn_ws_bytes(n_q, max_splits), dtype=torch.uint8, device=device)


def block_attn(*, q, kcache, vcache, n_q, n_kv, scale, dyn, kv_len_max, ws, max_splits, out_frag,
               causal: bool = False) -> None:
    check(lib().dfl_block_attn(_p(q, BF16, "q"), _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"),
                               kcache.shape[1], n_q, n_kv, scale, int(causal), _p(dyn, I32, "dyn"), kv_len_max,
                               _p(ws), max_splits, _p(out_frag, BF16, "out_frag"), _stream()), "dfl_block_attn")


def attn_fused_ws(n_q: int, n_kv: int, max_splits: int, device) -> torch.Tensor:
    """Zeroed workspace (split partials + arrival tickets) for attn_fused."""
    return torch.zeros(lib().dfl_attn_fused_ws_bytes(n_q, n_kv, max_splits), dtype=torch.uint8, device=device)


def attn_fused(*, qkv, nsplit, split_stride, ld, q_col, k_col, v_col, ctx_row0, blk_row0, n_q, n_kv, q_norm_w,
               k_norm_w, eps, cos_tab, sin_tab, kcache, vcache, scale, dyn, kv_len_max, ws, max_splits, out_frag,
               causal: bool = False) -> None:
    assert kcache.shape == vcache.shape and kcache.dim() == 3 and kcache.shape[2] == 128
    check(lib().dfl_attn_fused(
        _p(qkv, F32, "qkv"), nsplit, split_stride, ld, q_col, k_col, v_col, ctx_row0, blk_row0, n_q, n_kv,
        _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
        _p(sin_tab, BF16, "sin"), cos_tab.shape[0], _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"),
        kcache.shape[1], scale, int(causal), _p(dyn, I32, "dyn"), kv_len_max, _p(ws), max_splits,
        _p(out_frag, BF16, "out_frag"), _stream()), "dfl_attn_fused")


def attn_head_ws(n_q: int, max_splits: int, q_tiles: int, device) -> torch.Tensor:
    """Zeroed workspace (split partials + arrival tickets) for attn_head."""
    return torch.zeros(lib().dfl_attn_head_ws_bytes(n_q, max_splits, q_tiles), dtype=torch.uint8, device=device)


def attn_head(*, xq: torch.Tensor, q_col: int, k_col: int, v_col: int, n_q: int, n_kv: int, q_norm_w, k_norm_w, eps,
              cos_tab, sin_tab, kcache, vcache, scale: float, causal: bool, S: int, tau: int, bs: int, pos0: int,
              ws, max_splits: int, out_frag: torch.Tensor, xc: Optional[torch.Tensor] = None, ck_col: int = 0,
              cv_col: int = 0, dyn: Optional[torch.Tensor] = None, q_tiles: int = 1, out_tile_stride: int = 0) -> None:
    """xq [>= bs, ldq] bf16 rows (unit inner stride) holding q | k | v of the block rows; xc [>= tau, ldc] the
    context rows' k | v (draft).  dyn given: lengths from the device record, S = bound for the split count."""
    assert xq.is_cuda and xq.dtype == BF16 and xq.dim() == 2 and xq.stride(1) == 1 and xq.shape[0] >= min(bs, 16 * q_tiles)
    assert kcache.shape == vcache.shape and kcache.dim() == 3 and kcache.shape[2] == 128
    xcp, ldc = None, 0
    if xc is not None:
        assert xc.is_cuda and xc.dtype == BF16 and xc.dim() == 2 and xc.stride(1) == 1 and xc.shape[0] >= tau
        xcp, ldc = xc.data_ptr(), xc.stride(0)
    check(lib().dfl_attn_head(
        xq.data_ptr(), xq.stride(0), q_col, k_col, v_col, xcp, ldc, ck_col, cv_col, n_q, n_kv,
        _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
        _p(sin_tab, BF16, "sin"), cos_tab.shape[0], _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"),
        kcache.shape[1], scale, int(causal), _p(dyn, I32, "dyn"), S, tau, bs, pos0, q_tiles, _p(ws), max_splits,
        _p(out_frag, BF16, "out_frag"), out_tile_stride, _stream()), "dfl_attn_head")


ATTN_OPROJ_SYNC_WORDS = 1056   # DFL_ATTN_OPROJ_SYNC_WORDS: 32 counter replicas on their own lines + bookkeeping
ATTN_OPROJ_FAIL_WORD = 1025


def attn_head_oproj(*, xq: torch.Tensor, q_col: int, k_col: int, v_col: int, n_q: int, n_kv: int, q_norm_w, k_norm_w, eps,
                    cos_tab, sin_tab, kcache, vcache, scale: float, causal: bool, S: int, tau: int, bs: int, pos0: int,
                    ws, max_splits: int, attn_frag: torch.Tensor, wo, H: int, h_io: torch.Tensor, ss_out, sync,
                    xc: Optional[torch.Tensor] = None, ck_col: int = 0, cv_col: int = 0,
                    dyn: Optional[torch.Tensor] = None) -> None:
    """attn_head (one query tile) + the o_proj / residual GEMM behind it in one launch; sync: int32[ATTN_OPROJ_SYNC_WORDS],
    zeroed once (sync[ATTN_OPROJ_FAIL_WORD] != 0 afterwards: the launch gave up waiting, h_io is invalid)."""
    assert xq.is_cuda and xq.dtype == BF16 and xq.dim() == 2 and xq.stride(1) == 1 and xq.shape[0] >= min(bs, 16)
    assert kcache.shape == vcache.shape and kcache.dim() == 3 and kcache.shape[2] == 128
    assert h_io.is_cuda and h_io.dtype == BF16 and h_io.stride(1) == 1 and h_io.shape[0] >= 16
    assert sync.dtype == I32 and sync.numel() >= ATTN_OPROJ_SYNC_WORDS and attn_frag.numel() >= 16 * n_q * 128
    assert ss_out is None or ss_out.numel() >= H
    xcp, ldc = None, 0
    if xc is not None:
        assert xc.is_cuda and xc.dtype == BF16 and xc.dim() == 2 and xc.stride(1) == 1 and xc.shape[0] >= tau
        xcp, ldc = xc.data_ptr(), xc.stride(0)
    check(lib().dfl_attn_head_oproj(
        xq.data_ptr(), xq.stride(0), q_col, k_col, v_col, xcp, ldc, ck_col, cv_col, n_q, n_kv,
        _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
        _p(sin_tab, BF16, "sin"), cos_tab.shape[0], _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"),
        kcache.shape[1], scale, int(causal), _p(dyn, I32, "dyn"), S, tau, bs, pos0, _p(ws), max_splits,
        _p(attn_frag, BF16, "attn_frag"), _p(wo, BF16, "wo"), H, h_io.data_ptr(), h_io.stride(0), _p(ss_out, F32, "ss_out"),
        _p(sync, I32, "sync"), _stream()), "dfl_attn_head_oproj")


def attn_head_cand(*, xq: torch.Tensor, q_col: int, k_col: int, v_col: int, n_q: int, n_kv: int, q_norm_w, k_norm_w, eps,
                   cos_tab, sin_tab, kcache, vcache, scale: float, S: int, bs: int, ws, max_splits: int,
                   out_frag: torch.Tensor, k_out: torch.Tensor, v_out: torch.Tensor, q_tiles: int = 1) -> None:
    """xq [tiles, 16, ldq] bf16 candidate block rows; out_frag [tiles, 16*n_q*128]; k_out / v_out [C, n_kv, rows, 128]:
    the candidates' new K/V rows (rows 0..bs-1), NOT written to the cache.  q_tiles = 2 (bs 17..32): candidate c owns tiles
    2 c, 2 c + 1 of xq / out_frag."""
    assert xq.is_cuda and xq.dtype == BF16 and xq.dim() == 3 and xq.stride(2) == 1 and xq.shape[1] >= 16
    assert out_frag.dim() == 2 and out_frag.is_contiguous() and out_frag.shape[0] >= xq.shape[0]
    assert k_out.shape == v_out.shape and k_out.dim() == 4 and k_out.shape[3] == 128 and k_out.is_contiguous()
    assert v_out.is_contiguous() and k_out.shape[0] * q_tiles >= xq.shape[0] and k_out.shape[1] == n_kv
    assert kcache.shape == vcache.shape and kcache.dim() == 3 and kcache.shape[2] == 128
    if q_tiles != 1:
        assert xq.shape[0] % q_tiles == 0 and k_out.shape[2] >= bs
        check(lib().dfl_attn_head_cand_t(
            xq.data_ptr(), xq.stride(1), q_col, k_col, v_col, xq.shape[0] // q_tiles, q_tiles * xq.stride(0), n_q, n_kv,
            _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
            _p(sin_tab, BF16, "sin"), cos_tab.shape[0], _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"),
            kcache.shape[1], scale, S, bs, _p(ws), max_splits, _p(out_frag, BF16, "out_frag"), q_tiles * out_frag.stride(0),
            out_frag.stride(0), q_tiles, _p(k_out, BF16, "k_out"), _p(v_out, BF16, "v_out"), k_out.stride(0), k_out.shape[2],
            _stream()), "dfl_attn_head_cand_t")
        return
    check(lib().dfl_attn_head_cand(
        xq.data_ptr(), xq.stride(1), q_col, k_col, v_col, xq.shape[0], xq.stride(0), n_q, n_kv,
        _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
        _p(sin_tab, BF16, "sin"), cos_tab.shape[0], _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"),
        kcache.shape[1], scale, S, bs, _p(ws), max_splits, _p(out_frag, BF16, "out_frag"), out_frag.stride(0),
        _p(k_out, BF16, "k_out"), _p(v_out, BF16, "v_out"), k_out.stride(0), k_out.shape[2], _stream()),
        "dfl_attn_head_cand")


def attn_head_batch_ws(R: int, n_q: int, max_splits: int, device, q_tiles: int = 1) -> torch.Tensor:
    return torch.zeros(R * lib().dfl_attn_head_ws_bytes(n_q, max_splits, q_tiles), dtype=torch.uint8, device=device)


def attn_head_batch(*, xq: torch.Tensor, q_col: int, k_col: int, v_col: int, R: int, n_q: int, n_kv: int, q_norm_w,
                    k_norm_w, eps, cos_tab, sin_tab, kcache, vcache, layer: int, scale: float, causal: bool, dyn,
                    kv_len_max: int, ws, max_splits: int, out_frag: torch.Tensor, q_tiles: int = 1) -> None:
    """xq [MT, 16, ldq] bf16; kcache/vcache [requests, L, n_kv, rows, 128] (layer `layer` is used); out_frag
    [MT, 16*n_q*128].  q_tiles = 2: request r owns tiles 2 r, 2 r + 1 of xq / out_frag (blocks of 17..32 rows), dyn holds
    one record per REQUEST."""
    assert xq.is_cuda and xq.dtype == BF16 and xq.dim() == 3 and xq.stride(2) == 1
    assert kcache.shape == vcache.shape and kcache.dim() == 5 and kcache.shape[4] == 128 and kcache.is_contiguous()
    assert out_frag.dim() == 2 and out_frag.is_contiguous()
    kc, vc = kcache[0, layer], vcache[0, layer]
    if q_tiles != 1:
        check(lib().dfl_attn_head_batch_t(
            xq.data_ptr(), xq.stride(1), q_col, k_col, v_col, R, q_tiles * xq.stride(0), n_q, n_kv,
            _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
            _p(sin_tab, BF16, "sin"), cos_tab.shape[0], kc.data_ptr(), vc.data_ptr(), kcache.shape[3], kcache.stride(0), scale,
            int(causal), _p(dyn, I32, "dyn"), kv_len_max, _p(ws), max_splits, _p(out_frag, BF16, "out_frag"),
            q_tiles * out_frag.stride(0), out_frag.stride(0), q_tiles, _stream()), "dfl_attn_head_batch_t")
        return
    check(lib().dfl_attn_head_batch(
        xq.data_ptr(), xq.stride(1), q_col, k_col, v_col, R, xq.stride(0), n_q, n_kv, _p(q_norm_w, BF16, "q_norm_w"),
        _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"), _p(sin_tab, BF16, "sin"), cos_tab.shape[0],
        kc.data_ptr(), vc.data_ptr(), kcache.shape[3], kcache.stride(0), scale, int(causal), _p(dyn, I32, "dyn"),
        kv_len_max, _p(ws), max_splits, _p(out_frag, BF16, "out_frag"), out_frag.stride(0), _stream()),
        "dfl_attn_head_batch")


def attn_head_batch_f32(*, qkv_parts: torch.Tensor, nparts: int, MT: int, ld: int, q_col: int, k_col: int, v_col: int, R: int,
                        n_q: int, n_kv: int, q_norm_w, k_norm_w, eps, cos_tab, sin_tab, kcache, vcache, layer: int, scale: float,
                        causal: bool, dyn, kv_len_max: int, ws, max_splits: int, out_frag: torch.Tensor) -> None:
    """attn_head_batch on the fp32 K-part sums gemm_f32_batch(N = ld) left in qkv_parts ([nparts][MT * 16][ld] floats):
    the parts meet in the attention launch's row loads (no slab / ticket / combine phase in the GEMM)."""
    assert qkv_parts.is_cuda and qkv_parts.dtype == F32 and qkv_parts.is_contiguous() and qkv_parts.numel() >= nparts * MT * 16 * ld
    assert kcache.shape == vcache.shape and kcache.dim() == 5 and kcache.shape[4] == 128 and kcache.is_contiguous()
    assert out_frag.dim() == 2 and out_frag.is_contiguous()
    kc, vc = kcache[0, layer], vcache[0, layer]
    check(lib().dfl_attn_head_batch_f32(
        qkv_parts.data_ptr(), nparts, MT * 16 * ld, ld, q_col, k_col, v_col, R, 16 * ld, n_q, n_kv,
        _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"), _p(sin_tab, BF16, "sin"),
        cos_tab.shape[0], kc.data_ptr(), vc.data_ptr(), kcache.shape[3], kcache.stride(0), scale, int(causal),
        _p(dyn, I32, "dyn"), kv_len_max, _p(ws), max_splits, _p(out_frag, BF16, "out_frag"), out_frag.stride(0), _stream()),
        "dfl_attn_head_batch_f32")


def topk_rows(logits: torch.Tensor, k: int):
    """logits bf16 [rows, V] (unit inner stride) -> (values fp32 [rows, 8], indices int32 [rows, 8], lse fp32 [rows]);
    columns >= k are unspecified.  Order: value descending, index ascending."""
    assert logits.is_cuda and logits.dtype == BF16 and logits.dim() == 2 and logits.stride(1) == 1
    rows, V = logits.shape
    val = torch.empty(rows, 8, dtype=F32, device=logits.device)
    idx = torch.empty(rows, 8, dtype=I32, device=logits.device)
    lse = torch.empty(rows, dtype=F32, device=logits.device)
    check(lib().dfl_topk_rows(logits.data_ptr(), logits.stride(0), rows, V, k, _p(val), _p(idx), _p(lse), _stream()),
          "dfl_topk_rows")
    return val, idx, lse


def candidate_select(blocks: torch.Tensor, posterior: torch.Tensor, scores: torch.Tensor, bs: int, output_ids, dyn,
                     stop_ids, result: torch.Tensor) -> None:
    """blocks / posterior int64 [C, >= bs]; scores fp32 [C]; result int32 [12]."""
    assert blocks.dim() == 2 and posterior.dim() == 2 and blocks.stride(1) == 1 and posterior.stride(1) == 1
    assert result.numel() >= 12 and scores.numel() >= blocks.shape[0]
    n_stop = 0 if stop_ids is None else stop_ids.numel()
    check(lib().dfl_candidate_select(
        _p(blocks[0], I64, "blocks"), blocks.stride(0), _p(posterior[0], I64, "posterior"), posterior.stride(0),
        _p(scores, F32, "scores"), blocks.shape[0], bs, _p(output_ids, I64, "output_ids"), output_ids.numel(),
        _p(dyn, I32, "dyn"), _p(stop_ids, I64, "stop_ids") if n_stop else None, n_stop, _p(result, I32, "result"),
        _stream()), "dfl_candidate_select")


# ---- sparse-MoE MLP of a target verify (include/dflash_hip.h, moe section) ----------------------------------------
def moe_route(logits: torch.Tensor, E: int, top_k: int, norm_topk: bool, wt: torch.Tensor, active: torch.Tensor,
              lst: torch.Tensor, n_active: torch.Tensor, dyn=None, dyn_word: int = 0) -> None:
    """logits bf16 [16, >= E]; wt bf16 [16, E]; active / lst int32 [E]; n_active int32 [1]."""
    assert logits.dtype == BF16 and logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[0] >= 16
    assert wt.dtype == BF16 and wt.is_contiguous() and wt.numel() >= 16 * E
    check(lib().dfl_moe_route(logits.data_ptr(), logits.stride(0), E, top_k, int(norm_topk), _p(wt), _p(active, I32, "active"),
                              _p(lst, I32, "list"), _p(n_active, I32, "n_active"), _p(dyn, I32, "dyn"), dyn_word,
                              _stream()), "dfl_moe_route")


def moe_router(*, h: Optional[torch.Tensor], norm_w: Optional[torch.Tensor], eps: float, xn: torch.Tensor, wp_router: torch.Tensor,
               K: int, E: int, top_k: int, norm_topk: bool, rlog: torch.Tensor, wt: torch.Tensor, active: torch.Tensor,
               lst: torch.Tensor, n_active: torch.Tensor, ticket: torch.Tensor, dyn=None, dyn_word: int = 0) -> None:
    """RMSNorm (h [16, >= K] rows -> xn frag16; h None: xn is the input) + gate Linear + moe_route in one launch.
    rlog bf16 [16, ld]; ticket int32 [1], zero."""
    assert rlog.dtype == BF16 and rlog.dim() == 2 and rlog.stride(1) == 1 and rlog.shape[0] >= 16
    assert wt.dtype == BF16 and wt.is_contiguous() and wt.numel() >= 16 * E and xn.numel() >= 16 * K
    hp, ldh = None, 0
    if h is not None:
        assert h.is_cuda and h.dtype == BF16 and h.dim() == 2 and h.stride(1) == 1 and h.shape[0] >= 16 and h.shape[1] >= K
        hp, ldh = h.data_ptr(), h.stride(0)
    check(lib().dfl_moe_router(hp, ldh, _p(norm_w, BF16, "norm_w"), eps, _p(xn, BF16, "xn"), _p(wp_router, BF16, "wp_router"), K, E,
                               top_k, int(norm_topk), rlog.data_ptr(), rlog.stride(0), _p(wt), _p(active, I32, "active"),
                               _p(lst, I32, "list"), _p(n_active, I32, "n_active"), _p(dyn, I32, "dyn"), dyn_word,
                               _p(ticket, I32, "ticket"), _stream()), "dfl_moe_router")


def gemm_silu_mul_experts(wp_gu: torch.Tensor, x, E: int, I: int, K: int, act: torch.Tensor, lst: torch.Tensor,
                          n_active: torch.Tensor, dyn=None) -> None:
    """wp_gu bf16 [E, 2*I*K] packed per expert; act bf16 [E, 16*I] (frag16 per expert); lst / n_active: the active
    experts (dfl_moe_route).  Only their outputs are written."""
    assert wp_gu.dim() == 2 and wp_gu.is_contiguous() and act.dim() == 2 and act.is_contiguous()
    check(lib().dfl_gemm_silu_mul_experts(_p(wp_gu, BF16, "wp_gu"), wp_gu.stride(0), _src(x).ref, E, I, K,
                                          _p(act, BF16, "act"), act.stride(0), _p(lst, I32, "list"),
                                          _p(n_active, I32, "n_active"), _p(dyn, I32, "dyn"), _stream()),
          "dfl_gemm_silu_mul_experts")


def moe_gate_up(wp_gu: torch.Tensor, x_frag: torch.Tensor, E: int, I: int, K: int, act: torch.Tensor, lst: torch.Tensor,
                n_active: torch.Tensor, dyn=None, valid_word: int = -1) -> None:
    """gemm_silu_mul_experts for K <= 2048 from frag16 rows: one LDS meeting per (gate, up) tile pair."""
    assert wp_gu.dim() == 2 and wp_gu.is_contiguous() and wp_gu.shape[1] == 2 * I * K
    assert act.dim() == 2 and act.is_contiguous() and act.shape[1] == 16 * I and x_frag.numel() >= 16 * K
    check(lib().dfl_moe_gate_up(_p(wp_gu, BF16, "wp_gu"), _p(x_frag, BF16, "x_frag"), E, I, K, _p(act, BF16, "act"),
                                _p(lst, I32, "list"), _p(n_active, I32, "n_active"), _p(dyn, I32, "dyn"), valid_word,
                                _stream()), "dfl_moe_gate_up")


def moe_down(wp_down: torch.Tensor, act: torch.Tensor, wt: torch.Tensor, lst: torch.Tensor, n_active: torch.Tensor, E: int,
             N: int, I: int, nsplit: int, out: torch.Tensor) -> None:
    """wp_down bf16 [E, N*I] packed per expert; out fp32 [nsplit, 16, N]."""
    assert wp_down.dim() == 2 and wp_down.is_contiguous() and act.dim() == 2 and act.is_contiguous()
    assert out.dtype == F32 and out.is_contiguous() and out.numel() >= nsplit * 16 * N
    check(lib().dfl_moe_down(_p(wp_down, BF16, "wp_down"), wp_down.stride(0), _p(act, BF16, "act"), act.stride(0),
                             _p(wt, BF16, "wt"), _p(lst, I32, "list"), _p(n_active, I32, "n_active"), E, N, I, nsplit,
                             _p(out, F32, "out"), _stream()), "dfl_moe_down")


def argmax(logits: torch.Tensor) -> torch.Tensor:
    """First-max-index argmax over the last axis, int64 (model/utils.py:28-29)."""
    if logits.dtype not in (BF16, F32):
        raise TypeError(f"dflash_amd.argmax: bf16 or fp32 logits, got {logits.dtype}")
    x = logits.contiguous()
    v = x.shape[-1]
    rows = x.numel() // v
    out = torch.empty(x.shape[:-1], dtype=I64, device=x.device)
    check(lib().dfl_argmax(_p(x), 0 if x.dtype == BF16 else 1, rows, v, _p(out), _stream()), "dfl_argmax")
    return out


def accept_commit(block_ids, posterior, bs: int, output_ids, dyn, stop_ids=None, result=None, rearm=None, dyn_t=None) -> None:
    """result: int32[4] on the GPU, or PINNED host memory (the kernel stores the four words with one 16-byte store, a
    CPU thread may poll them).  rearm = (next_block int64 tensor, n, mask_id): the next cycle's block is written by
    the kernel (dfl_accept_commit_rearm)."""
    n_stop = 0 if stop_ids is None else stop_ids.numel()
    rp = None
    if result is not None:
        assert result.numel() >= 4 and result.dtype == I32
        if result.is_cuda:
            rp = _p(result, I32, "result")
        else:
            if not result.is_pinned():
                raise RuntimeError("dflash_amd: a host-side result buffer must be pinned memory")
            rp = result.data_ptr()
    common = (_p(block_ids, I64, "block_ids"), _p(posterior, I64, "posterior"), bs, _p(output_ids, I64, "output_ids"),
              output_ids.numel(), _p(dyn, I32, "dyn"), _p(stop_ids, I64, "stop_ids") if n_stop else None, n_stop, rp)
    if rearm is not None:
        nb, n, mask_id = rearm
        assert nb.numel() >= n
        if dyn_t is not None:    # the next verify's block-form record kept on the device too (graph replay)
            check(lib().dfl_accept_commit_rearm_t(*common, _p(nb, I64, "next_block"), int(n), int(mask_id),
                                                  _p(dyn_t, I32, "dyn_t"), _stream()), "dfl_accept_commit_rearm_t")
            return
        check(lib().dfl_accept_commit_rearm(*common, _p(nb, I64, "next_block"), int(n), int(mask_id), _stream()),
              "dfl_accept_commit_rearm")
        return
    check(lib().dfl_accept_commit(*common, _stream()), "dfl_accept_commit")


# ---- ragged batch of requests (include/dflash_hip.h, second half) ---------------------------
from ._lib import RowsBatch  # noqa: E402


def batch_tiles(R: int) -> int:
    return lib().dfl_batch_tiles(R)


def batch_ksplit(K: int) -> int:
    return lib().dfl_batch_ksplit(K)


class BatchRowSource:
    def __init__(self, struct: RowsBatch, *keep):
        self.struct, self.keep = struct, keep

    @property
    def ref(self):
        return _C.byref(self.struct)


def brows_frag(frag: torch.Tensor) -> BatchRowSource:
    """frag [MT, 16*K] bf16: request r's frag16 buffer is frag[r]."""
    assert frag.dim() == 2 and frag.is_contiguous()
    r0 = Rows(_p(frag, BF16, "frag"), None, 0, None, 0, None, 0.0, -1, 0)
    return BatchRowSource(RowsBatch(r0, frag.stride(0), 0, 0), frag)


def brows_plain(rows: torch.Tensor, valid_word: int = -1) -> BatchRowSource:
    """rows [MT, 16, K] bf16 (unit inner stride)."""
    assert rows.is_cuda and rows.dtype == BF16 and rows.dim() == 3 and rows.stride(2) == 1
    r0 = Rows(None, rows.data_ptr(), rows.stride(1), None, 0, None, 0.0, valid_word, 1)
    return BatchRowSource(RowsBatch(r0, 0, rows.stride(0), 0), rows)


def brows_normed(rows: torch.Tensor, ss: torch.Tensor, nss: int, norm_w: torch.Tensor, eps: float,
                 valid_word: int = -1) -> BatchRowSource:
    """rows [MT, 16, K] + ss [MT, >= nss*16] as a mode-2 source.  The batched GEMMs REJECT it
    (normalised rows come from norm_frag_batch); kept so that the rejection can be tested."""
    assert rows.is_cuda and rows.dtype == BF16 and rows.dim() == 3 and rows.stride(2) == 1
    assert ss.dim() == 2 and ss.shape[1] >= nss * 16 and ss.stride(1) == 1
    r0 = Rows(None, rows.data_ptr(), rows.stride(1), _p(ss[0], F32, "ss"), nss, _p(norm_w, BF16, "norm_w"), eps,
              valid_word, 2)
    return BatchRowSource(RowsBatch(r0, 0, rows.stride(0), ss.stride(0)), rows, ss, norm_w)


def gemm_batch_ws(N: int, K: int, device) -> torch.Tensor:
    return torch.zeros(lib().dfl_gemm_batch_ws_bytes(N, K), dtype=torch.uint8, device=device)


def gemm_f32_batch(wp, x: BatchRowSource, R: int, N: int, K: int, out: torch.Tensor, dyn) -> None:
    assert out.numel() >= batch_ksplit(K) * batch_tiles(R) * 16 * N
    check(lib().dfl_gemm_f32_batch(_p(wp, BF16, "wp"), x.ref, R, N, K, _p(out, F32, "out"), _p(dyn, I32, "dyn"),
                                   _stream()), "dfl_gemm_f32_batch")


def gemm_silu_mul_batch(wp_gu, x: BatchRowSource, R: int, I: int, K: int, act: torch.Tensor, ws, dyn) -> None:
    assert act.dim() == 2 and act.shape[1] >= 16 * I and act.shape[0] >= batch_tiles(R)
    check(lib().dfl_gemm_silu_mul_batch(_p(wp_gu, BF16, "wp_gu"), x.ref, R, I, K, _p(act, BF16, "act"), act.stride(0),
                                        _p(ws), _p(dyn, I32, "dyn"), _stream()), "dfl_gemm_silu_mul_batch")


def gemm_resid_batch(wp, x: BatchRowSource, R: int, N: int, K: int, h_io: torch.Tensor, *, add_residual: bool,
                     ws, dyn, ss_out=None, tap=None) -> None:
    """h_io [MT, 16, >=N]; tap: optional [MT, 16, *] view; ss_out [MT, >=N]."""
    assert h_io.is_cuda and h_io.dtype == BF16 and h_io.dim() == 3 and h_io.stride(2) == 1
    tp, ldt, tst = None, 0, 0
    if tap is not None:
        assert tap.is_cuda and tap.dtype == BF16 and tap.dim() == 3 and tap.stride(2) == 1
        tp, ldt, tst = tap.data_ptr(), tap.stride(1), tap.stride(0)
    sp, sst = None, 0
    if ss_out is not None:
        assert ss_out.dim() == 2 and ss_out.shape[1] >= N and ss_out.dtype == F32
        sp, sst = ss_out.data_ptr(), ss_out.stride(0)
    check(lib().dfl_gemm_resid_batch(_p(wp, BF16, "wp"), x.ref, R, N, K, h_io.data_ptr(), h_io.stride(1),
                                     h_io.stride(0), int(add_residual), tp, ldt, tst, sp, sst, _p(ws),
                                     _p(dyn, I32, "dyn"), _stream()), "dfl_gemm_resid_batch")


def gemm_argmax_batch(wp, x: BatchRowSource, R: int, V: int, K: int, row0: int, nrows: int, ws,
                      out_ids: torch.Tensor, out_off: int, dyn, nrows_dyn_word: int = -1,
                      logits: Optional[torch.Tensor] = None) -> None:
    """out_ids int64 [MT, n]: ids of request r's rows row0.. at out_ids[r, out_off..]."""
    assert out_ids.dim() == 2 and out_ids.dtype == I64 and out_ids.stride(1) == 1
    lp, lst = None, 0
    if logits is not None:
        assert logits.dim() == 3 and logits.shape[1] == 16 and logits.shape[2] == V and logits.is_contiguous()
        lp, lst = _p(logits, BF16, "logits"), logits.stride(0)
    check(lib().dfl_gemm_argmax_batch(_p(wp, BF16, "wp"), x.ref, R, V, K, row0, nrows, _p(dyn, I32, "dyn"),
                                      nrows_dyn_word, _p(ws), out_ids.data_ptr(), out_ids.stride(0), out_off, lp, lst,
                                      _stream()), "dfl_gemm_argmax_batch")


def embed_rows_batch(embed, ids: torch.Tensor, R: int, h_out: torch.Tensor, H: int, ss_out: torch.Tensor, dyn,
                     dyn_word: int) -> None:
    assert ids.dim() == 2 and ids.dtype == I64 and ids.stride(1) == 1 and h_out.dim() == 3 and ss_out.dim() == 2
    check(lib().dfl_embed_rows_batch(_p(embed, BF16, "embed"), ids.data_ptr(), ids.stride(0), R, h_out.data_ptr(),
                                     h_out.stride(0), H, ss_out.data_ptr(), ss_out.stride(0), _p(dyn, I32, "dyn"),
                                     dyn_word, _stream()), "dfl_embed_rows_batch")


def norm_frag_batch(h: torch.Tensor, R: int, norm_w: torch.Tensor, eps: float, frag: torch.Tensor, dyn,
                    dyn_word: int, part: Optional[torch.Tensor] = None, N: int = 0, K: int = 0,
                    tap: Optional[torch.Tensor] = None, nsplit: Optional[int] = None) -> None:
    """h [MT, 16, H] -> frag [MT, 16*H] = frag16 of the RMS-normalised rows.  part: the fp32
    partial sums gemm_f32_batch(N=H, K) left ([ksplit][MT*16][H]): added to h first (residual
    add, in place); tap: optional [MT, 16, *] view receiving the new rows."""
    assert h.dim() == 3 and h.stride(2) == 1 and frag.dim() == 2 and frag.is_contiguous()
    H = h.shape[2]
    pp, ns, ps, ldp = None, 0, 0, 0
    if part is not None:
        assert N == H and (K > 0 or nsplit)
        ns, ldp = (nsplit if nsplit else batch_ksplit(K)), N   # nsplit: partial sums that are not a K split (MoE expert shares)
        ps = batch_tiles(R) * 16 * N
        assert part.numel() >= ns * ps
        pp = _p(part, F32, "part")
    tp, ldt, tst = None, 0, 0
    if tap is not None:
        assert tap.is_cuda and tap.dtype == BF16 and tap.dim() == 3 and tap.stride(2) == 1
        tp, ldt, tst = tap.data_ptr(), tap.stride(1), tap.stride(0)
    check(lib().dfl_norm_frag_batch(_p(h[0, 0], BF16, "h"), h.stride(0), h.stride(1), R, pp, ns, ps, ldp, tp, ldt, tst,
                                    _p(norm_w, BF16, "norm_w"), eps, _p(frag, BF16, "frag"), frag.stride(0), H,
                                    _p(dyn, I32, "dyn"), dyn_word, _stream()), "dfl_norm_frag_batch")


def kv_append_batch(*, kv, nsplit, split_stride, ld, k_col, v_col, col_layer_stride, n_layers, R, n_kv, k_norm_w,
                    eps, cos_tab, sin_tab, kcache, vcache, dyn, tiles_per_req: int = 1) -> None:
    """kcache/vcache [MT, L, n_kv, rows, 128]; k_norm_w [L, 128] or None.  A 4-D cache [L, n_kv, rows, 128] is ONE
    request's cache shared by all R tiles (request stride 0): the tiles are then consecutive 16-row groups of that
    request's context rows (the large-M context prefill), each with its own S / pos0 in its dyn record."""
    assert kcache.shape == vcache.shape and kcache.shape[-1] == 128 and kcache.is_contiguous() and vcache.is_contiguous()
    if kcache.dim() == 4:
        rows, req_stride, layer_stride = kcache.shape[2], 0, kcache.stride(0)
    else:
        assert kcache.dim() == 5
        rows, req_stride, layer_stride = kcache.shape[3], kcache.stride(0), kcache.stride(1)
    if tiles_per_req != 1:    # R counts tiles, dyn holds one record per tile, tiles_per_req tiles share a request's cache
        check(lib().dfl_kv_append_batch_t(
            _p(kv, F32, "kv"), nsplit, split_stride, ld, k_col, v_col, col_layer_stride, n_layers, R, 16, n_kv,
            _p(k_norm_w, BF16, "k_norm_w"), 128, eps, _p(cos_tab, BF16, "cos"), _p(sin_tab, BF16, "sin"),
            cos_tab.shape[0], _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"), rows,
            req_stride, layer_stride, _p(dyn, I32, "dyn"), tiles_per_req, _stream()), "dfl_kv_append_batch_t")
        return
    check(lib().dfl_kv_append_batch(
        _p(kv, F32, "kv"), nsplit, split_stride, ld, k_col, v_col, col_layer_stride, n_layers, R, 16, n_kv,
        _p(k_norm_w, BF16, "k_norm_w"), 128, eps, _p(cos_tab, BF16, "cos"), _p(sin_tab, BF16, "sin"),
        cos_tab.shape[0], _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"), rows,
        req_stride, layer_stride, _p(dyn, I32, "dyn"), _stream()), "dfl_kv_append_batch")


def attn_fused_batch_ws(R: int, n_q: int, n_kv: int, max_splits: int, device) -> torch.Tensor:
    return torch.zeros(lib().dfl_attn_fused_batch_ws_bytes(R, n_q, n_kv, max_splits), dtype=torch.uint8,
                       device=device)


def attn_fused_batch(*, qkv, nsplit, split_stride, ld, q_col, k_col, v_col, R, n_q, n_kv, q_norm_w, k_norm_w, eps,
                     cos_tab, sin_tab, kcache, vcache, layer: int, scale, causal: bool, dyn, kv_len_max, ws,
                     max_splits, out_frag) -> None:
    """kcache/vcache [MT, L, n_kv, rows, 128] (layer `layer` is used); out_frag [MT, 16*n_q*128]."""
    assert kcache.shape == vcache.shape and kcache.dim() == 5 and kcache.shape[4] == 128 and kcache.is_contiguous()
    assert out_frag.dim() == 2 and out_frag.is_contiguous()
    kc, vc = kcache[0, layer], vcache[0, layer]
    check(lib().dfl_attn_fused_batch(
        _p(qkv, F32, "qkv"), nsplit, split_stride, ld, q_col, k_col, v_col, 0, 16, R, n_q, n_kv,
        _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps, _p(cos_tab, BF16, "cos"),
        _p(sin_tab, BF16, "sin"), cos_tab.shape[0], kc.data_ptr(), vc.data_ptr(), kcache.shape[3], kcache.stride(0),
        scale, int(causal), _p(dyn, I32, "dyn"), kv_len_max, _p(ws), max_splits, _p(out_frag, BF16, "out_frag"),
        out_frag.stride(0), _stream()), "dfl_attn_fused_batch")


def accept_commit_batch(block: torch.Tensor, posterior: torch.Tensor, R: int, output_ids: torch.Tensor, dyn_d, dyn_t,
                        stop_ids, result: torch.Tensor, rearm_mask_id: Optional[int] = None, tiles_per_req: int = 1,
                        dyn_d_tiles=None, dyn_t_tiles=None) -> None:
    """block/posterior int64 [requests, 16 * tiles_per_req]; output_ids int64 [requests, n]; result int32 [requests, 4];
    tiles_per_req = 2: the per-tile records dyn_d_tiles / dyn_t_tiles are kept as well (dfl_accept_commit_batch_t)."""
    assert block.dim() == 2 and posterior.dim() == 2 and output_ids.dim() == 2
    n_stop = 0 if stop_ids is None else stop_ids.numel()
    if tiles_per_req != 1:
        check(lib().dfl_accept_commit_batch_t(
            _p(block, I64, "block"), block.stride(0), _p(posterior, I64, "posterior"), posterior.stride(0), R,
            _p(output_ids, I64, "output_ids"), output_ids.stride(0), output_ids.shape[1], _p(dyn_d, I32, "dyn_d"),
            _p(dyn_t, I32, "dyn_t"), _p(stop_ids, I64, "stop_ids") if n_stop else None, n_stop,
            _p(result, I32, "result"), block.data_ptr() if rearm_mask_id is not None else None,
            int(rearm_mask_id) if rearm_mask_id is not None else 0, tiles_per_req, _p(dyn_d_tiles, I32, "dyn_d_tiles"),
            _p(dyn_t_tiles, I32, "dyn_t_tiles"), _stream()), "dfl_accept_commit_batch_t")
        return
    check(lib().dfl_accept_commit_batch(
        _p(block, I64, "block"), block.stride(0), _p(posterior, I64, "posterior"), posterior.stride(0), R,
        _p(output_ids, I64, "output_ids"), output_ids.stride(0), output_ids.shape[1], _p(dyn_d, I32, "dyn_d"),
        _p(dyn_t, I32, "dyn_t"), _p(stop_ids, I64, "stop_ids") if n_stop else None, n_stop,
        _p(result, I32, "result"), block.data_ptr() if rearm_mask_id is not None else None,
        int(rearm_mask_id) if rearm_mask_id is not None else 0, _stream()), "dfl_accept_commit_batch")


# ---- target prefill (csrc/prefill.hip)
def prefill_rows_padded(P: int) -> int:
    return lib().dfl_prefill_rows_padded(P)


def prefill_gemm_rows(wp, x_frag, P: int, N: int, K: int, out: torch.Tensor) -> None:
    assert out.dtype == BF16 and out.stride(1) == 1 and out.shape[0] >= prefill_rows_padded(P)
    check(lib().dfl_prefill_gemm_rows(_p(wp, BF16, "wp"), _p(x_frag, BF16, "x_frag"), P, N, K, out.data_ptr(),
                                      out.stride(0), _stream()), "dfl_prefill_gemm_rows")


def prefill_gemm_resid(wp, x_frag, P: int, N: int, K: int, h_io: torch.Tensor, tap=None) -> None:
    assert h_io.dtype == BF16 and h_io.stride(1) == 1 and h_io.shape[0] >= prefill_rows_padded(P)
    tp, ldt = None, 0
    if tap is not None:
        assert tap.is_cuda and tap.dtype == BF16 and tap.stride(1) == 1 and tap.shape[0] >= P
        tp, ldt = tap.data_ptr(), tap.stride(0)
    check(lib().dfl_prefill_gemm_resid(_p(wp, BF16, "wp"), _p(x_frag, BF16, "x_frag"), P, N, K, h_io.data_ptr(),
                                       h_io.stride(0), tp, ldt, _stream()), "dfl_prefill_gemm_resid")


def prefill_gemm_silu(wp_gu, x_frag, P: int, I: int, K: int, act_frag: torch.Tensor) -> None:
    assert act_frag.numel() >= prefill_rows_padded(P) * I
    check(lib().dfl_prefill_gemm_silu(_p(wp_gu, BF16, "wp_gu"), _p(x_frag, BF16, "x_frag"), P, I, K,
                                      _p(act_frag, BF16, "act_frag"), _stream()), "dfl_prefill_gemm_silu")


def prefill_norm_pack(h: torch.Tensor, P: int, H: int, norm_w, eps: float, x_frag: torch.Tensor) -> None:
    assert h.dtype == BF16 and h.stride(1) == 1 and h.shape[0] >= P and x_frag.numel() >= prefill_rows_padded(P) * H
    check(lib().dfl_prefill_norm_pack(h.data_ptr(), h.stride(0), P, H, _p(norm_w, BF16, "norm_w"), eps,
                                      _p(x_frag, BF16, "x_frag"), _stream()), "dfl_prefill_norm_pack")


def prefill_pack_rows(rows: torch.Tensor, P: int, K: int, x_frag: torch.Tensor) -> None:
    """rows [P, K] -> frag16 row tiles (any K % 32 == 0), no norm."""
    assert rows.dtype == BF16 and rows.stride(1) == 1 and rows.shape[0] >= P and x_frag.numel() >= prefill_rows_padded(P) * K
    check(lib().dfl_prefill_pack_rows(rows.data_ptr(), rows.stride(0), P, K, _p(x_frag, BF16, "x_frag"), _stream()),
          "dfl_prefill_pack_rows")


def prefill_qk_rope(qkv: torch.Tensor, P: int, q_col: int, k_col: int, v_col: int, n_q: int, n_kv: int, q_norm_w,
                    k_norm_w, eps: float, cos_tab, sin_tab, pos0: int, kcache, vcache, row0: int) -> None:
    assert qkv.dtype == BF16 and qkv.stride(1) == 1 and kcache.shape == vcache.shape and kcache.shape[2] == 128
    check(lib().dfl_prefill_qk_rope(qkv.data_ptr(), qkv.stride(0), P, q_col, k_col, v_col, n_q, n_kv,
                                    _p(q_norm_w, BF16, "q_norm_w"), _p(k_norm_w, BF16, "k_norm_w"), eps,
                                    _p(cos_tab, BF16, "cos"), _p(sin_tab, BF16, "sin"), cos_tab.shape[0], pos0,
                                    _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"), kcache.shape[1], row0,
                                    _stream()), "dfl_prefill_qk_rope")


def prefill_attn(qkv: torch.Tensor, P: int, q_col: int, kcache, vcache, n_q: int, n_kv: int, scale: float,
                 out_frag: torch.Tensor) -> None:
    """Causal attention of the prompt rows over themselves -> frag16 row tiles of n_q * 128 columns."""
    assert qkv.dtype == BF16 and qkv.stride(1) == 1 and kcache.shape == vcache.shape and kcache.shape[2] == 128
    assert out_frag.numel() >= (P + 15) // 16 * 16 * n_q * 128
    check(lib().dfl_prefill_attn(qkv.data_ptr(), qkv.stride(0), q_col, _p(kcache, BF16, "kcache"), _p(vcache, BF16, "vcache"),
                                 kcache.shape[1], P, n_q, n_kv, scale, _p(out_frag, BF16, "out_frag"), _stream()),
          "dfl_prefill_attn")


# ---- sparse-MoE MLP of the prefill (csrc/prefill.hip: k_pmoe_*, the grouped k_pgemm) ----------------------------
def prefill_moe_scratch(P: int, H: int, I: int, E: int, top_k: int, router_cols: int, device) -> dict:
    """Caller-owned scratch of dfl_prefill_moe_* for P prompt rows (sizes in include/dflash_hip.h)."""
    L = lib()
    mt, mi = int(L.dfl_prefill_moe_max_tiles(P, top_k, E)), int(L.dfl_prefill_moe_max_items(P, top_k, E))
    zi = lambda *s: torch.zeros(*s, dtype=I32, device=device)    # noqa: E731
    zf = lambda *s: torch.zeros(*s, dtype=F32, device=device)    # noqa: E731
    # (128-row work items where an expert serves ~48 rows or more on average, see include/dflash_hip.h)
    return dict(P=P, H=H, I=I, E=E, top_k=top_k, max_tiles=mt, max_items=mi, rows_per_item=128 if P * top_k >= 48 * E else 64,
                rlog=torch.zeros(prefill_rows_padded(P), router_cols, dtype=BF16, device=device),
                pair_e=zi(P, 8), posmap=zi(P, 8), pair_w=zf(P, 8), cnt=zi(E), tile_off=zi(E), src_row=zi(mt * 16),
                items=zi(3 * mi), n_items=zi(2), xg=torch.zeros(mt * 16 * H, dtype=BF16, device=device),
                act_g=torch.zeros(mt * 16 * I, dtype=BF16, device=device), row_w=zf(mt * 16), out32=zf(mt * 16, H),
                zeros=torch.zeros(512, dtype=BF16, device=device))


def prefill_moe_route(rlog: torch.Tensor, P: int, sc: dict, norm_topk: bool = True) -> None:
    """Routing of the P rows from their router logits (bf16 rows) + the plan of the grouped GEMMs, into the scratch."""
    check(lib().dfl_prefill_moe_route(rlog.data_ptr(), rlog.stride(0), P, sc["E"], sc["top_k"], int(bool(norm_topk)),
                                      sc["pair_e"].data_ptr(), sc["pair_w"].data_ptr(), sc["cnt"].data_ptr(),
                                      sc["tile_off"].data_ptr(), sc["items"].data_ptr(), sc["n_items"].data_ptr(),
                                      sc["posmap"].data_ptr(), sc["src_row"].data_ptr(), sc["row_w"].data_ptr(),
                                      sc["rows_per_item"], _stream()),
          "dfl_prefill_moe_route")


def prefill_moe_mlp(router_wp, gu_e, down_e, x_frag, P: int, H: int, I: int, E: int, top_k: int, norm_topk: bool,
                    h_io: torch.Tensor, sc: dict, tap=None) -> None:
    """Qwen3MoeSparseMoeBlock over the P prompt rows of one layer (tf:models/qwen3_moe/modeling_qwen3_moe.py):
    x_frag = the ln2-normalised rows as frag16 tiles; h_io <- h_io + MLP(x) (+ tap copy).  router_wp: the gate Linear's
    weight packed with its rows padded to a multiple of 128; gu_e / down_e: [E, ...] packed expert weights."""
    assert sc["P"] >= P and sc["H"] == H and sc["I"] == I and sc["E"] == E and sc["top_k"] == top_k
    assert gu_e.shape[0] == E and down_e.shape[0] == E and h_io.dtype == BF16 and h_io.stride(1) == 1
    L, st = lib(), _stream()
    rlog = sc["rlog"]
    prefill_gemm_rows(router_wp, x_frag, P, rlog.shape[1], H, rlog)
    prefill_moe_route(rlog, P, sc, norm_topk)
    # The rows are gathered per expert into a copy (xg) that the expert's column blocks then read coalesced.  Gathering
    # inside the gate/up GEMM's LDS-DMA addresses instead (src_row: no gather launch, no copy) was measured at P = 1024
    # on the 30B-A3B widths, same box: 455 -> 479 us per layer — every column block re-reads the rows as scattered 16-byte
    # pieces; it is what the <= 64 decode rows of NativeTarget._moe_mlp_shared use (241 -> 233 us there).
    check(L.dfl_prefill_moe_gather(_p(x_frag, BF16, "x_frag"), P, H, top_k, E, sc["src_row"].data_ptr(), sc["n_items"].data_ptr(),
                                   sc["xg"].data_ptr(), st), "dfl_prefill_moe_gather")
    check(L.dfl_prefill_moe_gemm_silu(_p(gu_e, BF16, "gu_e"), gu_e.stride(0), sc["xg"].data_ptr(), sc["items"].data_ptr(),
                                      sc["n_items"].data_ptr(), sc["max_items"], I, H, sc["act_g"].data_ptr(),
                                      sc["rows_per_item"], None, None, st), "dfl_prefill_moe_gemm_silu")
    check(L.dfl_prefill_moe_gemm_down(_p(down_e, BF16, "down_e"), down_e.stride(0), sc["act_g"].data_ptr(),
                                      sc["items"].data_ptr(), sc["n_items"].data_ptr(), sc["max_items"], H, I,
                                      sc["row_w"].data_ptr(), sc["out32"].data_ptr(), sc["rows_per_item"], st),
          "dfl_prefill_moe_gemm_down")
    tp, ldt = (None, 0) if tap is None else (_p(tap, BF16, "tap") if tap.is_contiguous() else tap.data_ptr(), tap.stride(0))
    check(L.dfl_prefill_moe_combine(sc["out32"].data_ptr(), sc["posmap"].data_ptr(), P, H, top_k, h_io.data_ptr(),
                                    h_io.stride(0), tp, ldt, None, st), "dfl_prefill_moe_combine")
