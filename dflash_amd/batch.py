"""Ragged batch of requests on one GPU: R <= 4 requests advance one decode cycle together
and share every weight byte of the draft forward, the lm_head and the target verify
(BASELINE.json configs[2]: 32 requests over 8 GPUs = 4 per GPU).

The reference has no batched form of the path: `spec_generate` is batch-1 by construction
(model/dflash.py:206-211,258) and its "batched" harness is a Python loop over prompts
(benchmark_batched.py:212-243, benchmark.py:445-470).  `dflash_generate_batch` therefore
keeps that contract — a list of prompts in, one `dflash_generate` namespace per prompt out,
each identical to what the single-request loop returns for that prompt — and changes only
how the cycles are executed: per cycle ONE pass over the weights for all live requests
(dfl_*_batch kernels), per-request lengths S / tau / start held on the device (`dyn`),
per-request preallocated KV caches, one device->host read of R x 4 ints.

Needs the native target (`dflash_amd.NativeTarget`): the HF forward cannot take requests
of different lengths without padding masks, which is exactly the cost this path removes.
Sparse-MoE targets (round 3): attention and the dense projections run batched, the expert
MLP per request (`BatchedDecoder._moe_mlp`).
"""
from __future__ import annotations

import os

from types import SimpleNamespace
from typing import Callable, Optional, Sequence

import torch

from . import ops
from .generate import _bf16_table, _taps, _trim, capture_graph, cuda_time
from .model import DFlashDraftModel
from .target import NativeTarget
from .utils import sample

BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64
MAX_GROUP = 4


class _View:
    """One request's slice of the group caches, with the fields the single-request
    prefill code expects (DFlashKVCache / TargetKVCache)."""

    def __init__(self, k, v, dyn, max_rows):
        self.k, self.v, self.dyn, self.max_rows, self.length = k, v, dyn, max_rows, 0

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.length


class BatchedDecoder:
    """Device state and per-cycle launch sequence of one group of R <= 4 requests."""

    def __init__(self, model: DFlashDraftModel, target: NativeTarget, n_requests: int, max_rows: int,
                 out_len: int, mask_token_id: int, stop_token_ids=None, max_splits: int = 32,
                 temperature: float = 0.0, tiles_per_request: int = 1):
        if not isinstance(target, NativeTarget):
            raise TypeError("BatchedDecoder needs a dflash_amd.NativeTarget (see module docstring)")
        if tiles_per_request not in (1, 2):
            raise ValueError("tiles_per_request is 1 (blocks of <= 16 rows) or 2 (blocks of <= 32 rows)")
        if not 1 <= n_requests * tiles_per_request <= MAX_GROUP:
            raise ValueError(f"a group holds 1..{MAX_GROUP} sixteen-row tiles (requests x tiles_per_request)")
        if tiles_per_request == 2 and temperature >= 1e-5:
            raise NotImplementedError("blocks of more than 16 rows in the ragged batch are greedy (T = 0)")
        if model.w is None:
            raise RuntimeError("draft weights not loaded")
        c, t = model.config, target
        if c.hidden_size != t.H:
            raise ValueError("draft and target hidden sizes differ")
        self.model, self.target, self.cfg = model, target, c
        # Blocks of 17..32 rows (benchmark.py's block-size sweep): a request takes TPR = 2 consecutive 16-row tiles of every
        # per-tile launch (GEMMs, norms, embedding, context K/V append), one cache, one slot of the attention / accept
        # launches.  Two kinds of length records then: per REQUEST (dyn_d / dyn_t: attention, accept) and per TILE (dyn_dt /
        # dyn_tt: valid rows of each tile), both kept by dfl_accept_commit_batch_t.  TPR = 1: the same tensors.
        self.TPR = TPR = tiles_per_request
        self.R, self.NT = n_requests, n_requests * TPR
        self.MT = ops.batch_tiles(self.NT)
        self.BW = 16 * TPR                    # slots of a request's block
        NREQ = self.MT // TPR                 # request slots of the caches / id buffers
        self.dev = dev = model.device
        self.max_rows, self.out_len, self.mask_id = int(max_rows), int(out_len), int(mask_token_id)
        self.max_splits = max_splits
        self.temperature = float(temperature)
        self._logits = None
        R, MT, H, I = self.R, self.MT, c.hidden_size, c.intermediate_size
        z = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731

        # ---- lengths (device): draft form and block form, see dfl_accept_commit_batch
        self.dyn_d, self.dyn_t = z(MT, 8, dt=I32), z(MT, 8, dt=I32)
        self.dyn_dt, self.dyn_tt = (self.dyn_d, self.dyn_t) if TPR == 1 else (z(MT, 8, dt=I32), z(MT, 8, dt=I32))
        # ---- caches [request][layer][kv head][row][128]
        Ld, Lt = c.num_hidden_layers, t.L
        self.dk, self.dv = z(NREQ, Ld, c.num_key_value_heads, max_rows, 128), z(NREQ, Ld, c.num_key_value_heads, max_rows, 128)
        self.tk, self.tv = z(NREQ, Lt, t.n_kv, max_rows, 128), z(NREQ, Lt, t.n_kv, max_rows, 128)
        # ---- ids
        self.block = torch.full((NREQ, self.BW), self.mask_id, dtype=I64, device=dev)
        self.post = z(NREQ, self.BW, dt=I64)
        self.ids_tmp = z(MT, 16, dt=I64)      # TPR = 2: the draft's ids of every tile row (row 0 of tile 0 is not a draft token)
        self.result = z(MT, 4, dt=I32)
        self.output_ids = torch.full((NREQ, out_len), self.mask_id, dtype=I64, device=dev)
        self.stop_t = torch.tensor(stop_token_ids, dtype=I64, device=dev) if stop_token_ids else None
        # ---- draft scratch
        self.nqkv_d = c.q_dim + 2 * c.kv_dim
        self.nkv_all = Ld * 2 * c.kv_dim
        ks = ops.batch_ksplit
        self.d = dict(h=z(MT, 16, H), ctxh=z(MT, 16, H), ss_emb=z(MT, 16, dt=F32), xn=z(MT, 16 * H),
                      attn=z(MT, 16 * c.q_dim), act=z(MT, 16 * I),
                      part_qkv=z(ks(H) * MT * 16 * self.nqkv_d, dt=F32), part_kv=z(ks(H) * MT * 16 * self.nkv_all, dt=F32),
                      part_h=z(max(ks(c.q_dim), ks(I)) * MT * 16 * H, dt=F32), taps=z(MT, 16, c.fc_in))
        # context K/V weights of all layers as ONE packed weight: the k/v column tiles of each
        # layer's packed qkv, concatenated (tile-major layout: a plain cat of tile ranges)
        L = model.w["layers"]
        self.kv_all = torch.cat([lw["qkv"][c.q_dim * H:] for lw in L]).contiguous()
        self.k_norm_all = torch.stack([lw["k_norm"] for lw in L]).contiguous()
        # ---- target scratch
        self.nqkv_t = t.nqkv
        self.t = dict(h=z(MT, 16, H), ss_emb=z(MT, 16, dt=F32), xn=z(MT, 16 * H), attn=z(MT, 16 * t.q_dim),
                      act=z(MT, 16 * t.I), part_qkv=z(ks(H) * MT * 16 * t.nqkv, dt=F32),
                      part_h=z(max(ks(t.q_dim), ks(t.I), getattr(t, "moe_nsplit", 0) if getattr(t, "is_moe", False) else 0)
                               * MT * 16 * H, dt=F32))
        # ---- shared workspaces (launches are stream-ordered)
        nmax = max(c.vocab_size, t.V, 2 * I, 2 * t.I, self.nqkv_d, t.nqkv)
        kmax = max(H, I, t.I, c.fc_in, c.q_dim)
        self.gws = torch.zeros(max(ops.lib().dfl_gemm_batch_ws_bytes(n, k) for n, k in
                                   ((nmax, H), (H, kmax))), dtype=torch.uint8, device=dev)
        # one per model: the arrival tickets sit behind the partials, whose size depends on n_q
        self.aws_d = ops.attn_fused_batch_ws(MT, c.num_attention_heads, c.num_key_value_heads, max_splits, dev)
        self.aws_t = ops.attn_fused_batch_ws(MT, t.n_q, t.n_kv, max_splits, dev)
        # round 2: the attention stage on finished bf16 q/k/v rows (dfl_attn_head_batch); "fused" keeps the round-1 stage
        self.attn_impl = getattr(model, "attn_impl", "head")
        # round 4: the q/k/v projection leaves fp32 K-part sums and the attention launch sums them while it loads its rows
        # (dfl_attn_head_batch_f32) — blocks of <= 16 rows, <= 2 K parts (hidden <= 4096); DFL_QKV_PARTS=0: the round-3 form
        # (finished bf16 rows: slabs + ticket + combine inside the GEMM), kept as the second implementation
        self.qkv_parts = (os.environ.get("DFL_QKV_PARTS", "1") != "0" and tiles_per_request == 1 and ops.batch_ksplit(H) <= 2)
        self.hws_d = ops.attn_head_batch_ws(MT, c.num_attention_heads, max_splits, dev, q_tiles=TPR)
        self.hws_t = ops.attn_head_batch_ws(MT, t.n_q, max_splits, dev, q_tiles=TPR)
        if TPR == 2 and self.attn_impl != "head":
            raise NotImplementedError("blocks of more than 16 rows need the 'head' attention stage")
        self.d["xq"], self.t["xq"] = z(MT, 16, self.nqkv_d), z(MT, 16, t.nqkv)
        # ---- row sources
        # (normalised operands come from dfl_norm_frag_batch: at 4 tiles the in-GEMM norm of the
        # single-request path, replicated in every workgroup, costs more than that launch)
        d, tt = self.d, self.t
        self.src_d = dict(taps=ops.brows_plain(d["taps"], ops.DYN_TAU), xn=ops.brows_frag(d["xn"]),
                          attn=ops.brows_frag(d["attn"]), act=ops.brows_frag(d["act"]))
        self.src_t = dict(xn=ops.brows_frag(tt["xn"]), attn=ops.brows_frag(tt["attn"]), act=ops.brows_frag(tt["act"]))
        self.lm_wp = None
        self.embed_w = None
        # ---- host mirror of the lengths
        self.start = [0] * R
        self.n_in = [0] * R
        self.live = [False] * R
        self.hook_calls = [0] * R
        self.bs = [self.BW] * R
        self.events = None  # set to a dict to have cycle() record (start, end) event pairs per phase
        self._ahead, self._ahead_ev = False, None   # a run-ahead draft is in flight (cycle(ahead_ok=True))
        self.run_ahead = os.environ.get("DFL_RUN_AHEAD", "1") != "0"

    def _mark(self, key, which):
        if self.events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.setdefault(key, [None, None])[which] = e

    # ------------------------------------------------------------------ admission
    @torch.inference_mode()
    def admit(self, r: int, input_ids: torch.Tensor, temperature: float = 0.0) -> None:
        """Prefill request r (model/dflash.py:218-229): target prefill through the wrapped
        model, K/V into the group cache, first token sampled, the prompt's context rows
        projected into the draft cache except the last <= 16, which become the first
        cycle's context tile."""
        m, t, c = self.model, self.target, self.cfg
        if input_ids.shape[0] != 1 or not input_ids.is_cuda:
            raise ValueError("admit: input_ids must be a [1, P] GPU tensor")
        P = input_ids.shape[1]
        if P + 2 * self.BW > self.max_rows or P + 1 > self.out_len:
            raise ValueError("admit: prompt does not fit the group's caches")
        if self.lm_wp is None:
            self.lm_wp = m.packed_lm_head(t.lm_head)
            t.share_lm_head(self.lm_wp)
            self.embed_w = _bf16_table(t.model.embed_tokens.weight, self.dev)
        tc = _View(self.tk[r], self.tv[r], None, self.max_rows)
        out = t.prefill(input_ids, tc, output_hidden_states=True, tap_layers=self.model.target_layer_ids)
        self.output_ids[r].fill_(self.mask_id)
        self.output_ids[r, :P] = input_ids[0]
        first = sample(out.logits, temperature)
        self.output_ids[r, P:P + 1] = first[0]
        th = _taps(out.hidden_states, m.target_layer_ids)[0]          # [P, fc_in]
        n_tail = min(16, P)
        dc = _View(self.dk[r], self.dv[r], torch.zeros(8, dtype=I32, device=self.dev), self.max_rows)
        if P > n_tail:
            m.prefill_context(dc, th[:P - n_tail], 0)
        t0, BW = r * self.TPR, self.BW
        self.d["taps"][t0:t0 + self.TPR].zero_()
        self.d["taps"][t0, :n_tail] = th[P - n_tail:]      # (the tail rows fit the request's first tile)
        self.block[r].fill_(self.mask_id)
        self.block[r, 0:1] = first[0]
        S = P - n_tail
        self.dyn_d[r] = torch.tensor([S, n_tail, BW, S, P, 0, 0, 0], dtype=I32)
        self.dyn_t[r] = torch.tensor([P, 0, BW, P, P, 0, 0, 0], dtype=I32)
        if self.TPR == 2:
            for j in range(2):
                self.dyn_dt[t0 + j] = torch.tensor([S + 16 * j, n_tail if j == 0 else 0, 0, S + 16 * j, P, 0, 0, 0], dtype=I32)
                self.dyn_tt[t0 + j] = torch.tensor([P, 0, 16, P, P, 0, 0, 0], dtype=I32)
        self.start[r], self.n_in[r], self.live[r], self.hook_calls[r], self.bs[r] = P, P, True, 0, BW

    def park(self, r: int) -> None:
        """Request r is finished: its tile stays in the launches but does no work."""
        self.live[r] = False
        self.dyn_d[r, ops.DYN_TAU:ops.DYN_BS + 1] = 0
        self.dyn_t[r, ops.DYN_BS] = 0
        if self.TPR == 2:
            self.dyn_dt[2 * r:2 * r + 2, ops.DYN_TAU:ops.DYN_BS + 1] = 0
            self.dyn_tt[2 * r:2 * r + 2, ops.DYN_BS] = 0

    def set_block_size(self, r: int, bs: int) -> None:
        """Tail clamp (benchmark.py:104-105): a host write only when the size changes."""
        if bs != self.bs[r]:
            self.dyn_d[r, ops.DYN_BS] = bs
            self.dyn_t[r, ops.DYN_BS] = bs
            if self.TPR == 2:
                self.dyn_tt[2 * r, ops.DYN_BS] = min(bs, 16)
                self.dyn_tt[2 * r + 1, ops.DYN_BS] = max(0, bs - 16)
            self.bs[r] = bs

    # ------------------------------------------------------------------ one cycle
    def _kv_len_max(self) -> int:
        return max([self.start[r] for r in range(self.R) if self.live[r]] + [1]) + self.BW

    def draft(self) -> None:
        """Draft forward + lm_head + greedy unmask for every live request
        (model/dflash.py:237-247): block[r, 1:bs] <- argmax."""
        self._draft_body(self._kv_len_max())
        self._mark("lm_head", 0)
        self._draft_head()
        self._mark("lm_head", 1)

    def _draft_body(self, kvmax: int) -> None:
        m, c, d, s, MT = self.model, self.cfg, self.d, self.src_d, self.MT
        R, RQ, TPR = self.NT, self.R, self.TPR     # R: 16-row tiles of the per-tile launches; RQ: requests (attention)
        dyn_d, dyn_t = self.dyn_dt, self.dyn_tt    # per-tile records (the per-request ones when TPR = 1)
        H, I = c.hidden_size, c.intermediate_size
        L = m.w["layers"]
        cos, sin = m._rope_tab(kvmax + 64)
        ops.embed_rows_batch(self.embed_w, self.block.view(-1, 16), R, d["h"], H, d["ss_emb"], dyn_t, ops.DYN_BS)
        # context rows: fc, then K/V of all layers appended to the draft caches
        eps = c.rms_norm_eps
        ops.gemm_resid_batch(m.w["fc"], s["taps"], R, H, c.fc_in, d["ctxh"], add_residual=False, ws=self.gws,
                             dyn=dyn_d)
        ops.norm_frag_batch(d["ctxh"], R, m.w["hidden_norm"], eps, d["xn"], dyn_d, ops.DYN_TAU)
        ops.gemm_f32_batch(self.kv_all, s["xn"], R, self.nkv_all, H, d["part_kv"], dyn_d)
        nsp = ops.batch_ksplit(H)
        ops.kv_append_batch(kv=d["part_kv"], nsplit=nsp, split_stride=MT * 16 * self.nkv_all, ld=self.nkv_all, k_col=0,
                            v_col=c.kv_dim, col_layer_stride=2 * c.kv_dim, n_layers=c.num_hidden_layers, R=R,
                            n_kv=c.num_key_value_heads, k_norm_w=self.k_norm_all, eps=c.rms_norm_eps, cos_tab=cos,
                            sin_tab=sin, kcache=self.dk, vcache=self.dv, dyn=dyn_d, tiles_per_req=TPR)
        # block rows.  o_proj / down_proj leave fp32 K-part sums; the residual add happens in the
        # norm launch that follows (the parts meet at the launch boundary, not inside the GEMM)
        pend = 0  # K of the GEMM whose sums are waiting in part_h (0: none)
        for i, lw in enumerate(L):
            ops.norm_frag_batch(d["h"], R, lw["ln1"], eps, d["xn"], dyn_t, ops.DYN_BS,
                                part=d["part_h"] if pend else None, N=H, K=pend)
            if self.attn_impl == "head" and self.qkv_parts:
                # q/k/v as fp32 K-part sums; the parts meet in the attention launch's row loads (round 4)
                ops.gemm_f32_batch(lw["qkv"], s["xn"], R, self.nqkv_d, H, d["part_qkv"], dyn_t)
                ops.attn_head_batch_f32(qkv_parts=d["part_qkv"], nparts=nsp, MT=MT, ld=self.nqkv_d, q_col=0, k_col=c.q_dim,
                                        v_col=c.q_dim + c.kv_dim, R=RQ, n_q=c.num_attention_heads,
                                        n_kv=c.num_key_value_heads, q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"],
                                        eps=c.rms_norm_eps, cos_tab=cos, sin_tab=sin, kcache=self.dk, vcache=self.dv, layer=i,
                                        scale=c.head_dim ** -0.5, causal=False, dyn=self.dyn_t, kv_len_max=kvmax,
                                        ws=self.hws_d, max_splits=self.max_splits, out_frag=d["attn"])
            elif self.attn_impl == "head":
                ops.gemm_resid_batch(lw["qkv"], s["xn"], R, self.nqkv_d, H, d["xq"], add_residual=False, ws=self.gws,
                                     dyn=dyn_t)
                ops.attn_head_batch(xq=d["xq"], q_col=0, k_col=c.q_dim, v_col=c.q_dim + c.kv_dim, R=RQ,
                                    n_q=c.num_attention_heads, n_kv=c.num_key_value_heads, q_norm_w=lw["q_norm"],
                                    k_norm_w=lw["k_norm"], eps=c.rms_norm_eps, cos_tab=cos, sin_tab=sin, kcache=self.dk,
                                    vcache=self.dv, layer=i, scale=c.head_dim ** -0.5, causal=False, dyn=self.dyn_t,
                                    kv_len_max=kvmax, ws=self.hws_d, max_splits=self.max_splits, out_frag=d["attn"],
                                    q_tiles=TPR)
            else:
                ops.gemm_f32_batch(lw["qkv"], s["xn"], R, self.nqkv_d, H, d["part_qkv"], dyn_t)
                ops.attn_fused_batch(qkv=d["part_qkv"], nsplit=nsp, split_stride=MT * 16 * self.nqkv_d, ld=self.nqkv_d,
                                     q_col=0, k_col=c.q_dim, v_col=c.q_dim + c.kv_dim, R=R, n_q=c.num_attention_heads,
                                     n_kv=c.num_key_value_heads, q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"],
                                     eps=c.rms_norm_eps, cos_tab=cos, sin_tab=sin, kcache=self.dk, vcache=self.dv,
                                     layer=i, scale=c.head_dim ** -0.5, causal=False, dyn=dyn_t, kv_len_max=kvmax,
                                     ws=self.aws_d, max_splits=self.max_splits, out_frag=d["attn"])
            ops.gemm_f32_batch(lw["o"], s["attn"], R, H, c.q_dim, d["part_h"], dyn_t)
            ops.norm_frag_batch(d["h"], R, lw["ln2"], eps, d["xn"], dyn_t, ops.DYN_BS, part=d["part_h"], N=H,
                                K=c.q_dim)
            ops.gemm_silu_mul_batch(lw["gu"], s["xn"], R, I, H, d["act"], self.gws, dyn_t)
            ops.gemm_f32_batch(lw["down"], s["act"], R, H, I, d["part_h"], dyn_t)
            pend = I

    def _draft_head(self) -> None:
        m, c, d, s, R = self.model, self.cfg, self.d, self.src_d, self.NT
        ops.norm_frag_batch(d["h"], R, m.w["norm"], c.rms_norm_eps, d["xn"], self.dyn_tt, ops.DYN_BS,
                            part=d["part_h"], N=c.hidden_size, K=c.intermediate_size)  # last down_proj + final norm
        if self.TPR == 1:
            ops.gemm_argmax_batch(self.lm_wp, s["xn"], R, c.vocab_size, c.hidden_size, 1, 15, self.gws, self.block, 1,
                                  self.dyn_tt, nrows_dyn_word=ops.DYN_BS)
        else:   # row 0 of a request's SECOND tile is a draft row: all 16 rows of every tile, row 0 of the block dropped here
            ops.gemm_argmax_batch(self.lm_wp, s["xn"], R, c.vocab_size, c.hidden_size, 0, 16, self.gws, self.ids_tmp, 0,
                                  self.dyn_tt, nrows_dyn_word=ops.DYN_BS)
            self.block[:, 1:].copy_(self.ids_tmp.view(-1, self.BW)[:, 1:])

    def verify(self, kvmax: Optional[int] = None) -> None:
        """Target verify of every live request's block (model/dflash.py:249-257, T = 0):
        post[r] <- the target's greedy tokens, taps[r] <- the tapped layers' hidden rows."""
        t, tt, s, R, MT, H = self.target, self.t, self.src_t, self.NT, self.MT, self.cfg.hidden_size
        RQ, TPR, dyn_t = self.R, self.TPR, self.dyn_tt    # (R: tiles, RQ: requests, dyn_t: per-tile records, see _draft_body)
        kvmax = kvmax or self._kv_len_max()
        cos, sin = t._rope_tab(kvmax + 64)
        taps = self.d["taps"]
        tl = list(self.model.target_layer_ids)
        if max(tl) >= t.L - 1:
            raise NotImplementedError("tapping the last layer (post-norm state) is not supported")
        nsp = ops.batch_ksplit(H)
        ops.embed_rows_batch(t.embed, self.block.view(-1, 16), R, tt["h"], H, tt["ss_emb"], dyn_t, ops.DYN_BS)
        pend, ptap, pdup = 0, None, ()  # K and tap view of the down_proj whose sums wait in part_h
        slots = {}   # tapped layer -> its slots in the tap rows (build_target_layer_ids repeats layers for
        for j, l in enumerate(tl):   # shallow targets: model/utils.py:16-25 concatenates the state twice)
            slots.setdefault(l, []).append(j)

        def spread(dups):   # the other slots of a repeated tap id get the same rows
            for a, b in dups:
                taps[:, :, b * H:(b + 1) * H].copy_(taps[:, :, a * H:(a + 1) * H])

        self._pend_ns = None   # part count of the pending sums when they are expert shares, not K parts
        for i, lw in enumerate(t.layers):
            ops.norm_frag_batch(tt["h"], R, lw["ln1"], t.eps, tt["xn"], dyn_t, ops.DYN_BS,
                                part=tt["part_h"] if pend else None, N=H, K=pend, tap=ptap, nsplit=self._pend_ns)
            spread(pdup)
            if self.attn_impl == "head" and self.qkv_parts:
                ops.gemm_f32_batch(lw["qkv"], s["xn"], R, t.nqkv, H, tt["part_qkv"], dyn_t)
                ops.attn_head_batch_f32(qkv_parts=tt["part_qkv"], nparts=nsp, MT=MT, ld=t.nqkv, q_col=0, k_col=t.q_dim,
                                        v_col=t.q_dim + t.kv_dim, R=RQ, n_q=t.n_q, n_kv=t.n_kv, q_norm_w=lw["q_norm"],
                                        k_norm_w=lw["k_norm"], eps=t.eps, cos_tab=cos, sin_tab=sin, kcache=self.tk,
                                        vcache=self.tv, layer=i, scale=128 ** -0.5, causal=True, dyn=self.dyn_t,
                                        kv_len_max=kvmax, ws=self.hws_t, max_splits=self.max_splits, out_frag=tt["attn"])
            elif self.attn_impl == "head":
                ops.gemm_resid_batch(lw["qkv"], s["xn"], R, t.nqkv, H, tt["xq"], add_residual=False, ws=self.gws,
                                     dyn=dyn_t)
                ops.attn_head_batch(xq=tt["xq"], q_col=0, k_col=t.q_dim, v_col=t.q_dim + t.kv_dim, R=RQ, n_q=t.n_q,
                                    n_kv=t.n_kv, q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"], eps=t.eps, cos_tab=cos,
                                    sin_tab=sin, kcache=self.tk, vcache=self.tv, layer=i, scale=128 ** -0.5, causal=True,
                                    dyn=self.dyn_t, kv_len_max=kvmax, ws=self.hws_t, max_splits=self.max_splits,
                                    out_frag=tt["attn"], q_tiles=TPR)
            else:
                ops.gemm_f32_batch(lw["qkv"], s["xn"], R, t.nqkv, H, tt["part_qkv"], dyn_t)
                ops.attn_fused_batch(qkv=tt["part_qkv"], nsplit=nsp, split_stride=MT * 16 * t.nqkv, ld=t.nqkv, q_col=0,
                                     k_col=t.q_dim, v_col=t.q_dim + t.kv_dim, R=R, n_q=t.n_q, n_kv=t.n_kv,
                                     q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"], eps=t.eps, cos_tab=cos, sin_tab=sin,
                                     kcache=self.tk, vcache=self.tv, layer=i, scale=128 ** -0.5, causal=True,
                                     dyn=dyn_t, kv_len_max=kvmax, ws=self.aws_t,
                                     max_splits=self.max_splits, out_frag=tt["attn"])
            ops.gemm_f32_batch(lw["o"], s["attn"], R, H, t.q_dim, tt["part_h"], dyn_t)
            ops.norm_frag_batch(tt["h"], R, lw["ln2"], t.eps, tt["xn"], dyn_t, ops.DYN_BS, part=tt["part_h"],
                                N=H, K=t.q_dim)
            pns = None
            if "gu_e" in lw:   # sparse-MoE layer (Qwen3MoeSparseMoeBlock): the requests share attention and projections
                pns = self._moe_mlp(lw)   # above; routing and expert weights are per request
                pend = 1
            else:
                ops.gemm_silu_mul_batch(lw["gu"], s["xn"], R, t.I, H, tt["act"], self.gws, dyn_t)
                ops.gemm_f32_batch(lw["down"], s["act"], R, H, t.I, tt["part_h"], dyn_t)
                pend = t.I
            # the layer's output (a tapped layer's hidden rows, model/utils.py:16-25) exists once the
            # next norm launch has added these sums: it writes the tap
            sl = slots.get(i, ())
            ptap = taps[:, :, sl[0] * H:(sl[0] + 1) * H] if sl else None
            pdup = [(sl[0], b) for b in sl[1:]]
            self._pend_ns = pns
        ops.norm_frag_batch(tt["h"], R, t.norm, t.eps, tt["xn"], dyn_t, ops.DYN_BS, part=tt["part_h"], N=H, K=pend,
                            tap=ptap, nsplit=self._pend_ns)
        spread(pdup)
        if self.temperature < 1e-5:
            ops.gemm_argmax_batch(self.lm_wp, s["xn"], R, t.V, H, 0, 16, self.gws, self.post.view(-1, 16), 0, dyn_t,
                                  nrows_dyn_word=ops.DYN_BS)
        else:
            # T > 0 (model/utils.py:30-34): the same GEMM materialises the bf16 logits and the
            # reference's own softmax + torch.multinomial draws the posterior (caller's RNG stream;
            # one draw over all requests' rows, so the stream differs from R sequential runs)
            if self._logits is None:
                self._logits = torch.zeros(MT, 16, t.V, dtype=BF16, device=self.dev)
            ops.gemm_argmax_batch(self.lm_wp, s["xn"], R, t.V, H, 0, 16, self.gws, self.post, 0, dyn_t,
                                  nrows_dyn_word=ops.DYN_BS, logits=self._logits)
            self.post[:R] = sample(self._logits[:R], self.temperature)

    def _moe_mlp(self, lw: dict) -> int:
        """Sparse-MoE MLP of one target layer for the requests of the group: NativeTarget.moe_mlp_tiles."""
        return self.target.moe_mlp_tiles(lw, self.NT, self.MT, self.dyn_tt, self.t["xn"], self.t["part_h"])

    def _accept_launch(self) -> None:
        ops.accept_commit_batch(self.block, self.post, self.R, self.output_ids, self.dyn_d, self.dyn_t, self.stop_t,
                                self.result, rearm_mask_id=self.mask_id, tiles_per_req=self.TPR,
                                dyn_d_tiles=self.dyn_dt, dyn_t_tiles=self.dyn_tt)

    def accept(self, launch: bool = True) -> list:
        """Acceptance scan + commit + rollback bookkeeping of all requests (:258-268) and
        the cycle's one device->host read.  Returns per request (tau, stop) or None."""
        if launch:
            self._accept_launch()
        res = self.result[:self.R].tolist()
        out = []
        for r in range(self.R):
            if not self.live[r]:
                out.append(None)
                continue
            self.start[r] = res[r][1]
            out.append((res[r][0] + 1, bool(res[r][2])))
        return out

    # ------------------------------------------------------------------ hipGraph
    @torch.inference_mode()
    def capture(self) -> None:
        """Capture the cycle's launch sequence (~300 kernels) into three hipGraphs: draft body,
        lm_head + unmask, verify + accept.  Every length the kernels need is read from `dyn`
        on the device, so one capture serves every cycle; the key-split count of the attention
        launches is fixed at the cache capacity (splits past a request's keys are empty).
        The host then spends three graph launches per cycle instead of ~300 ctypes calls."""
        kv = self.max_rows
        self.model._rope_tab(kv + 64)
        self.target._rope_tab(kv + 64)
        torch.cuda.synchronize(self.dev)
        self.graphs = {}
        for name, fn in (("body", lambda: self._draft_body(kv)), ("head", self._draft_head),
                         ("verify", lambda: (self.verify(kv), self._accept_launch()))):
            self.graphs[name] = capture_graph(fn)   # (not torch.cuda.graph(): its empty_cache(), see generate.capture_graph)

    @torch.inference_mode()
    def cycle_graph(self, draft_token_hook: Optional[Callable] = None) -> list:
        """`cycle` through the captured graphs (call `capture()` once after the first eager
        cycle).  Note the capture itself replays nothing: state only advances here."""
        g = self.graphs
        self._mark("draft", 0)
        g["body"].replay()
        self._mark("lm_head", 0)
        g["head"].replay()
        self._mark("lm_head", 1)
        self._mark("draft", 1)
        if draft_token_hook is not None:
            for r in range(self.R):
                if self.live[r]:
                    draft_token_hook(r, self.block[r:r + 1], self.start[r], self.hook_calls[r])
                    self.hook_calls[r] += 1
        self._mark("target", 0)
        g["verify"].replay()
        self._mark("target", 1)
        return self.accept(launch=False)

    @torch.inference_mode()
    def cycle(self, draft_token_hook: Optional[Callable] = None, after_draft: Optional[Callable] = None,
              ahead_ok: bool = False) -> list:
        """One decode cycle of every live request.  draft_token_hook(r, block_row, start,
        call): test/bench instrumentation for scripted acceptance, as in DecodeSession.
        ahead_ok: the caller promises that the NEXT cycle runs with the same block sizes and the same live requests
        (no tail clamp, no request ending on this cycle's result): its draft forward — every length it needs is in the
        device records the accept kernel writes — is then enqueued behind this cycle's accept kernel, before the host
        reads the result, and the GPU does not idle through the host's turnaround."""
        if self._ahead:
            self._ahead = False
            if self.events is not None and self._ahead_ev:
                e = self._ahead_ev
                self.events["draft"], self.events["lm_head"] = [e[0], e[1]], [e[2], e[3]]
        else:
            self._mark("draft", 0)
            self.draft()
            self._mark("draft", 1)
        if after_draft is not None:
            after_draft()
        if draft_token_hook is not None:
            for r in range(self.R):
                if self.live[r]:
                    draft_token_hook(r, self.block[r:r + 1], self.start[r], self.hook_calls[r])
                    self.hook_calls[r] += 1
        self._mark("target", 0)
        self.verify()
        self._mark("target", 1)
        if not (ahead_ok and self.run_ahead):
            return self.accept()
        self._accept_launch()
        ev = None
        if self.events is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        self._draft_body(self._kv_len_max() + self.BW)   # (the new starts are at most a block further)
        if ev:
            ev[2].record()
        self._draft_head()
        if ev:
            ev[3].record()
            ev[1].record()
        self._ahead, self._ahead_ev = True, ev
        return self.accept(launch=False)


@torch.inference_mode()
def dflash_generate_batch(model: DFlashDraftModel, target: NativeTarget, input_ids: Sequence[torch.Tensor],
                          mask_token_id: int, max_new_tokens: int, block_size: int, stop_token_ids,
                          temperature: float = 0.0, draft_token_hook: Optional[Callable] = None,
                          group_size: int = MAX_GROUP, hook_block_view: bool = False) -> list:
    """`dflash_generate` (benchmark.py:44-251) for a list of prompts: requests run in
    groups of `group_size` <= 4 (<= 2 with block sizes of 17..32 rows) that share the weight stream; returns one namespace per
    prompt with the fields of benchmark.py:242-251 (timing fields are the group's).
    draft_token_hook(request_index, block, start, call)."""
    if not 1 <= block_size <= 32:
        raise NotImplementedError("the batched loop takes blocks of 1..16 rows (one tile per request) or 17..32 rows (two)")
    tpr = 1 if block_size <= 16 else 2      # blocks of 17..32 rows: a request takes two of the group's four tiles
    group_size = min(group_size, MAX_GROUP // tpr)
    n = len(input_ids)
    results = [None] * n
    for g0 in range(0, n, group_size):
        idx = list(range(g0, min(n, g0 + group_size)))
        prompts = [input_ids[i] for i in idx]
        pmax = max(p.shape[1] for p in prompts)
        max_len = [p.shape[1] + max_new_tokens for p in prompts]
        dec = BatchedDecoder(model, target, len(idx), max_rows=pmax + max_new_tokens + 3 * 16 * tpr,
                             out_len=pmax + max_new_tokens + 16 * tpr, mask_token_id=mask_token_id,
                             stop_token_ids=stop_token_ids, temperature=temperature, tiles_per_request=tpr)
        t0 = cuda_time()
        for r, p in enumerate(prompts):
            dec.admit(r, p, temperature)
        ttft = cuda_time() - t0
        taus = [[] for _ in idx]
        # (hook_block_view: the hook sees the block as the single-request loop hands it over, bs slots; default: the
        # request's whole 16- / 32-slot row)
        hook = ((lambda r, blk, start, call: draft_token_hook(idx[r], blk[:, :max(1, dec.bs[r])] if hook_block_view else blk,
                                                              start, call)) if draft_token_hook else None)
        t1 = cuda_time()
        clock = [t1]
        first = True
        stop_always = stop_token_ids is not None and mask_token_id in stop_token_ids
        for r in range(len(idx)):   # nothing to generate
            if dec.start[r] >= max_len[r]:
                dec.park(r)
        while any(dec.live):
            for r in range(len(idx)):  # tail clamp (benchmark.py:104-105)
                if dec.live[r]:
                    dec.set_block_size(r, max(1, min(block_size, max_len[r] - dec.start[r])))
            # run-ahead draft: only while no live request can finish or hit its tail clamp on this cycle's result
            ahead = (stop_token_ids is None and not stop_always
                     and all(dec.start[r] + 2 * block_size <= max_len[r] for r in range(len(idx)) if dec.live[r]))
            if first:   # the clock restarts after the first draft, before its verify (benchmark.py:145-147)
                first = False
                out = dec.cycle(hook, after_draft=lambda: clock.__setitem__(0, cuda_time()), ahead_ok=ahead)
                t1 = clock[0]
            else:
                out = dec.cycle(hook, ahead_ok=ahead)
            for r, o in enumerate(out):
                if o is None:
                    continue
                taus[r].append(o[0])
                if o[1] or stop_always or dec.start[r] >= max_len[r]:
                    dec.park(r)
        decode_s = cuda_time() - t1
        for r, i in enumerate(idx):
            ids = _trim(dec.output_ids[r:r + 1], max_len[r], mask_token_id, stop_token_ids, dec.n_in[r])
            n_out = ids.shape[1] - dec.n_in[r]
            results[i] = SimpleNamespace(output_ids=ids.clone(), num_input_tokens=dec.n_in[r], num_output_tokens=n_out,
                                         time_to_first_token=ttft, time_per_output_token=decode_s / max(1, n_out),
                                         acceptance_lengths=taus[r], cycle_trace=[], profile_summary=None)
        del dec
    return results


@torch.inference_mode()
def dflash_generate_policy_batch(*, model: DFlashDraftModel, target: NativeTarget, input_ids: Sequence[torch.Tensor],
                                 mask_token_id: int, max_new_tokens: int, stop_token_ids, temperature: float,
                                 schedulers: Sequence, draft_token_hook: Optional[Callable] = None,
                                 group_size: int = MAX_GROUP) -> list:
    """`dflash_generate_policy` (benchmark_dynamic_schedule.py:260-434) for a list of prompts, the requests of a group
    sharing the weight stream: every request has its OWN scheduler (schedulers[i], e.g. an EWMAPerformanceScheduler) and
    therefore its own block size per cycle — the kernels read each request's size from its length record, so a group
    may mix 8-, 12- and 16-row blocks in one pass over the weights.  Returns one namespace per prompt with the fields of
    :425-434.  The cycle time a scheduler is fed is the GROUP's cycle wall time (what its request actually waited).
    Block sizes <= 16 (one tile per request); T = 0 (the reference samples the draft with T too, :342 — a per-request
    RNG stream over a shared launch has no counterpart in it): both raise NotImplementedError otherwise."""
    if temperature >= 1e-5:
        raise NotImplementedError("the batched policy loop is greedy; run T > 0 requests through dflash_generate_policy")
    n = len(input_ids)
    if len(schedulers) != n:
        raise ValueError("one scheduler per prompt")
    for sc in schedulers:
        if max(sc.candidates) > 16:
            raise NotImplementedError("the batched kernels take blocks of at most 16 rows")
    results = [None] * n
    stop_t = None
    for g0 in range(0, n, group_size):
        idx = list(range(g0, min(n, g0 + group_size)))
        prompts = [input_ids[i] for i in idx]
        scheds = [schedulers[i] for i in idx]
        pmax = max(p.shape[1] for p in prompts)
        max_len = [p.shape[1] + max_new_tokens for p in prompts]
        dec = BatchedDecoder(model, target, len(idx), max_rows=pmax + max_new_tokens + 3 * 16,
                             out_len=pmax + max_new_tokens + 16, mask_token_id=mask_token_id,
                             stop_token_ids=stop_token_ids, temperature=0.0)
        stop_t = dec.stop_t
        t0 = cuda_time()
        for r, p in enumerate(prompts):
            dec.admit(r, p, 0.0)
        ttft = cuda_time() - t0
        R = len(idx)
        taus, used, traces, cyc = [[] for _ in idx], [[] for _ in idx], [[] for _ in idx], [0] * R
        hook = (lambda r, blk, start, call: draft_token_hook(idx[r], blk, start, call)) if draft_token_hook else None
        stop_always = stop_token_ids is not None and mask_token_id in stop_token_ids
        for r in range(R):
            if dec.start[r] >= max_len[r]:
                dec.park(r)
        t1 = cuda_time()
        clock = [t1]
        first = True
        while any(dec.live):
            chosen, bs, lgen, start_idx = [0] * R, [0] * R, [0.0] * R, list(dec.start)
            for r in range(R):
                if dec.live[r]:
                    chosen[r] = scheds[r].select(cyc[r])
                    bs[r] = max(1, min(chosen[r], max_len[r] - dec.start[r]))
                    dec.set_block_size(r, bs[r])
                    lgen[r] = float(bs[r])

            def after_draft(first_now=first):
                # EOS-aware generated length (benchmark_dynamic_schedule.py:344-349)
                if stop_t is not None:
                    for r in range(R):
                        if dec.live[r] and bs[r] > 1:
                            pos = torch.isin(dec.block[r, 1:bs[r]], stop_t).nonzero(as_tuple=True)[0]
                            if pos.numel() > 0:
                                lgen[r] = float(min(int(pos[0].item()) + 1, bs[r]))
                if first_now:   # the TPOT clock restarts after the first draft (benchmark_dynamic_schedule.py:352-354)
                    clock[0] = cuda_time()

            c0 = cuda_time()
            out = dec.cycle(hook, after_draft=after_draft)
            cycle_s = cuda_time() - c0
            if first:
                first, t1 = False, clock[0]
            for r, o in enumerate(out):
                if o is None:
                    continue
                sc = scheds[r]
                tau = o[0]
                taus[r].append(tau)
                used[r].append(bs[r])
                sc.update(tau=tau, cycle_s=cycle_s, effective_bs=bs[r], cycle_idx=cyc[r], l_gen=lgen[r])
                traces[r].append({
                    "cycle_idx": cyc[r], "start_idx": int(start_idx[r]), "block_size": int(bs[r]),
                    "chosen_block_size": int(chosen[r]), "tau": int(tau), "l_gen": float(lgen[r]),
                    "acceptance_ratio": float(tau / max(1, bs[r])), "cycle_s": float(cycle_s),
                    "tau_hat": sc.tau_hat.get(bs[r]), "cycle_hat": sc.cycle_hat.get(bs[r]),
                    "score_hat": sc.score_hat.get(bs[r]), "current_block_size": int(sc.current),
                    "adl_lgen_hat": sc.adl_lgen_hat, "adl_lacc_hat": sc.adl_lacc_hat,
                    "adl_target_k": int(sc.adl_target_k), "adl_target_bs": int(sc.adl_target_bs)})
                cyc[r] += 1
                if o[1] or stop_always or dec.start[r] >= max_len[r]:
                    dec.park(r)
        decode_s = cuda_time() - t1
        for r, i in enumerate(idx):
            ids = _trim(dec.output_ids[r:r + 1], max_len[r], mask_token_id, stop_token_ids, dec.n_in[r])
            n_out = ids.shape[1] - dec.n_in[r]
            results[i] = SimpleNamespace(output_ids=ids.clone(), num_input_tokens=dec.n_in[r], num_output_tokens=n_out,
                                         time_to_first_token=ttft, time_per_output_token=decode_s / max(1, n_out),
                                         acceptance_lengths=taus[r], used_block_sizes=used[r], cycle_trace=traces[r])
        del dec
    return results
