"""`DFlashDraftModel` — host-side mirror of the reference's draft-model operator API
(model/dflash.py:147-277) over the gfx950 kernels.

Same construction (`DFlashDraftModel(config)` + `load_state_dict` with the
reference's key names, or `from_pretrained(dir)`), same `forward(...)` and
`spec_generate(...)` signatures and return values, same attributes
(`block_size`, `mask_token_id`, `target_layer_ids`, `device`).  What differs is
underneath: weights are re-laid once into MFMA-fragment order, the draft KV
cache is preallocated (`DFlashKVCache`, O(1) rollback instead of `torch.cat` /
`crop`), and one cycle's draft forward + lm_head + argmax is ~50 kernel launches
on the caller's stream with all lengths read from device memory.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch

from . import ops
from .config import DFlashConfig

BF16 = torch.bfloat16


def _rope_tables(head_dim: int, theta: float, max_pos: int, device):
    """cos/sin rows for positions 0..max_pos-1, first half of the head only (the two
    halves are equal).  Same arithmetic as Qwen3RotaryEmbedding.forward
    (tf:models/qwen3/modeling_qwen3.py:125-137): fp32 product, fp32 cos/sin, cast."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    pos = torch.arange(max_pos, dtype=torch.float)
    freqs = (inv[None, :, None] @ pos[None, None, :]).transpose(1, 2)[0]  # [max_pos, head_dim/2], fp32 on the host
    return freqs.cos().to(BF16).to(device).contiguous(), freqs.sin().to(BF16).to(device).contiguous()


class DFlashKVCache:
    """Preallocated draft KV cache: post-norm, post-RoPE K and V of the committed
    context rows.  Offers the two methods the reference loop calls
    on its DynamicCache (`get_seq_length`, `crop`, model/dflash.py:241,246); crop is
    a counter update."""

    def __init__(self, cfg: DFlashConfig, max_rows: int, device):
        self.cfg = cfg
        self.max_rows = int(max_rows)
        L, kv = cfg.num_hidden_layers, cfg.num_key_value_heads
        self.k = torch.zeros(L, kv, self.max_rows, cfg.head_dim, dtype=BF16, device=device)
        self.v = torch.zeros_like(self.k)
        self.dyn = torch.zeros(16, dtype=torch.int32, device=device)   # one length record per 16-row block tile
        self.length = 0

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.length

    def crop(self, max_length: int) -> None:
        if 0 < max_length < self.length:  # legacy positive form = absolute length (tf:cache_utils.py:169-180)
            self.length = int(max_length)
        elif max_length < 0:
            self.length = max(0, self.length + int(max_length))



class WideRows:
    """What draft_block returns for a 17..32-row block on the one-pass path: the final-normed rows of both 16-row
    tiles as frag16 (a ragged-batch row source, R = 2); draft_tokens runs ONE lm_head pass over them."""

    def __init__(self, src, dyn, tiles: int = 2):
        self.src, self.dyn, self.tiles = src, dyn, tiles   # dyn: the two tiles' length records [2, 8]

    def __len__(self):
        return self.tiles


class DFlashDraftModel:
    def __init__(self, config, device=None):
        self.config = DFlashConfig.from_any(config)
        c = self.config
        if c.head_dim != 128:
            raise NotImplementedError("the gfx950 kernels are built for head_dim == 128")
        if c.hidden_size % 32 or c.intermediate_size % 32 or c.vocab_size % 16:
            raise NotImplementedError("need hidden/intermediate % 32 == 0 and vocab % 16 == 0")
        # hidden_size > 4096 (a 14B / 32B-class target): the single-request kernels keep a whole K = hidden row slice per
        # workgroup; such models run through the ragged-batch kernels (K cut over workgroups) as a group of one request:
        # dflash_generate / spec_generate route there (generate.py), forward() / draft_block() are not available
        self.wide_hidden = c.hidden_size // 32 > 128
        self.block_size = c.block_size
        self.mask_token_id = c.mask_token_id
        self.target_layer_ids = list(c.target_layer_ids)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dtype = BF16
        self.w: Optional[dict] = None
        self._ws = None
        self._lm_head_cache = {}
        self._rope = None
        # "head": dfl_attn_head on finished bf16 q/k/v rows (round 2); "fused": round-1 stage on fp32 partials
        self.attn_impl = "head"
        self.fuse_oproj = False    # True: attention + o_proj in one launch (dfl_attn_head_oproj); measured slower
        # blocks of 17..32 rows: True = ONE pass over the weights through the ragged-batch GEMMs (R = 2), False = the
        # single-request GEMMs once per 16-row tile (two passes; kept for A/B and as a second implementation for tests)
        self.wide_one_pass = True
        self._wide = None
        self.lm_head_events = None  # (start, end) torch.cuda.Event pair around the next lm_head GEMM launch (bench.py)
        self.lm_head_events_log = None  # a list: every pair actually recorded is appended (a run-ahead draft uses the
        #                                 pair of the cycle in which it is ENQUEUED)
        self.wide_prefill = True   # False: the round-1 context prefill in 16-row groups (kept for A/B timing and tests)
        self.rows_prefill = True   # False: prompts of >= 128 context rows go 64 at a time (round 2) instead of at once

    # ------------------------------------------------------------------ weights
    def eval(self):
        return self

    def load_state_dict(self, sd: dict, strict: bool = True):
        """Takes the reference module's state dict (SURVEY.md §8b key names), moves
        each tensor to the GPU in bf16 and re-lays the matrices for streaming."""
        c = self.config
        want = c.state_dict_shapes()
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want and "rotary_emb" not in k]
        if strict and (missing or unexpected):
            raise KeyError(f"load_state_dict: missing={missing[:4]}... unexpected={unexpected[:4]}...")
        for k, shape in want.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {tuple(shape)}")

        def dev(name):
            return sd[name].to(device=self.device, dtype=BF16).contiguous()

        w = {"fc": ops.pack_weight(dev("fc.weight")), "hidden_norm": dev("hidden_norm.weight"),
             "norm": dev("norm.weight"), "layers": []}
        for i in range(c.num_hidden_layers):
            p = f"layers.{i}."
            qkv = torch.cat([dev(p + "self_attn.q_proj.weight"), dev(p + "self_attn.k_proj.weight"),
                             dev(p + "self_attn.v_proj.weight")], dim=0)
            w["layers"].append({
                "qkv": ops.pack_weight(qkv),
                "o": ops.pack_weight(dev(p + "self_attn.o_proj.weight")),
                "gu": ops.pack_weight_gateup(dev(p + "mlp.gate_proj.weight"), dev(p + "mlp.up_proj.weight")),
                "down": ops.pack_weight(dev(p + "mlp.down_proj.weight")),
                "q_norm": dev(p + "self_attn.q_norm.weight"), "k_norm": dev(p + "self_attn.k_norm.weight"),
                "ln1": dev(p + "input_layernorm.weight"), "ln2": dev(p + "post_attention_layernorm.weight"),
            })
            del qkv
        # context K/V weights of ALL layers as one packed weight (tile-major layout: the k/v column tiles of
        # each layer's packed qkv, concatenated): the context rows' K/V of every layer depend only on the
        # context rows, so one GEMM per cycle produces them all (model/dflash.py:73-78, context half)
        w["kv_all"] = torch.cat([lw["qkv"][c.q_dim * c.hidden_size:] for lw in w["layers"]]).contiguous()
        w["k_norm_all"] = torch.stack([lw["k_norm"] for lw in w["layers"]]).contiguous()
        torch.cuda.synchronize(self.device)
        self.w = w
        self._ws = None  # row sources point at the weights: rebuild with them
        return self

    @classmethod
    def from_pretrained(cls, path: str, device=None, **_):
        """HF checkpoint directory: config.json (+ block_size / num_target_layers /
        dflash_config, model/dflash.py:157,162-163) and *.safetensors shards."""
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        model = cls(cfg, device=device)
        sd = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                sd.update(load_file(os.path.join(path, fn)))
        sd = {k[len("model."):] if k.startswith("model.") and k[6:] in model.config.state_dict_shapes() else k: v
              for k, v in sd.items()}
        return model.load_state_dict(sd)

    # ------------------------------------------------------------------ scratch
    def _workspace(self):
        if self._ws is None:
            c, d = self.config, self.device
            H, I = c.hidden_size, c.intermediate_size
            nqkv = c.q_dim + 2 * c.kv_dim
            self.ks_fc = ops.pick_ksplit(H, c.fc_in, 1)
            self.ks_qkv = ops.pick_ksplit(nqkv, H, 2)
            self.ks_o = ops.pick_ksplit(H, c.q_dim, 1)
            self.ks_down = ops.pick_ksplit(H, I, 1)
            self.ks_kv = ops.pick_ksplit(2 * c.kv_dim, H, 1)
            npart = max(self.ks_fc * 16 * H, self.ks_qkv * 32 * nqkv, self.ks_o * 16 * H, self.ks_down * 16 * H,
                        self.ks_kv * 16 * 2 * c.kv_dim)
            self.max_splits = 32
            NT = 2  # block rows as up to two 16-row tiles (block sizes 17..32)
            self._ws = dict(
                th_frag=torch.zeros(16 * c.fc_in, dtype=BF16, device=d),
                ctx_frag=torch.zeros(16 * H, dtype=BF16, device=d),
                xn_frag=torch.zeros(NT, 16 * H, dtype=BF16, device=d),
                attn_frag=torch.zeros(NT, 16 * c.q_dim, dtype=BF16, device=d),
                act_frag=torch.zeros(NT, 16 * I, dtype=BF16, device=d),
                h=torch.zeros(16 * NT, H, dtype=BF16, device=d),
                ctxh=torch.zeros(16, H, dtype=BF16, device=d),
                ss_emb=torch.zeros(16 * NT, dtype=torch.float32, device=d),
                ss_h=torch.zeros(NT, H, dtype=torch.float32, device=d),       # per tile [H/16 tiles][16 rows]
                ss_ctx=torch.zeros(H, dtype=torch.float32, device=d),
                q_rot=torch.zeros(c.num_attention_heads, 16, 128, dtype=BF16, device=d),
                part=torch.zeros(npart, dtype=torch.float32, device=d),
                attn_ws=ops.attn_fused_ws(c.num_attention_heads, c.num_key_value_heads, self.max_splits, d),
                argmax_ws=ops.argmax_ws(d),
                ids16=torch.zeros(16, dtype=torch.int64, device=d),
                xq=torch.zeros(16 * NT, nqkv, dtype=BF16, device=d),
                xc=torch.zeros(16, c.num_hidden_layers * 2 * c.kv_dim, dtype=BF16, device=d),
                head_ws=ops.attn_head_ws(c.num_attention_heads, self.max_splits, NT, d),
                sync=torch.zeros(ops.ATTN_OPROJ_SYNC_WORDS, dtype=torch.int32, device=d),
            )
            # row sources of the fused pipeline (pointers are fixed for the model's lifetime), one per block tile:
            # the GEMM that consumes a normalised activation applies the RMSNorm itself
            ws, eps, nt = self._ws, c.rms_norm_eps, H // 16
            L = self.w["layers"]
            hs = [ws["h"][16 * t:16 * t + 16] for t in range(NT)]
            self._src = dict(
                ctx=ops.rows_normed(ws["ctxh"], ws["ss_ctx"], nt, self.w["hidden_norm"], eps, ops.DYN_TAU),
                ln1_first=[ops.rows_normed(hs[t], ws["ss_emb"][16 * t:], 1, L[0]["ln1"], eps, ops.DYN_BS) for t in range(NT)],
                ln1=[[ops.rows_normed(hs[t], ws["ss_h"][t], nt, lw["ln1"], eps, ops.DYN_BS) for t in range(NT)] for lw in L],
                ln2=[[ops.rows_normed(hs[t], ws["ss_h"][t], nt, lw["ln2"], eps, ops.DYN_BS) for t in range(NT)] for lw in L],
                final=[ops.rows_normed(hs[t], ws["ss_h"][t], nt, self.w["norm"], eps, ops.DYN_BS) for t in range(NT)],
                attn=[ops.rows_frag(ws["attn_frag"][t]) for t in range(NT)],
                act=[ops.rows_frag(ws["act_frag"][t]) for t in range(NT)],
            )
        return self._ws

    def _rope_tab(self, need: int):
        if self._rope is None or self._rope[0].shape[0] < need:
            n = max(need, 4096)
            n = 1 << (n - 1).bit_length()
            self._rope = _rope_tables(self.config.head_dim, self.config.rope_theta, n, self.device)
        return self._rope

    def new_cache(self, max_rows: int) -> DFlashKVCache:
        return DFlashKVCache(self.config, max_rows, self.device)

    def packed_lm_head(self, lm_head) -> torch.Tensor:
        """The target's lm_head weight [V, H] in streaming layout, packed once per
        weight tensor (keyed on its storage) and kept on the GPU."""
        wt = lm_head.weight if hasattr(lm_head, "weight") else lm_head
        key = (wt.data_ptr(), tuple(wt.shape), wt._version)
        if key not in self._lm_head_cache:
            self._lm_head_cache.clear()
            self._lm_head_cache[key] = ops.pack_weight(wt.detach().to(device=self.device, dtype=BF16).contiguous())
        return self._lm_head_cache[key]

    # ------------------------------------------------------------------ kernels
    def _ctx_rows(self, th_rows: torch.Tensor, n: int, dyn, dyn_word: Optional[int]):
        """taps [n<=16, fc_in] -> fc -> hidden_norm -> ctx_frag (model/dflash.py:177)."""
        c, ws, w = self.config, self._workspace(), self.w
        ops.pack_rows(th_rows, n, ws["th_frag"], dyn if dyn_word is not None else None, dyn_word or 0)
        ops.gemm_f32(w["fc"], ws["th_frag"], None, 1, c.hidden_size, c.fc_in, self.ks_fc, ws["part"])
        ops.norm_pack(norm_w=w["hidden_norm"], frag=ws["ctx_frag"], H=c.hidden_size, eps=c.rms_norm_eps,
                      part=ws["part"], nsplit=self.ks_fc, part_split=16 * c.hidden_size, ldp=c.hidden_size,
                      dyn=dyn if dyn_word is not None else None, dyn_word=dyn_word or 0)

    def _prefill_ws(self):
        if getattr(self, "_pws", None) is None:
            c, d, MT = self.config, self.device, 4
            H, nkv = c.hidden_size, c.num_hidden_layers * 2 * c.kv_dim
            z = lambda *s_, dt=BF16: torch.zeros(*s_, dtype=dt, device=d)  # noqa: E731
            self._pws = dict(taps=z(MT, 16, c.fc_in), ctxh=z(MT, 16, H), xn=z(MT, 16 * H),
                             part=z(ops.batch_ksplit(H) * MT * 16 * nkv, dt=torch.float32),
                             dyn=z(MT, 8, dt=torch.int32),
                             gws=torch.zeros(ops.lib().dfl_gemm_batch_ws_bytes(H, c.fc_in), dtype=torch.uint8, device=d))
            p = self._pws
            p["src_taps"], p["src_xn"] = ops.brows_plain(p["taps"], ops.DYN_TAU), ops.brows_frag(p["xn"])
        return self._pws

    def _prefill_context_wide(self, cache: DFlashKVCache, th: torch.Tensor, pos0: int) -> None:
        """The prompt's context rows 64 at a time (four 16-row tiles of the ragged-batch GEMMs, model/dflash.py:73-85
        for ctx = P rows): per pass ONE stream of fc (168 MB) and of the 5 layers' k/v weights (84 MB) and five launches
        (fc, hidden_norm, k/v GEMM of all layers, K/V append of all layers) — instead of re-streaming both per 16-row
        group with 12 launches each (P = 1024: 4 GB and 80 launches instead of 13 GB and ~700)."""
        c, w, p = self.config, self.w, self._prefill_ws()
        MT, H, L = 4, c.hidden_size, c.num_hidden_layers
        nkv = L * 2 * c.kv_dim
        n, S = th.shape[0], cache.length
        cos, sin = self._rope_tab(pos0 + n + 64)
        nsp = ops.batch_ksplit(H)
        for g0 in range(0, n, 64):
            rows = min(64, n - g0)
            R = (rows + 15) // 16
            rec = [[S + g0 + 16 * r, max(0, min(16, rows - 16 * r)), 0, pos0 + g0 + 16 * r, 0, 0, 0, 0] for r in range(MT)]
            p["dyn"].copy_(torch.tensor(rec, dtype=torch.int32))
            p["taps"].view(MT * 16, c.fc_in)[:rows].copy_(th[g0:g0 + rows])
            ops.gemm_resid_batch(w["fc"], p["src_taps"], R, H, c.fc_in, p["ctxh"], add_residual=False, ws=p["gws"],
                                 dyn=p["dyn"])
            ops.norm_frag_batch(p["ctxh"], R, w["hidden_norm"], c.rms_norm_eps, p["xn"], p["dyn"], ops.DYN_TAU)
            ops.gemm_f32_batch(w["kv_all"], p["src_xn"], R, nkv, H, p["part"], p["dyn"])
            ops.kv_append_batch(kv=p["part"], nsplit=nsp, split_stride=ops.batch_tiles(R) * 16 * nkv, ld=nkv, k_col=0,
                                v_col=c.kv_dim, col_layer_stride=2 * c.kv_dim, n_layers=L, R=R,
                                n_kv=c.num_key_value_heads, k_norm_w=w["k_norm_all"], eps=c.rms_norm_eps, cos_tab=cos,
                                sin_tab=sin, kcache=cache.k, vcache=cache.v, dyn=p["dyn"])
        cache.length = S + n

    def _prefill_context_rows(self, cache: DFlashKVCache, th: torch.Tensor, pos0: int) -> None:
        """A whole prompt's context rows at once on the prefill kernels (csrc/prefill.hip, model/dflash.py:73-85 for
        ctx = P rows): fc as one prompt-length MFMA GEMM (K = 5 H tapped states per row), hidden_norm + pack, ONE k/v GEMM
        for the five layers, then k-norm + RoPE + cache write per layer — the same roundings as the tile-by-tile forms
        (Linear outputs in bf16), 13 launches for any P (P = 1024: 0.5 ms instead of 1.7 ms in 80 launches)."""
        c, w = self.config, self.w
        H, L, n, S = c.hidden_size, c.num_hidden_layers, th.shape[0], cache.length
        nkv = L * 2 * c.kv_dim
        Pp = ops.prefill_rows_padded(n)
        r = getattr(self, "_rws", None)
        if r is None or r["h"].shape[0] < Pp:
            z = lambda *s_: torch.zeros(*s_, dtype=BF16, device=self.device)  # noqa: E731
            r = self._rws = dict(xf=z(Pp * c.fc_in), h=z(Pp, H), xn=z(Pp * H), kv=z(Pp, nkv))
        cos, sin = self._rope_tab(pos0 + n + 64)
        ops.prefill_pack_rows(th, n, c.fc_in, r["xf"])
        ops.prefill_gemm_rows(w["fc"], r["xf"], n, H, c.fc_in, r["h"])
        ops.prefill_norm_pack(r["h"], n, H, w["hidden_norm"], c.rms_norm_eps, r["xn"])
        ops.prefill_gemm_rows(w["kv_all"], r["xn"], n, nkv, H, r["kv"])
        for i, lw in enumerate(w["layers"]):
            ops.prefill_qk_rope(r["kv"], n, 0, i * 2 * c.kv_dim, i * 2 * c.kv_dim + c.kv_dim, 0, c.num_key_value_heads, None,
                                lw["k_norm"], c.rms_norm_eps, cos, sin, pos0, cache.k[i], cache.v[i], S)
        cache.length = S + n

    def prefill_context(self, cache: DFlashKVCache, target_hidden: torch.Tensor, pos0: int) -> None:
        """Append K/V of `target_hidden` rows (context only, no block) to the cache at
        rows/positions cache.length...  Cycle 0 of the reference projects the P prompt rows
        together with the first block (model/dflash.py:73-85); K/V rows are row-independent, so
        doing them first is the same arithmetic.  More than 16 rows go 64 at a time through the
        ragged-batch GEMMs (`_prefill_context_wide`), up to 16 through the single-tile kernels."""
        c, ws, w = self.config, self._workspace(), self.w
        th = target_hidden.reshape(-1, c.fc_in)
        if th.dtype != BF16:
            raise TypeError("target_hidden must be bf16")
        n = th.shape[0]
        S = cache.length
        if S + n > cache.max_rows:
            raise ValueError("draft KV cache too small")
        if (n >= 128 and self.rows_prefill and c.hidden_size % 128 == 0 and c.fc_in % 64 == 0
                and (c.num_hidden_layers * 2 * c.kv_dim) % 128 == 0 and "kv_all" in self.w):
            return self._prefill_context_rows(cache, th.contiguous(), pos0)
        if (n > 16 and self.wide_prefill) or self.wide_hidden:
            return self._prefill_context_wide(cache, th, pos0)
        cos, sin = self._rope_tab(pos0 + n + 64)
        ops.set_dyn(cache.dyn, S, 0, 0, pos0)
        nkv2 = 2 * c.kv_dim
        for g0 in range(0, n, 16):
            rows = min(16, n - g0)
            self._ctx_rows(th[g0:g0 + rows], rows, None, None)
            for i, lw in enumerate(w["layers"]):
                # k,v column tiles only: the packed qkv weight is tile-major, q tiles first
                kv_wp = lw["qkv"][c.q_dim * c.hidden_size:]
                ops.gemm_f32(kv_wp, ws["ctx_frag"], None, 1, nkv2, c.hidden_size, self.ks_kv, ws["part"])
                ops.qknorm_rope_append(qkv=ws["part"], nsplit=self.ks_kv, split_stride=16 * nkv2, ld=nkv2, q_col=-1,
                                       k_col=0, v_col=c.kv_dim, ctx_row0=0, blk_row0=-1,
                                       n_q=c.num_attention_heads, n_kv=c.num_key_value_heads, q_norm_w=lw["q_norm"],
                                       k_norm_w=lw["k_norm"], eps=c.rms_norm_eps, cos_tab=cos, sin_tab=sin,
                                       q_out=None, kcache=cache.k[i], vcache=cache.v[i], dyn=cache.dyn,
                                       ctx_rows_override=rows, row_base=g0)
        cache.length = S + n


    def raise_if_failed(self) -> None:
        """fuse_oproj only: a dfl_attn_head_oproj launch whose o_proj workgroups gave up waiting (2 ms) leaves a flag."""
        if self.fuse_oproj and self._ws is not None and int(self._ws["sync"][ops.ATTN_OPROJ_FAIL_WORD]) != 0:
            self._ws["sync"].zero_()
            raise RuntimeError("dfl_attn_head_oproj: an o_proj workgroup gave up waiting for the attention stage")

    def draft_block(self, cache: DFlashKVCache, *, th_rows: Optional[torch.Tensor], tau: int, bs: int, pos0: int,
                    block_ids: Optional[torch.Tensor] = None, embed: Optional[torch.Tensor] = None,
                    noise: Optional[torch.Tensor] = None, append: bool = True, dyn_ready: bool = False,
                    s_bound: Optional[int] = None) -> list:
        """One draft forward over the block (model/dflash.py:166-190 for ctx <= 16 rows).
        Context rows `th_rows` [tau, fc_in] and the block (token ids + embedding table,
        or a ready `noise` [bs, H]) -> the row sources of the final-normed hidden states, one per
        16-row block tile (the lm_head GEMM applies the final RMSNorm; scratch, valid until the next
        call).  K/V of tau+bs rows are written at cache rows S.. ; `append` advances the host length
        by tau (the block rows are dropped again, as crop(start) does at :246).
        bs <= 32: blocks of 17..32 rows run as two 16-row tiles — one pass over the weights through the
        ragged-batch GEMMs (`_draft_block_wide`, returns a WideRows), or with wide_one_pass = False one launch
        per tile of every single-request GEMM; the attention takes both query tiles either way.
        s_bound (with dyn_ready, bs <= 16): the launches take S / tau / pos0 from the device record alone — the decode
        loop enqueues this forward BEFORE the host knows the previous cycle's acceptance length (generate.py, run-ahead
        draft).  th_rows then holds 16 rows (the valid count is the record's tau), tau / pos0 are upper bounds,
        s_bound bounds S (it sizes the attention's key splits) and the host-side cache length is left to the caller."""
        c, ws, w = self.config, self._workspace(), self.w
        if bs < 1 or bs > 32 or tau < 0 or tau > 16:
            raise ValueError(f"bs={bs} / tau={tau}: the kernels take 1..32 block rows and 0..16 context rows")
        head = self.attn_impl == "head"
        if bs > 16 and not head:
            raise ValueError("blocks of more than 16 rows need attn_impl='head'")
        ahead = s_bound is not None
        if ahead and not (dyn_ready and bs <= 16 and head and tau > 0 and noise is None):
            raise ValueError("s_bound needs dyn_ready, a block of <= 16 rows, context rows and the 'head' attention stage")
        S = int(s_bound) if ahead else cache.length
        if S + tau + bs > cache.max_rows:
            raise ValueError("draft KV cache too small")
        H, I = c.hidden_size, c.intermediate_size
        nqkv = c.q_dim + 2 * c.kv_dim
        cos, sin = self._rope_tab(pos0 + tau + bs + 64)
        src = self._src
        if not (dyn_ready and bs <= 16):
            # (dyn_ready: the caller's last dfl_accept_commit on this record left exactly these words — S, tau, pos0,
            # start — and the block size has not changed: the decode loop's steady state)
            ops.set_dyn2(cache.dyn, S, tau, bs, pos0)
        dyn = cache.dyn[:8]
        tiles = [(t, cache.dyn[8 * t:8 * t + 8]) for t in range((bs + 15) // 16)]
        if tau > 0:
            # fc straight off the tap rows; hidden_norm is applied by each layer's qkv GEMM (:177)
            ops.gemm_resid(w["fc"], ops.rows_plain(th_rows, ops.DYN_TAU), H, c.fc_in, ws["ctxh"], add_residual=False,
                           ss_out=ws["ss_ctx"], dyn=dyn)
        if head and tau > 0:   # K/V Linear outputs of the context rows, all layers, one pass over 84 MB
            ops.gemm_resid(w["kv_all"], src["ctx"], c.num_hidden_layers * 2 * c.kv_dim, H, ws["xc"], add_residual=False,
                           dyn=dyn)
        L = w["layers"]
        if len(tiles) == 2 and self.wide_one_pass and head:
            return self._draft_block_wide(cache, ws, S, tau, bs, pos0, block_ids, embed, noise, append, cos, sin)
        if noise is not None:  # public forward(): the caller embedded the block itself
            ws["h"][:bs].copy_(noise[:bs])
            ws["ss_emb"][:bs].copy_(noise[:bs].float().pow(2).sum(-1))
        else:
            for t, dt in tiles:
                ops.embed_rows(embed, block_ids[16 * t:], ws["h"][16 * t:], H, ws["ss_emb"][16 * t:], dt, ops.DYN_BS)
        hrow = [ws["h"][16 * t:16 * t + 16] for t in range(2)]
        fuse_o = self.fuse_oproj and head and len(tiles) == 1 and c.q_dim <= 4096
        for i, lw in enumerate(L):
            x1 = src["ln1_first"] if i == 0 else src["ln1"][i]
            if head:
                for t, dt in tiles:
                    ops.gemm_resid(lw["qkv"], x1[t], nqkv, H, ws["xq"][16 * t:], add_residual=False, dyn=dt)
                kw = dict(xq=ws["xq"], q_col=0, k_col=c.q_dim, v_col=c.q_dim + c.kv_dim,
                          xc=ws["xc"] if tau > 0 else None, ck_col=i * 2 * c.kv_dim,
                          cv_col=i * 2 * c.kv_dim + c.kv_dim, n_q=c.num_attention_heads,
                          n_kv=c.num_key_value_heads, q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"],
                          eps=c.rms_norm_eps, cos_tab=cos, sin_tab=sin, kcache=cache.k[i], vcache=cache.v[i],
                          scale=c.head_dim ** -0.5, causal=False, S=S, tau=tau, bs=bs, pos0=pos0,
                          ws=ws["head_ws"], max_splits=self.max_splits, dyn=dyn if ahead else None)
                if fuse_o:
                    ops.attn_head_oproj(**kw, attn_frag=ws["attn_frag"][0], wo=lw["o"], H=H, h_io=hrow[0],
                                        ss_out=ws["ss_h"][0], sync=ws["sync"])
                else:
                    ops.attn_head(**kw, out_frag=ws["attn_frag"], q_tiles=len(tiles),
                                  out_tile_stride=ws["attn_frag"].stride(0))
            else:
                ops.gemm_f32(lw["qkv"], src["ctx"], x1[0], 2, nqkv, H, self.ks_qkv, ws["part"], dyn)
                # one launch: q/k-norm + RoPE + KV append + attention + split merge
                ops.attn_fused(qkv=ws["part"], nsplit=self.ks_qkv, split_stride=32 * nqkv, ld=nqkv, q_col=0,
                               k_col=c.q_dim, v_col=c.q_dim + c.kv_dim, ctx_row0=0, blk_row0=16,
                               n_q=c.num_attention_heads, n_kv=c.num_key_value_heads, q_norm_w=lw["q_norm"],
                               k_norm_w=lw["k_norm"], eps=c.rms_norm_eps, cos_tab=cos, sin_tab=sin, kcache=cache.k[i],
                               vcache=cache.v[i], dyn=dyn, scale=c.head_dim ** -0.5, kv_len_max=S + tau + bs,
                               ws=ws["attn_ws"], max_splits=self.max_splits, out_frag=ws["attn_frag"][0])
            for t, dt in tiles:
                if not fuse_o:
                    ops.gemm_resid(lw["o"], src["attn"][t], H, c.q_dim, hrow[t], add_residual=True, ss_out=ws["ss_h"][t],
                                   dyn=dt)
            for t, dt in tiles:
                ops.gemm_silu_mul(lw["gu"], src["ln2"][i][t], I, H, ws["act_frag"][t], dt)
            for t, dt in tiles:
                ops.gemm_resid(lw["down"], src["act"][t], H, I, hrow[t], add_residual=True, ss_out=ws["ss_h"][t],
                               dyn=dt)
        if append and not ahead:
            cache.length = S + tau
        return src["final"][:len(tiles)]

    def _draft_block_wide(self, cache, ws, S, tau, bs, pos0, block_ids, embed, noise, append, cos, sin) -> WideRows:
        """The block rows of a 17..32-row block in ONE pass over the layer weights: the two 16-row tiles go through the
        ragged-batch GEMMs (R = 2: fp32 K-part sums of o_proj / down_proj, residual add + RMSNorm in
        dfl_norm_frag_batch) and ONE attention launch with two query tiles (model/dflash.py:166-190)."""
        c, w, R = self.config, self.w, 2
        H, I, eps = c.hidden_size, c.intermediate_size, c.rms_norm_eps
        nqkv = c.q_dim + 2 * c.kv_dim
        if self._wide is None:
            ks, d = ops.batch_ksplit, self.device
            xn = torch.zeros(2, 16 * H, dtype=BF16, device=d)
            nmax, kmax = max(c.vocab_size, 2 * I, nqkv), max(H, I, c.q_dim)
            self._wide = dict(
                xn=xn, ids=torch.zeros(2, 16, dtype=torch.int64, device=d),
                part_h=torch.zeros(max(ks(c.q_dim), ks(I)) * 2 * 16 * H, dtype=torch.float32, device=d),
                gws=torch.zeros(max(ops.lib().dfl_gemm_batch_ws_bytes(n, k) for n, k in ((nmax, H), (H, kmax))),
                                dtype=torch.uint8, device=d),
                src=dict(xn=ops.brows_frag(xn), attn=ops.brows_frag(ws["attn_frag"]), act=ops.brows_frag(ws["act_frag"])))
        ww = self._wide
        s, gws, part_h, xn = ww["src"], ww["gws"], ww["part_h"], ww["xn"]
        dyn2 = cache.dyn[:16].view(2, 8)
        h3, xq3 = ws["h"].view(2, 16, H), ws["xq"].view(2, 16, nqkv)
        if noise is not None:   # public forward(): the caller embedded the block itself
            ws["h"][:bs].copy_(noise[:bs])
        else:
            ww["ids"].view(-1)[:bs].copy_(block_ids[:bs])
            ops.embed_rows_batch(embed, ww["ids"], R, h3, H, ws["ss_emb"].view(2, 16), dyn2, ops.DYN_BS)
        pend = 0   # K of the GEMM whose fp32 sums wait in part_h (added by the next norm launch)
        for i, lw in enumerate(w["layers"]):
            ops.norm_frag_batch(h3, R, lw["ln1"], eps, xn, dyn2, ops.DYN_BS, part=part_h if pend else None, N=H, K=pend)
            ops.gemm_resid_batch(lw["qkv"], s["xn"], R, nqkv, H, xq3, add_residual=False, ws=gws, dyn=dyn2)
            ops.attn_head(xq=ws["xq"], q_col=0, k_col=c.q_dim, v_col=c.q_dim + c.kv_dim,
                          xc=ws["xc"] if tau > 0 else None, ck_col=i * 2 * c.kv_dim, cv_col=i * 2 * c.kv_dim + c.kv_dim,
                          n_q=c.num_attention_heads, n_kv=c.num_key_value_heads, q_norm_w=lw["q_norm"],
                          k_norm_w=lw["k_norm"], eps=eps, cos_tab=cos, sin_tab=sin, kcache=cache.k[i], vcache=cache.v[i],
                          scale=c.head_dim ** -0.5, causal=False, S=S, tau=tau, bs=bs, pos0=pos0, ws=ws["head_ws"],
                          max_splits=self.max_splits, out_frag=ws["attn_frag"], q_tiles=2,
                          out_tile_stride=ws["attn_frag"].stride(0))
            ops.gemm_f32_batch(lw["o"], s["attn"], R, H, c.q_dim, part_h, dyn2)
            ops.norm_frag_batch(h3, R, lw["ln2"], eps, xn, dyn2, ops.DYN_BS, part=part_h, N=H, K=c.q_dim)
            ops.gemm_silu_mul_batch(lw["gu"], s["xn"], R, I, H, ws["act_frag"], gws, dyn2)
            ops.gemm_f32_batch(lw["down"], s["act"], R, H, I, part_h, dyn2)
            pend = I
        # last down_proj's sums -> h (the rows forward() returns after its own norm), final norm -> frag16
        ops.norm_frag_batch(h3, R, w["norm"], eps, xn, dyn2, ops.DYN_BS, part=part_h, N=H, K=pend)
        if append:
            cache.length = S + tau
        return WideRows(s["xn"], dyn2)

    def draft_tokens(self, hid_frag, lm_head_wp: torch.Tensor, bs: int, block_ids: torch.Tensor,
                     logits: Optional[torch.Tensor] = None, margins: Optional[torch.Tensor] = None) -> None:
        """block_ids[1:bs] <- argmax(lm_head(hidden[1:bs])) (model/dflash.py:238,245,247).
        hid_frag: what draft_block returned (one row source per 16-row tile; a single source = one tile).
        logits (bf16 [16 * tiles, V]) / margins (fp32 [>= bs]): margins[j] <- top-1 minus top-2 draft
        logit of block slot j >= 1, the reference's per-position confidence
        (benchmark_candidate_solutions.py:296-302)."""
        c, ws = self.config, self._workspace()
        if isinstance(hid_frag, WideRows):   # both tiles in one lm_head pass (ragged-batch GEMM, R = 2)
            if margins is not None:
                raise NotImplementedError("top-2 margins are computed for blocks of <= 16 rows")
            ww = self._wide
            ops.gemm_argmax_batch(lm_head_wp, hid_frag.src, 2, c.vocab_size, c.hidden_size, 0, 16, ww["gws"], ww["ids"], 0,
                                  hid_frag.dyn, nrows_dyn_word=ops.DYN_BS,
                                  logits=None if logits is None else logits.view(2, 16, c.vocab_size))
            block_ids[1:bs].copy_(ww["ids"].view(-1)[1:bs])
            return
        srcs = hid_frag if isinstance(hid_frag, (list, tuple)) else [hid_frag]
        for t, x in enumerate(srcs):
            row0 = 1 if t == 0 else 0
            nrows = min(bs - 16 * t, 16) - row0
            if nrows <= 0:
                continue
            ev = self.lm_head_events if t == 0 else None   # bench.py: the lm_head kernel itself between two events
            if ev is not None and self.lm_head_events_log is not None:
                self.lm_head_events_log.append(ev)
            ops.gemm_argmax(lm_head_wp, x, c.vocab_size, c.hidden_size, row0, nrows, ws["argmax_ws"], block_ids,
                            16 * t + row0, logits=None if logits is None else logits[16 * t:16 * t + 16],
                            margins=margins, events=ev)

    # ------------------------------------------------------------------ reference API
    @torch.inference_mode()
    def forward(self, position_ids: torch.LongTensor, attention_mask=None, noise_embedding=None, target_hidden=None,
                past_key_values: Optional[DFlashKVCache] = None, use_cache: bool = False, **kwargs) -> torch.Tensor:
        """model/dflash.py:166-190: returns final-normed hidden states [1, q_len, H]
        (not logits).  `past_key_values` is a `DFlashKVCache` (`model.new_cache(n)`);
        as in the reference the K/V of context AND block rows are appended and the
        caller crops the block rows away.  `position_ids` must be the contiguous
        range the reference passes (`arange(cache_len, start + q_len)`)."""
        if self.w is None:
            raise RuntimeError("weights not loaded")
        if attention_mask is not None:
            raise NotImplementedError("the draft attends without a mask (is_causal=False, attention_mask=None)")
        c = self.config
        if noise_embedding.shape[0] != 1:
            raise NotImplementedError("batch = 1 by construction (SURVEY.md §1)")
        q_len, ctx = noise_embedding.shape[1], target_hidden.shape[1]
        pos0 = int(position_ids[0, 0])
        if position_ids.shape[1] != ctx + q_len:
            raise ValueError("position_ids must cover context + block rows")
        cache = past_key_values if past_key_values is not None else self.new_cache(ctx + q_len)
        if past_key_values is None:
            cache.length = 0
        th = target_hidden[0].to(BF16)
        head = max(0, ctx - 16) if ctx > 16 else 0
        if head:
            self.prefill_context(cache, th[:head], pos0)
        tau = ctx - head
        frag = self.draft_block(cache, th_rows=th[head:].contiguous() if tau else None, tau=tau, bs=q_len,
                                pos0=pos0 + head, noise=noise_embedding[0].to(BF16).contiguous())
        cache.length += q_len  # the reference's cache holds the block rows until crop()
        H, ws = c.hidden_size, self._workspace()
        out = []
        for t in range(len(frag)):
            ops.norm_pack(norm_w=self.w["norm"], frag=ws["xn_frag"][t], H=H, eps=c.rms_norm_eps,
                          resid_in=ws["h"][16 * t:16 * t + 16], dyn=cache.dyn[8 * t:8 * t + 8], dyn_word=ops.DYN_BS)
            out.append(ws["xn_frag"][t].view(H // 8, 16, 8).permute(1, 0, 2).reshape(16, H))
        return torch.cat(out)[:q_len].unsqueeze(0).clone()

    __call__ = forward

    @torch.inference_mode()
    def spec_generate(self, target, input_ids: torch.LongTensor, max_new_tokens: int, stop_token_ids,
                      temperature: float, draft_token_hook=None) -> torch.LongTensor:
        """model/dflash.py:192-277."""
        from .generate import _generate_wide_hidden, run_decode
        if self.wide_hidden:    # (ids are the same with or without the harness form's tail clamp)
            return _generate_wide_hidden(self, target, input_ids, self.mask_token_id, max_new_tokens, self.block_size,
                                         stop_token_ids, temperature, draft_token_hook=draft_token_hook).output_ids
        r = run_decode(self, target, input_ids, mask_token_id=self.mask_token_id, max_new_tokens=max_new_tokens,
                       block_size=self.block_size, stop_token_ids=stop_token_ids, temperature=temperature,
                       clamp_tail=False, draft_token_hook=draft_token_hook)
        return r.output_ids
