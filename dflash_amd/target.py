"""`NativeTarget` — the caller-side neighbour of the hot path: the
target's 16-token *verify* forward (model/dflash.py:249-255) on the same gfx950
kernels as the draft, over a preallocated target KV cache.

It wraps the caller's HF-style dense causal LM (Qwen3 / Llama layout) and stays
call-compatible with it: `target(input_ids, position_ids=..., past_key_values=...,
output_hidden_states=...)`, `.model.embed_tokens`, `.lm_head`, `.device` all forward to
the wrapped model, so the reference's loop still runs unchanged on it.  The decode
loops of this package detect it and take the fast path instead:

* prefill (M = prompt length, a plain library GEMM problem) runs through the wrapped
  model once; its K/V are copied into the preallocated cache;
* every verify = 36 x {qkv GEMM, q/k-norm + RoPE + append, causal block attention,
  o_proj, residual + norm, gate/up SiLU GEMM, down GEMM, residual + norm [+ tap copy]}
  + lm_head GEMM with fused argmax over all 16 rows: the posterior ids come back
  without materialising 16 x V logits, rollback is a counter, and only the tapped
  layers' hidden rows are kept (the reference keeps all 37, model/utils.py:16-25).
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional, Sequence

import torch

from . import ops
from .model import _rope_tables
from .utils import sample

BF16 = torch.bfloat16


class TargetKVCache:
    """Preallocated target KV: [layer][kv head][rows][128] bf16; crop = counter."""

    def __init__(self, n_layers, n_kv, max_rows, device):
        self.max_rows = int(max_rows)
        self.k = torch.zeros(n_layers, n_kv, self.max_rows, 128, dtype=BF16, device=device)
        self.v = torch.zeros_like(self.k)
        self.dyn = torch.zeros(16, dtype=torch.int32, device=device)   # one length record per 16-row block tile
        self.length = 0

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.length

    def crop(self, max_length: int) -> None:
        if 0 < max_length < self.length:
            self.length = int(max_length)
        elif max_length < 0:
            self.length = max(0, self.length + int(max_length))


class _HiddenStates:
    """hidden_states of a native prefill: indexable like the HF tuple (index 0 = embedding output, l + 1 = output of
    decoder layer l, model/utils.py:16-25) for the states that were kept."""

    def __init__(self, kept: dict, n: int):
        self._kept, self._n = kept, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if i < 0:
            i += self._n
        if i not in self._kept:
            raise KeyError(f"hidden state {i} was not kept by this prefill (tap_layers = {sorted(k - 1 for k in self._kept)})")
        return self._kept[i]


class _Weight:
    """`.weight` holder standing in for the wrapped model's embed_tokens / lm_head once it is gone (keep_hf = False)."""

    def __init__(self, w):
        self.weight = w


class NativeTarget:
    def __init__(self, hf_model, max_splits: int = 32, attn_impl: str = "head", keep_hf: bool = True,
                 prefill: str = "native"):
        """attn_impl: "head" = dfl_attn_head on finished bf16 q/k/v rows (round 2, default); "fused" = the
        round-1 stage (fp32 K-split partials -> dfl_attn_fused), kept for A/B measurement and as a second
        implementation the tests compare against.
        prefill: "native" (round 3, default where the shapes allow: dense target, projection widths % 128) = the prompt
        through the MFMA GEMMs of csrc/prefill.hip on the same packed weights; "hf" = through the wrapped model.
        keep_hf = False: the wrapped model is dropped after packing (ONE copy of the layers' weights in memory; the
        caller must drop its own reference too); needs the native prefill."""
        if attn_impl not in ("head", "fused"):
            raise ValueError("attn_impl must be 'head' or 'fused'")
        if prefill not in ("native", "hf"):
            raise ValueError("prefill must be 'native' or 'hf'")
        self.attn_impl = attn_impl
        # True: attention + o_proj in one launch (dfl_attn_head_oproj) where its range allows.  Measured SLOWER than the
        # two launches (23.2 vs 20.3 us per layer): off by default, kept for A/B and tests
        self.fuse_oproj = False
        # blocks of 17..32 rows: True = one pass over the weights through the ragged-batch GEMMs (R = 2), False = the
        # single-request GEMMs once per 16-row tile (two passes; kept for A/B and as a second implementation for tests)
        self.wide_one_pass = True
        self._wide = None
        self.moe_pair_kernel = True   # MoE gate/up at K <= 2048 through dfl_moe_gate_up (False: the general kernel)
        self.moe_shared_pass = True   # three or four tiles (ragged batch, candidates): one pass over the experts they share
        self.moe_shared_min = 2       # tiles from which the shared pass is taken (30B-A3B layer: 2 tiles 291 -> 216 us, 4: 576 -> 241)
        self._moe_sh = None
        cfg = hf_model.config
        self.check_config(cfg)   # (before anything touches the device: a rejected architecture never launches a kernel)
        self.hf = hf_model
        self.model = hf_model.model
        self.lm_head = hf_model.lm_head
        self.config = cfg
        dev = hf_model.lm_head.weight.device
        if dev.type != "cuda":
            raise RuntimeError("NativeTarget: the wrapped model must live on the GPU")
        self._dev = dev
        self.H = cfg.hidden_size
        self.L = cfg.num_hidden_layers
        self.n_q = cfg.num_attention_heads
        self.n_kv = getattr(cfg, "num_key_value_heads", self.n_q)
        self.hd = getattr(cfg, "head_dim", None) or self.H // self.n_q
        self.I = cfg.intermediate_size
        self.V = cfg.vocab_size
        self.eps = float(cfg.rms_norm_eps)
        rope = getattr(cfg, "rope_parameters", None) or {}
        rtype = rope.get("rope_type", "default") if isinstance(rope, dict) else "default"
        theta = (rope.get("rope_theta") if isinstance(rope, dict) else None) or getattr(cfg, "rope_theta", None)
        if self.hd != 128 or self.H % 32 or (self.I % 32 and not getattr(cfg, "num_experts", 0)) or self.V % 16:
            raise NotImplementedError("NativeTarget: needs head_dim 128, hidden %32, vocab %16")
        # hidden > 4096: prefill on the prefill kernels and decode through the ragged-batch kernels (BatchedDecoder, a
        # group of one request); verify() — the single-request kernels, a whole K = hidden slice per workgroup — is not available
        self.wide_hidden = self.H // 32 > 128
        # position-only RoPE variants: the cos/sin tables are taken from the wrapped model's own
        # rotary module (Llama-3.1's "llama3" frequency scaling, BASELINE config 4; linear; yarn)
        if rtype not in ("default", "llama3", "linear", "yarn") or (rtype == "default" and theta is None):
            raise NotImplementedError(f"NativeTarget: rope_type {rtype!r} is not supported; keep the HF target")
        self.rope_type = rtype
        # sparse-MoE targets (BASELINE configs[4], Qwen3-Coder-30B-A3B): HF Qwen3-MoE layout (fused per-expert
        # gate_up_proj / down_proj tensors, softmax -> top-k -> renormalise router)
        self.E = int(getattr(cfg, "num_experts", 0) or 0)
        if getattr(cfg, "num_local_experts", 0) and not self.E:
            raise NotImplementedError("NativeTarget: only the Qwen3-MoE expert layout is supported; keep the HF target")
        if self.E:
            self.top_k = int(cfg.num_experts_per_tok)
            self.Ie = int(cfg.moe_intermediate_size)
            self.norm_topk = bool(getattr(cfg, "norm_topk_prob", False))
            if self.E > 256 or self.top_k > 8 or self.Ie % 32:
                raise NotImplementedError("NativeTarget: MoE needs <= 256 experts, top-k <= 8, moe_intermediate_size % 32 == 0")
        if getattr(cfg, "attention_bias", False) or getattr(cfg, "mlp_bias", False):
            raise NotImplementedError("NativeTarget: biased projections are not supported")
        self.theta = float(theta) if theta is not None else None
        sd = hf_model.state_dict()

        def w(name):
            return sd[name].detach().to(device=dev, dtype=BF16).contiguous()

        self.layers = []
        for i in range(self.L):
            p = f"model.layers.{i}."
            qkv = torch.cat([w(p + "self_attn.q_proj.weight"), w(p + "self_attn.k_proj.weight"),
                             w(p + "self_attn.v_proj.weight")], dim=0)
            has_qk = (p + "self_attn.q_norm.weight") in sd
            lw = {"qkv": ops.pack_weight(qkv), "o": ops.pack_weight(w(p + "self_attn.o_proj.weight")),
                  "q_norm": w(p + "self_attn.q_norm.weight") if has_qk else None,
                  "k_norm": w(p + "self_attn.k_norm.weight") if has_qk else None,
                  "ln1": w(p + "input_layernorm.weight"), "ln2": w(p + "post_attention_layernorm.weight")}
            if (p + "mlp.experts.0.gate_proj.weight") in sd:   # per-expert ModuleList layout (older transformers)
                raise NotImplementedError("NativeTarget: per-expert ModuleList MoE weights (mlp.experts.{e}.gate_proj) are "
                                          "not supported, only the fused gate_up_proj / down_proj tensors; keep the HF target")
            if (p + "mlp.experts.gate_up_proj") in sd:       # sparse-MoE layer
                gup, dwn = w(p + "mlp.experts.gate_up_proj"), w(p + "mlp.experts.down_proj")   # [E, 2I, H], [E, H, I]
                Ie = self.Ie
                if gup.shape != (self.E, 2 * Ie, self.H) or dwn.shape != (self.E, self.H, Ie):
                    raise ValueError(f"layer {i}: unexpected expert tensor shapes {tuple(gup.shape)} {tuple(dwn.shape)}")
                lw["gu_e"] = torch.stack([ops.pack_weight_gateup(gup[e, :Ie].contiguous(), gup[e, Ie:].contiguous())
                                          for e in range(self.E)])
                lw["down_e"] = torch.stack([ops.pack_weight(dwn[e].contiguous()) for e in range(self.E)])
                ep = (self.E + 15) // 16 * 16               # router rows padded to whole column tiles (never routed to)
                rw = torch.zeros(ep, self.H, dtype=BF16, device=dev)
                rw[:self.E] = w(p + "mlp.gate.weight")
                lw["router"] = ops.pack_weight(rw)
                if ep % 128:    # the prefill's router GEMM (prefill.hip:k_pgemm) takes column blocks of 128
                    rp = torch.zeros((self.E + 127) // 128 * 128, self.H, dtype=BF16, device=dev)
                    rp[:self.E] = rw[:self.E]
                    lw["router_p"] = ops.pack_weight(rp)
                    del rp
                else:
                    lw["router_p"] = lw["router"]
                del gup, dwn, rw
            else:
                lw["gu"] = ops.pack_weight_gateup(w(p + "mlp.gate_proj.weight"), w(p + "mlp.up_proj.weight"))
                lw["down"] = ops.pack_weight(w(p + "mlp.down_proj.weight"))
            self.layers.append(lw)
            del qkv
        self.norm = w("model.norm.weight")
        self.embed = w("model.embed_tokens.weight")
        self.lm_wp = None  # set by share_lm_head() or packed on first use
        self.max_splits = max_splits
        self.q_dim, self.kv_dim = self.n_q * 128, self.n_kv * 128
        self.nqkv = self.q_dim + 2 * self.kv_dim
        self.ks_qkv = ops.pick_ksplit(self.nqkv, self.H, 1)
        self.ks_o = ops.pick_ksplit(self.H, self.q_dim, 1)
        self.ks_down = ops.pick_ksplit(self.H, self.I, 1)
        npart = max(self.ks_qkv * 16 * self.nqkv, self.ks_o * 16 * self.H, self.ks_down * 16 * self.H)
        z = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        NT = 2  # block rows as up to two 16-row tiles (blocks of 17..32 rows: one GEMM launch per tile)
        self.ws = dict(attn=z(NT, 16 * self.q_dim), act=z(NT, 16 * self.I), h=z(16 * NT, self.H),
                       ss_emb=z(16 * NT, dt=torch.float32), ss_h=z(NT, self.H, dt=torch.float32),
                       q=z(self.n_q, 16, 128), part=z(npart, dt=torch.float32),
                       attn_ws=ops.attn_fused_ws(self.n_q, self.n_kv, max_splits, dev), argmax_ws=ops.argmax_ws(dev),
                       xq=z(16 * NT, self.nqkv), head_ws=ops.attn_head_ws(self.n_q, max_splits, NT, dev),
                       sync=torch.zeros(ops.ATTN_OPROJ_SYNC_WORDS, dtype=torch.int32, device=dev),
                       post=torch.zeros(16 * NT, dtype=torch.int64, device=dev))
        self.is_moe = any("gu_e" in lw for lw in self.layers)
        if self.is_moe:
            ep = (self.E + 15) // 16 * 16
            # shares of the active experts (grid.y of dfl_moe_down): 4 with its two-tiles-per-workgroup form (the default)
            self.moe_nsplit = 2 if (os.environ.get("DFL_MOE_DOWN_CT") == "1" or self.H % 32) else 4
            self.ws.update(xn=z(NT, 16 * self.H), xn1=z(NT, 16 * self.H), rlog=z(NT, 16, ep), wt=z(NT, 16, self.E),
                           act_e=z(self.E, 16 * self.Ie), moe_part=z(self.moe_nsplit, 16, self.H, dt=torch.float32),
                           active=torch.zeros(self.E, dtype=torch.int32, device=dev),
                           elist=torch.zeros(self.E, dtype=torch.int32, device=dev),
                           n_active=torch.zeros(1, dtype=torch.int32, device=dev),
                           rticket=torch.zeros(1, dtype=torch.int32, device=dev))
            # norm + gate Linear + routing as ONE launch (dfl_moe_router; DFL_MOE_ROUTER=split: the three launches)
            self.moe_router_fused = (os.environ.get("DFL_MOE_ROUTER", "fused") != "split" and self.H <= 4096 and self.E % 2 == 0)
        ws, nt = self.ws, self.H // 16
        hs = [ws["h"][16 * t:16 * t + 16] for t in range(NT)]
        # row sources, one per tile: the consuming GEMM applies the RMSNorm itself (no norm launches)
        self.src = dict(
            ln1=[[ops.rows_normed(hs[t], ws["ss_emb"][16 * t:] if i == 0 else ws["ss_h"][t], 1 if i == 0 else nt,
                                  lw["ln1"], self.eps, ops.DYN_BS) for t in range(NT)]
                 for i, lw in enumerate(self.layers)],
            ln2=[[ops.rows_normed(hs[t], ws["ss_h"][t], nt, lw["ln2"], self.eps, ops.DYN_BS) for t in range(NT)]
                 for lw in self.layers],
            final=[ops.rows_normed(hs[t], ws["ss_h"][t], nt, self.norm, self.eps, ops.DYN_BS) for t in range(NT)],
            attn=[ops.rows_frag(ws["attn"][t]) for t in range(NT)], act=[ops.rows_frag(ws["act"][t]) for t in range(NT)])
        if self.is_moe:   # MoE layers hand their successor ready-normalised rows (the residual add + norm launch after the experts)
            self.src["xn"] = [ops.rows_frag(ws["xn"][t]) for t in range(NT)]
            self.src["xn1"] = [ops.rows_frag(ws["xn1"][t]) for t in range(NT)]
        q_dim, kv_dim = self.n_q * 128, self.n_kv * 128
        self._pf = self._pf_moe = None
        dense_ok = all("gu_e" in lw for lw in self.layers) or self.I % 64 == 0   # (a dense MLP layer needs I % 64)
        self.native_prefill = (prefill == "native" and (q_dim + 2 * kv_dim) % 128 == 0 and self.H % 128 == 0 and dense_ok
                               and q_dim % 64 == 0 and (not self.is_moe or self.Ie % 64 == 0))
        self.prefill_attn = "native"   # "sdpa": the causal attention core of the prefill through torch (round-3 first form)
        self._rotary = getattr(hf_model.model, "rotary_emb", None)
        if not keep_hf:
            if not self.native_prefill:
                raise NotImplementedError("NativeTarget(keep_hf=False) needs the native prefill (widths % 128, FFN widths % 64)")
            if self.lm_wp is None:
                self.lm_wp = ops.pack_weight(self.lm_head.weight.detach().to(BF16).contiguous())
            self.hf = None
            self.model = SimpleNamespace(embed_tokens=_Weight(self.embed), rotary_emb=self._rotary)
            self.lm_head = _Weight(self.lm_head.weight.detach())
        self.debug_routing = None
        self.gu_events = None   # (layer, start, end): torch.cuda.Event pair recorded around that layer's gate/up GEMM launch
        # (layer, start, end, n_active_out): the same around that MoE layer's expert gate/up launch; n_active_out (int32 [1],
        # device) receives the launch's active-expert count right behind the end event (bench.py: bytes = count x expert bytes)
        self.moe_events = None
        self._rope = None
        self._taps = {}
        torch.cuda.synchronize(dev)

    @staticmethod
    def check_config(cfg) -> None:
        """Architectures the kernels do not cover are rejected loudly, before any launch (the reference is shape-agnostic:
        model/dflash.py:36,42-56 take head_dim, attention_bias and sliding_window from the config; no BASELINE config
        needs them): head_dim != 128, biased projections, sliding-window attention layers."""
        n_q = cfg.num_attention_heads
        hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // n_q
        if hd != 128:
            raise NotImplementedError(f"NativeTarget: the gfx950 kernels are built for head_dim == 128 (got {hd}); keep the HF target")
        if getattr(cfg, "attention_bias", False) or getattr(cfg, "mlp_bias", False):
            raise NotImplementedError("NativeTarget: biased projections are not supported; keep the HF target")
        lt = getattr(cfg, "layer_types", None) or ()
        if any(t == "sliding_attention" for t in lt) and getattr(cfg, "sliding_window", None):
            raise NotImplementedError("NativeTarget: sliding-window attention layers are not supported; keep the HF target")

    # ---- HF-compatible surface (the reference loop can still drive the wrapped model)
    @property
    def device(self):
        return self._dev

    def __call__(self, *a, **kw):
        if self.hf is None:
            raise RuntimeError("NativeTarget(keep_hf=False): the wrapped model is gone; use prefill() / verify()")
        return self.hf(*a, **kw)

    def share_lm_head(self, packed: torch.Tensor) -> None:
        self.lm_wp = packed

    def new_cache(self, max_rows: Optional[int] = None) -> TargetKVCache:
        if max_rows is None:
            raise ValueError("NativeTarget.new_cache needs max_rows (prompt + new tokens + block)")
        return TargetKVCache(self.L, self.n_kv, max_rows, self._dev)

    def _rope_tab(self, need: int):
        if self._rope is None or self._rope[0].shape[0] < need:
            n = 1 << (max(need, 4096) - 1).bit_length()
            if self.rope_type == "default":
                self._rope = _rope_tables(128, self.theta, n, self._dev)
            else:  # what the wrapped model itself multiplies by (scaled frequencies, attention factor), in bf16
                pos = torch.arange(n, device=self._dev).unsqueeze(0)
                cos, sin = self._rotary(torch.zeros(1, dtype=BF16, device=self._dev), pos)
                self._rope = (cos[0, :, :64].to(BF16).contiguous(), sin[0, :, :64].to(BF16).contiguous())
        return self._rope

    # ---- prefill (model/dflash.py:218-225)
    @torch.inference_mode()
    def prefill(self, input_ids: torch.Tensor, cache: TargetKVCache, output_hidden_states: bool = True,
                tap_layers: Optional[Sequence[int]] = None):
        """Returns an object with `.logits` [1, 1, V] (last prompt row, logits_to_keep = 1) and `.hidden_states`
        (indexable like HF's tuple).  tap_layers: keep only hidden_states[l + 1] for these layers (the draft's taps,
        model/utils.py:16-25) instead of all of them."""
        P = input_ids.shape[1]
        if P > cache.max_rows:
            raise ValueError("target KV cache too small for the prompt")
        if self.native_prefill:
            return self._prefill_native(input_ids, cache, output_hidden_states, tap_layers)
        return self._prefill_hf(input_ids, cache, output_hidden_states)

    def _prefill_native(self, input_ids, cache, output_hidden_states, tap_layers):
        """The prompt rows on the kernels: per layer RMSNorm -> frag16 tiles, q/k/v GEMM (bf16 rows), q/k-norm + RoPE +
        cache write, causal attention over the prompt (k_pattn: one wave per head and 16-row query tile), o_proj +
        residual, RMSNorm, gate/up + SiLU,
        down_proj + residual (+ tap).  Last row: final norm + lm_head through the decode path's skinny GEMM."""
        import torch.nn.functional as F
        P = input_ids.shape[1]
        H, I, dev = self.H, self.I, self._dev
        Pp = ops.prefill_rows_padded(P)
        q_dim, kv_dim, nqkv = self.q_dim, self.kv_dim, self.nqkv
        if self._pf is None or self._pf["h"].shape[0] < Pp:
            z = lambda *s: torch.zeros(*s, dtype=BF16, device=dev)  # noqa: E731
            dense_I = I if any("gu" in lw for lw in self.layers) else 0
            self._pf = dict(h=z(Pp, H), xf=z(Pp * max(H, q_dim)), qkv=z(Pp, nqkv), act=z(Pp * max(dense_I, 1)), attn=z(Pp, q_dim),
                            logits=z(16, self.V), ids=torch.zeros(16, dtype=torch.int64, device=dev))
        pf = self._pf
        h, xf, qkv, act, attn = pf["h"], pf["xf"], pf["qkv"], pf["act"], pf["attn"]
        h.zero_()
        torch.index_select(self.embed, 0, input_ids[0], out=h[:P])
        cos, sin = self._rope_tab(P + 64)
        want = None
        if output_hidden_states:
            want = set(range(self.L + 1)) if tap_layers is None else {int(l) + 1 for l in tap_layers}
            if max(want) >= self.L and tap_layers is not None:
                raise NotImplementedError("tapping the last layer (post-norm state) is not supported")
        kept = {}
        if want and 0 in want:
            kept[0] = h[:P].clone().unsqueeze(0)
        for i, lw in enumerate(self.layers):
            ops.prefill_norm_pack(h, P, H, lw["ln1"], self.eps, xf)
            ops.prefill_gemm_rows(lw["qkv"], xf, P, nqkv, H, qkv)
            ops.prefill_qk_rope(qkv, P, 0, q_dim, q_dim + kv_dim, self.n_q, self.n_kv, lw["q_norm"], lw["k_norm"], self.eps,
                                cos, sin, 0, cache.k[i], cache.v[i], 0)
            if self.prefill_attn == "native":   # csrc/prefill.hip:k_pattn, straight into o_proj's frag16 operand
                ops.prefill_attn(qkv, P, 0, cache.k[i], cache.v[i], self.n_q, self.n_kv, 128 ** -0.5, xf)
            else:                               # torch SDPA on the same rows (second implementation: tests, A/B)
                q = qkv[:P, :q_dim].view(P, self.n_q, 128).transpose(0, 1).unsqueeze(0)
                o = F.scaled_dot_product_attention(q, cache.k[i][:, :P].unsqueeze(0), cache.v[i][:, :P].unsqueeze(0),
                                                   is_causal=True, scale=128 ** -0.5, enable_gqa=self.n_kv != self.n_q)
                attn[:P].view(P, self.n_q, 128).copy_(o[0].transpose(0, 1))
                ops.prefill_norm_pack(attn, P, q_dim, None, self.eps, xf)      # rows -> frag16 tiles, no norm
            ops.prefill_gemm_resid(lw["o"], xf, P, H, q_dim, h)
            ops.prefill_norm_pack(h, P, H, lw["ln2"], self.eps, xf)
            tap = None
            if want and (i + 1) in want and i + 1 < self.L:   # (the last layer's output only exists final-normed in HF)
                tap = torch.empty(P, H, dtype=BF16, device=dev)
                kept[i + 1] = tap.unsqueeze(0)
            if "gu_e" in lw:   # sparse-MoE layer: rows sorted by expert, grouped MFMA GEMMs over each expert's rows
                if self._pf_moe is None or self._pf_moe["P"] < P:
                    self._pf_moe = ops.prefill_moe_scratch(P, H, self.Ie, self.E, self.top_k,
                                                           (self.E + 127) // 128 * 128, dev)
                ops.prefill_moe_mlp(lw["router_p"], lw["gu_e"], lw["down_e"], xf, P, H, self.Ie, self.E, self.top_k,
                                    self.norm_topk, h, self._pf_moe, tap=tap)
                if self.debug_routing is not None:    # tests: the experts every prompt row was routed to
                    self.debug_routing.append((i, self._pf_moe["pair_e"][:P, :self.top_k].clone()))
                continue
            ops.prefill_gemm_silu(lw["gu"], xf, P, I, H, act)
            ops.prefill_gemm_resid(lw["down"], act, P, H, I, h, tap=tap)
        # last prompt row: final norm + lm_head on its 16-row tile (the decode path's GEMM, norm applied in its prologue)
        if self.lm_wp is None:
            self.lm_wp = ops.pack_weight(self.lm_head.weight.detach().to(BF16).contiguous())
        t0 = (P - 1) // 16 * 16
        rows = h[t0:t0 + 16]
        if self.wide_hidden:   # K = hidden cut over workgroups: the ragged-batch form of norm + lm_head, one tile
            if "wxn" not in pf:
                pf["wxn"] = torch.zeros(2, 16 * H, dtype=BF16, device=dev)
                pf["wdyn"] = torch.tensor([[0, 0, 16, 0, 0, 0, 0, 0]] * 2, dtype=torch.int32, device=dev)
                pf["wgws"] = torch.zeros(ops.lib().dfl_gemm_batch_ws_bytes(self.V, H), dtype=torch.uint8, device=dev)
                pf["wh"] = torch.zeros(2, 16, H, dtype=BF16, device=dev)
                pf["wlogits"], pf["wids"] = torch.zeros(2, 16, self.V, dtype=BF16, device=dev), torch.zeros(2, 16, dtype=torch.int64, device=dev)
            pf["wh"][0].copy_(rows)
            ops.norm_frag_batch(pf["wh"], 1, self.norm, self.eps, pf["wxn"], pf["wdyn"], ops.DYN_BS)
            ops.gemm_argmax_batch(self.lm_wp, ops.brows_frag(pf["wxn"]), 1, self.V, H, 0, 16, pf["wgws"], pf["wids"], 0,
                                  pf["wdyn"], nrows_dyn_word=ops.DYN_BS, logits=pf["wlogits"])
            pf["logits"].copy_(pf["wlogits"][0])
        else:
            ss = rows.float().pow(2).sum(-1).contiguous()
            ops.gemm_argmax(self.lm_wp, ops.rows_normed(rows, ss, 1, self.norm, self.eps), self.V, H, 0, 16,
                            self.ws["argmax_ws"], pf["ids"], 0, logits=pf["logits"])
        logits = pf["logits"][P - 1 - t0].clone().view(1, 1, self.V)
        # (hidden_states[L], HF's final-normed state, is not kept: nothing on the path reads it — model/utils.py:16-25
        # taps layer OUTPUTS, and build_target_layer_ids never picks the last layer)
        cache.length = P
        return SimpleNamespace(logits=logits, hidden_states=_HiddenStates(kept, self.L + 1) if output_hidden_states else None)

    def _prefill_hf(self, input_ids, cache, output_hidden_states):
        """Through the wrapped model, K/V copied into the preallocated cache (odd widths, prefill="hf")."""
        from transformers import DynamicCache
        P = input_ids.shape[1]
        tmp = DynamicCache()
        pos = torch.arange(P, device=self._dev).unsqueeze(0)
        out = self.hf(input_ids, position_ids=pos, past_key_values=tmp, use_cache=True, logits_to_keep=1,
                      output_hidden_states=output_hidden_states)
        for i in range(self.L):
            cache.k[i, :, :P].copy_(tmp.layers[i].keys[0])
            cache.v[i, :, :P].copy_(tmp.layers[i].values[0])
        cache.length = P
        return out

    def _moe_mlp(self, i: int, lw: dict, tiles, hrow, taps, sl) -> None:
        """Qwen3MoeSparseMoeBlock of layer i on the block rows (tf:models/qwen3_moe/modeling_qwen3_moe.py): norm, router
        GEMM, softmax/top-k/renormalise, gate/up of every active expert in one launch, the routing-weighted down
        projections as fp32 K-part sums, then residual add + tap + the NEXT stage's RMSNorm in one row-wise launch."""
        ws, H, E = self.ws, self.H, self.E
        nxt = self.layers[i + 1]["ln1"] if i + 1 < self.L else self.norm
        for t, dt in tiles:
            if self.moe_router_fused:
                ops.moe_router(h=hrow[t], norm_w=lw["ln2"], eps=self.eps, xn=ws["xn"][t], wp_router=lw["router"], K=H, E=E,
                               top_k=self.top_k, norm_topk=self.norm_topk, rlog=ws["rlog"][t], wt=ws["wt"][t],
                               active=ws["active"], lst=ws["elist"], n_active=ws["n_active"], ticket=ws["rticket"], dyn=dt,
                               dyn_word=ops.DYN_BS)
            else:
                ops.norm_pack(norm_w=lw["ln2"], frag=ws["xn"][t], H=H, eps=self.eps, resid_in=hrow[t], dyn=dt,
                              dyn_word=ops.DYN_BS)
                ops.gemm_resid(lw["router"], self.src["xn"][t], ws["rlog"].shape[2], H, ws["rlog"][t], add_residual=False,
                               dyn=dt)
                ops.moe_route(ws["rlog"][t], E, self.top_k, self.norm_topk, ws["wt"][t], ws["active"], ws["elist"],
                              ws["n_active"], dyn=dt, dyn_word=ops.DYN_BS)
            if self.debug_routing is not None and t == 0:    # tests: the routing weights of every MoE layer
                self.debug_routing.append((i, ws["wt"][0].clone()))
            mev = self.moe_events if (self.moe_events is not None and self.moe_events[0] == i and t == 0) else None
            if mev is not None:
                mev[1].record()
            if H <= 2048 and self.moe_pair_kernel:   # one LDS meeting per (gate, up) tile pair: 64 KB tiles at K = 2048
                ops.moe_gate_up(lw["gu_e"], ws["xn"][t], E, self.Ie, H, ws["act_e"], ws["elist"], ws["n_active"], dyn=dt,
                                valid_word=ops.DYN_BS)
            else:
                ops.gemm_silu_mul_experts(lw["gu_e"], self.src["xn"][t], E, self.Ie, H, ws["act_e"], ws["elist"],
                                          ws["n_active"], dyn=dt)
            if mev is not None:
                mev[2].record()
                mev[3].copy_(ws["n_active"])
            ops.moe_down(lw["down_e"], ws["act_e"], ws["wt"][t], ws["elist"], ws["n_active"], E, H, self.Ie,
                         self.moe_nsplit, ws["moe_part"])
            tap = taps[16 * t:16 * t + 16, sl[0] * H:(sl[0] + 1) * H] if sl else None
            ops.norm_pack(norm_w=nxt, frag=ws["xn1"][t], H=H, eps=self.eps, part=ws["moe_part"], nsplit=self.moe_nsplit,
                          part_split=16 * H, ldp=H, resid_in=hrow[t], h_out=hrow[t], h_out2=tap,
                          ld2=taps.stride(0) if tap is not None else 0, dyn=dt, dyn_word=ops.DYN_BS)

    def moe_mlp_tiles(self, lw: dict, R: int, MT: int, dyn: torch.Tensor, xn: torch.Tensor, part: torch.Tensor) -> int:
        """Qwen3MoeSparseMoeBlock (tf:models/qwen3_moe/modeling_qwen3_moe.py) of one layer for R 16-row tiles that went
        through the layer's attention and dense projections together (requests of a ragged batch, candidate blocks of
        one verify): tile r routes ITS rows — router GEMM on its ln2-normalised fragments xn[r], fp32 softmax / top-k /
        renormalise — and streams ITS active experts (gate/up + SiLU, routing-weighted down projections); the experts a
        tile uses are its own, so there is no weight stream to share unless two tiles pick the same expert.  The
        routing-weighted sums land as fp32 shares in the batch partial-sum layout part[share][MT * 16][H], where the
        next dfl_norm_frag_batch adds them to the residual stream (one rounding).  dyn: [MT, 8] length records (valid
        rows = the DYN_BS word).  Returns the share count."""
        w, ns, H = self.ws, self.moe_nsplit, self.H
        if R >= self.moe_shared_min and self.moe_shared_pass and H % 128 == 0 and self.Ie % 64 == 0:
            return self._moe_mlp_shared(lw, R, MT, dyn, xn, part)
        # The share stride is what the consumer reads with — dfl_norm_frag_batch(nsplit=ns) strides by
        # batch_tiles(R) * 16 * H (ops.norm_frag_batch) — NOT the caller's tile-slot count: a candidate verifier
        # always allocates MT = 4 slots, and for R <= 2 (batch_tiles = 2) share 1 would otherwise land at 64 H
        # while it is read at 32 H (half of the experts' down projections dropped).
        mt = ops.batch_tiles(R)
        if mt > MT:
            raise ValueError(f"moe_mlp_tiles: {R} tiles need {mt} tile slots, the buffers have {MT}")
        pv = part[:ns * mt * 16 * H].view(ns, mt * 16, H)
        for r in range(R):
            dt, x = dyn[r], xn[r]
            if self.moe_router_fused:   # gate Linear + routing in one launch on the tile's normalised fragments
                ops.moe_router(h=None, norm_w=None, eps=self.eps, xn=x, wp_router=lw["router"], K=H, E=self.E,
                               top_k=self.top_k, norm_topk=self.norm_topk, rlog=w["rlog"][0], wt=w["wt"][0],
                               active=w["active"], lst=w["elist"], n_active=w["n_active"], ticket=w["rticket"], dyn=dt,
                               dyn_word=ops.DYN_BS)
            else:
                ops.gemm_resid(lw["router"], ops.rows_frag(x), w["rlog"].shape[2], H, w["rlog"][0], add_residual=False, dyn=dt)
                ops.moe_route(w["rlog"][0], self.E, self.top_k, self.norm_topk, w["wt"][0], w["active"], w["elist"],
                              w["n_active"], dyn=dt, dyn_word=ops.DYN_BS)
            if H <= 2048 and self.moe_pair_kernel:
                ops.moe_gate_up(lw["gu_e"], x, self.E, self.Ie, H, w["act_e"], w["elist"], w["n_active"], dyn=dt,
                                valid_word=ops.DYN_BS)
            else:
                ops.gemm_silu_mul_experts(lw["gu_e"], ops.rows_frag(x), self.E, self.Ie, H, w["act_e"], w["elist"],
                                          w["n_active"], dyn=dt)
            ops.moe_down(lw["down_e"], w["act_e"], w["wt"][0], w["elist"], w["n_active"], self.E, H, self.Ie, ns,
                         w["moe_part"])
            pv[:, r * 16:(r + 1) * 16].copy_(w["moe_part"])
        return ns

    def _moe_mlp_shared(self, lw: dict, R: int, MT: int, dyn: torch.Tensor, xn: torch.Tensor, part: torch.Tensor) -> int:
        """The same for three or four tiles in ONE pass over the experts: the R x 16 rows are routed, sorted by expert and
        gathered like prompt rows (dfl_prefill_moe_*, csrc/prefill.hip), so that an expert several tiles picked is
        streamed once (four tiles of 16 rows x top-8 touch nearly all of 128 experts: one pass over them instead of four
        over ~80 each).  Same rounding points as the per-tile kernels; the rows' fp32 sums land as ONE share in `part`.
        Rows beyond a tile's valid count are routed too (zero fragments) and ignored by the norm launch that follows."""
        H, P = self.H, R * 16
        if self._moe_sh is None:
            ep = (self.E + 127) // 128 * 128
            self._moe_sh = dict(sc=ops.prefill_moe_scratch(4 * 16, H, self.Ie, self.E, self.top_k, ep, self._dev),   # (<= 4 tiles)
                                rlog=torch.zeros(4, 16, ep, dtype=BF16, device=self._dev),
                                gws=torch.zeros(ops.lib().dfl_gemm_batch_ws_bytes(ep, H), dtype=torch.uint8, device=self._dev))
        sh = self._moe_sh
        sc, L, st = sh["sc"], ops.lib(), ops._stream()
        ops.gemm_resid_batch(lw["router_p"], ops.brows_frag(xn), R, sh["rlog"].shape[2], H, sh["rlog"], add_residual=False,
                             ws=sh["gws"], dyn=dyn)
        ops.prefill_moe_route(sh["rlog"].view(64, -1), P, sc, self.norm_topk)
        gu, dn = lw["gu_e"], lw["down_e"]
        ops.check(L.dfl_prefill_moe_gemm_silu(gu.data_ptr(), gu.stride(0), xn.data_ptr(), sc["items"].data_ptr(),
                                              sc["n_items"].data_ptr(), sc["max_items"], self.Ie, H, sc["act_g"].data_ptr(),
                                              sc["rows_per_item"], sc["src_row"].data_ptr(), sc["zeros"].data_ptr(), st),
                  "dfl_prefill_moe_gemm_silu")
        ops.check(L.dfl_prefill_moe_gemm_down(dn.data_ptr(), dn.stride(0), sc["act_g"].data_ptr(), sc["items"].data_ptr(),
                                              sc["n_items"].data_ptr(), sc["max_items"], H, self.Ie, sc["row_w"].data_ptr(),
                                              sc["out32"].data_ptr(), sc["rows_per_item"], st), "dfl_prefill_moe_gemm_down")
        ops.check(L.dfl_prefill_moe_combine(sc["out32"].data_ptr(), sc["posmap"].data_ptr(), P, H, self.top_k, None, 0, None, 0,
                                            part.data_ptr(), st), "dfl_prefill_moe_combine")
        return 1

    # ---- the verify forward on the kernels
    def _verify_wide(self, block_ids, start, cache, bs, tap_layers, taps, logits_out, temperature, cos, sin):
        """Blocks of 17..32 rows in ONE pass over the weights: the two 16-row tiles go through the ragged-batch
        GEMMs (dfl_*_batch with R = 2: fp32 K-part sums of o_proj / down_proj, residual add + RMSNorm in
        dfl_norm_frag_batch) as if they were two requests, and through ONE attention launch with two query tiles on the
        request's single cache.  Same lines as verify(): model/dflash.py:249-257."""
        ws, H, R = self.ws, self.H, 2
        if self._wide is None:
            ks, dev = ops.batch_ksplit, self._dev
            xn = torch.zeros(2, 16 * H, dtype=BF16, device=dev)
            nmax, kmax = max(self.V, 2 * self.I, self.nqkv), max(H, self.I, self.q_dim)
            self._wide = dict(
                xn=xn, ids=torch.zeros(2, 16, dtype=torch.int64, device=dev),
                part_h=torch.zeros(max(ks(self.q_dim), ks(self.I), self.moe_nsplit if self.is_moe else 0) * 2 * 16 * H,
                                   dtype=torch.float32, device=dev),
                gws=torch.zeros(max(ops.lib().dfl_gemm_batch_ws_bytes(n, k) for n, k in ((nmax, H), (H, kmax))),
                                dtype=torch.uint8, device=dev),
                src=dict(xn=ops.brows_frag(xn), attn=ops.brows_frag(ws["attn"]), act=ops.brows_frag(ws["act"])))
        ww = self._wide
        s, gws, part_h, xn = ww["src"], ww["gws"], ww["part_h"], ww["xn"]
        dyn2 = cache.dyn[:16].view(2, 8)
        h3, xq3 = ws["h"].view(2, 16, H), ws["xq"].view(2, 16, self.nqkv)
        ww["ids"].view(-1)[:bs].copy_(block_ids[:bs])
        ops.embed_rows_batch(self.embed, ww["ids"], R, h3, H, ws["ss_emb"].view(2, 16), dyn2, ops.DYN_BS)
        tap3 = None if taps is None else taps.view(2, 16, taps.shape[1])
        slots = {}
        for j, l in enumerate(tap_layers):
            slots.setdefault(l, []).append(j)
        pend, ptap, pdup = 0, None, ()   # K and tap view of the down_proj whose sums wait in part_h

        def spread(dups):   # the other slots of a repeated tap id get the same rows (model/utils.py:16-25)
            for a, b in dups:
                taps[:, b * H:(b + 1) * H].copy_(taps[:, a * H:(a + 1) * H])

        pns = None   # share count of the pending sums when they are MoE expert shares, not K parts
        for i, lw in enumerate(self.layers):
            ops.norm_frag_batch(h3, R, lw["ln1"], self.eps, xn, dyn2, ops.DYN_BS, part=part_h if pend else None, N=H,
                                K=pend, tap=ptap, nsplit=pns)
            spread(pdup)
            ops.gemm_resid_batch(lw["qkv"], s["xn"], R, self.nqkv, H, xq3, add_residual=False, ws=gws, dyn=dyn2)
            ops.attn_head(xq=ws["xq"], q_col=0, k_col=self.q_dim, v_col=self.q_dim + self.kv_dim, n_q=self.n_q,
                          n_kv=self.n_kv, q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"], eps=self.eps, cos_tab=cos,
                          sin_tab=sin, kcache=cache.k[i], vcache=cache.v[i], scale=128 ** -0.5, causal=True, S=start,
                          tau=0, bs=bs, pos0=start, ws=ws["head_ws"], max_splits=self.max_splits, out_frag=ws["attn"],
                          q_tiles=2, out_tile_stride=ws["attn"].stride(0))
            ops.gemm_f32_batch(lw["o"], s["attn"], R, H, self.q_dim, part_h, dyn2)
            ops.norm_frag_batch(h3, R, lw["ln2"], self.eps, xn, dyn2, ops.DYN_BS, part=part_h, N=H, K=self.q_dim)
            if "gu_e" in lw:   # sparse-MoE layer: each of the two tiles routes its own rows (round 3)
                pns, pend = self.moe_mlp_tiles(lw, R, 2, dyn2, xn, part_h), 1
            else:
                ops.gemm_silu_mul_batch(lw["gu"], s["xn"], R, self.I, H, ws["act"], gws, dyn2)
                ops.gemm_f32_batch(lw["down"], s["act"], R, H, self.I, part_h, dyn2)
                pns, pend = None, self.I
            sl = slots.get(i, ())   # a tapped layer's rows exist once the next norm launch has added these sums
            ptap = tap3[:, :, sl[0] * H:(sl[0] + 1) * H] if sl else None
            pdup = [(sl[0], b) for b in sl[1:]]
        ops.norm_frag_batch(h3, R, self.norm, self.eps, xn, dyn2, ops.DYN_BS, part=part_h, N=H, K=pend, tap=ptap,
                            nsplit=pns)
        spread(pdup)
        post = ws["post"]
        logits = logits_out
        if temperature >= 1e-5:
            logits = torch.empty(32, self.V, dtype=BF16, device=self._dev)
        ops.gemm_argmax_batch(self.lm_wp, s["xn"], R, self.V, H, 0, 16, gws, post.view(2, 16), 0, dyn2,
                              nrows_dyn_word=ops.DYN_BS, logits=None if logits is None else logits.view(2, 16, self.V))
        posterior = post[:bs].unsqueeze(0) if temperature < 1e-5 else sample(logits[:bs].unsqueeze(0), temperature)
        cache.length = start + bs
        return posterior, taps

    def raise_if_failed(self) -> None:
        """fuse_oproj only: a dfl_attn_head_oproj launch whose o_proj workgroups gave up waiting (2 ms) leaves a flag."""
        if self.fuse_oproj and int(self.ws["sync"][ops.ATTN_OPROJ_FAIL_WORD]) != 0:
            self.ws["sync"].zero_()
            raise RuntimeError("dfl_attn_head_oproj: an o_proj workgroup gave up waiting for the attention stage")

    @torch.inference_mode()
    def verify(self, block_ids: torch.Tensor, start: int, cache: TargetKVCache, *, tap_layers: Sequence[int] = (),
               temperature: float = 0.0, logits_out: Optional[torch.Tensor] = None,
               taps_out: Optional[torch.Tensor] = None, dyn_lengths: bool = False):
        """block_ids int64 [bs] at positions start..start+bs-1 (cache rows alike), bs <= 32.
        Returns (posterior ids int64 [1, bs], taps bf16 [32, len(tap_layers)*H] or None).
        K/V of all bs rows are written; the caller crops to what it accepts.
        taps_out: the caller's own [32, len(tap_layers)*H] buffer (a decode session keeps the
        rows until its next draft; sessions interleaved on one target must not share one).
        logits_out: bf16 [16 * tiles, V].  Blocks of 17..32 rows run as two 16-row tiles: one pass over the
        weights through the ragged-batch GEMMs (`_verify_wide`; an MoE layer's expert MLP per tile), or one launch per
        tile of every single-request GEMM (wide_one_pass = False); both query tiles share the attention launch.
        dyn_lengths (bs <= 16, attention stage "head"): the launches take S / pos0 from the cache's device record alone
        (kept by dfl_accept_commit_rearm_t) — `start` is then only an upper bound that sizes the attention's key splits
        and the RoPE table, and the sequence can be captured into a hipGraph (DecodeSession.capture)."""
        bs = block_ids.numel()
        if self.wide_hidden:
            raise NotImplementedError("NativeTarget.verify: hidden > 4096 runs through the ragged-batch path "
                                      "(dflash_generate / dflash_generate_batch / BatchedDecoder)")
        if bs < 1 or bs > 32:
            raise ValueError("verify takes 1..32 block rows")
        if bs > 16 and self.attn_impl != "head":
            raise ValueError("blocks of more than 16 rows need attn_impl='head'")
        if start + bs > cache.max_rows:
            raise ValueError("target KV cache too small")
        ws, H = self.ws, self.H
        if self.lm_wp is None:
            self.lm_wp = ops.pack_weight(self.lm_head.weight.detach().to(BF16).contiguous())
        cos, sin = self._rope_tab(start + bs + 64)
        # the verify's kernels read only the tiles' valid-row counts from the record (the attention takes immediates):
        # it is rewritten when the block size changes, not every cycle
        # — the round-1 fused stage (attn_impl="fused") reads S / TAU / POS0 from the record: rewritten every call there
        if dyn_lengths:
            if bs > 16 or self.attn_impl != "head" or self.fuse_oproj:
                raise ValueError("dyn_lengths needs a block of <= 16 rows and the plain 'head' attention stage")
        elif self.attn_impl != "head" or getattr(cache, "_dyn_bs", None) != bs:
            ops.set_dyn2(cache.dyn, start, 0, bs, start)
            cache._dyn_bs = bs
        dyn = cache.dyn[:8]
        tiles = [(t, cache.dyn[8 * t:8 * t + 8]) for t in range((bs + 15) // 16)]
        taps = None
        tap_layers = list(tap_layers)
        if tap_layers:
            if max(tap_layers) >= self.L - 1:
                raise NotImplementedError("tapping the last layer (post-norm state) is not supported")
            key = len(tap_layers)
            if taps_out is not None:
                if taps_out.shape != (32, key * H) or taps_out.dtype != BF16 or not taps_out.is_contiguous():
                    raise ValueError("taps_out must be a contiguous bf16 [32, len(tap_layers)*H] tensor")
                taps = taps_out
            else:
                if key not in self._taps:
                    self._taps[key] = torch.zeros(32, key * H, dtype=BF16, device=self._dev)
                taps = self._taps[key]
        Ls, src = self.layers, self.src
        if len(tiles) == 2 and self.wide_one_pass and self.attn_impl == "head":
            return self._verify_wide(block_ids, start, cache, bs, tap_layers, taps, logits_out, temperature, cos, sin)
        hrow = [ws["h"][16 * t:16 * t + 16] for t in range(2)]
        for t, dt in tiles:
            ops.embed_rows(self.embed, block_ids[16 * t:], hrow[t], H, ws["ss_emb"][16 * t:], dt, ops.DYN_BS)
        fuse_o = self.fuse_oproj and self.attn_impl == "head" and len(tiles) == 1 and self.q_dim <= 4096
        prev_moe = False   # an MoE layer leaves its successor's input already normalised (frag16 in ws["xn1"])
        for i, lw in enumerate(Ls):
            x1 = src["xn1"] if prev_moe else src["ln1"][i]
            if self.attn_impl == "head":
                # q/k/v as finished bf16 Linear outputs (no K split: 192 workgroups x 2 column tiles), then
                # one launch: q/k-norm + RoPE + append + causal attention + split merge
                for t, dt in tiles:
                    ops.gemm_resid(lw["qkv"], x1[t], self.nqkv, H, ws["xq"][16 * t:], add_residual=False, dyn=dt)
                kw = dict(xq=ws["xq"], q_col=0, k_col=self.q_dim, v_col=self.q_dim + self.kv_dim, n_q=self.n_q,
                          n_kv=self.n_kv, q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"], eps=self.eps, cos_tab=cos,
                          sin_tab=sin, kcache=cache.k[i], vcache=cache.v[i], scale=128 ** -0.5, causal=True,
                          S=start, tau=0, bs=bs, pos0=start, ws=ws["head_ws"], max_splits=self.max_splits,
                          dyn=dyn if dyn_lengths else None)
                if fuse_o:
                    ops.attn_head_oproj(**kw, attn_frag=ws["attn"][0], wo=lw["o"], H=H, h_io=hrow[0],
                                        ss_out=ws["ss_h"][0], sync=ws["sync"])
                else:
                    ops.attn_head(**kw, out_frag=ws["attn"], q_tiles=len(tiles), out_tile_stride=ws["attn"].stride(0))
            else:
                ops.gemm_f32(lw["qkv"], x1[0], None, 1, self.nqkv, H, self.ks_qkv, ws["part"], dyn)
                ops.attn_fused(qkv=ws["part"], nsplit=self.ks_qkv, split_stride=16 * self.nqkv, ld=self.nqkv, q_col=0,
                               k_col=self.q_dim, v_col=self.q_dim + self.kv_dim, ctx_row0=0, blk_row0=0, n_q=self.n_q,
                               n_kv=self.n_kv, q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"], eps=self.eps,
                               cos_tab=cos, sin_tab=sin, kcache=cache.k[i], vcache=cache.v[i], dyn=dyn,
                               scale=128 ** -0.5, kv_len_max=start + bs, ws=ws["attn_ws"],
                               max_splits=self.max_splits, out_frag=ws["attn"][0], causal=True)
            for t, dt in tiles:
                if not fuse_o:
                    # (an MoE block normalises its rows itself — dfl_moe_router: no partial sums of squares wanted, and
                    # without them a narrow o_proj may run in half tiles on twice the workgroups)
                    ops.gemm_resid(lw["o"], src["attn"][t], H, self.q_dim, hrow[t], add_residual=True,
                                   ss_out=None if "gu_e" in lw else ws["ss_h"][t], dyn=dt)
            # every slot j with tap_layers[j] == i: build_target_layer_ids repeats layers for shallow
            # targets and the reference concatenates the same state twice (model/utils.py:16-25)
            sl = [j for j, l in enumerate(tap_layers) if l == i]
            prev_moe = "gu_e" in lw
            if prev_moe:
                self._moe_mlp(i, lw, tiles, hrow, taps, sl)
            else:
                gev = self.gu_events if (self.gu_events is not None and self.gu_events[0] == i) else None
                if gev is not None:       # bench.py: this layer's gate/up launch between two events on the launch stream
                    gev[1].record()
                for t, dt in tiles:
                    ops.gemm_silu_mul(lw["gu"], src["ln2"][i][t], self.I, H, ws["act"][t], dt)
                if gev is not None:
                    gev[2].record()
                for t, dt in tiles:
                    tap = taps[16 * t:16 * t + 16, sl[0] * H:(sl[0] + 1) * H] if sl else None
                    ops.gemm_resid(lw["down"], src["act"][t], H, self.I, hrow[t], add_residual=True,
                                   ss_out=ws["ss_h"][t], tap=tap, dyn=dt)
            for j in sl[1:]:
                taps[:, j * H:(j + 1) * H].copy_(taps[:, sl[0] * H:(sl[0] + 1) * H])
        post = ws["post"]
        logits = logits_out
        if temperature >= 1e-5:
            logits = torch.empty(16 * len(tiles), self.V, dtype=BF16, device=self._dev)
        fin = src["xn1"] if prev_moe else src["final"]   # after an MoE layer the rows are final-normed already
        for t, dt in tiles:
            ops.gemm_argmax(self.lm_wp, fin[t], self.V, H, 0, min(16, bs - 16 * t), ws["argmax_ws"], post, 16 * t,
                            dyn=dt, logits=None if logits is None else logits[16 * t:16 * t + 16])
        if temperature < 1e-5:
            posterior = post[:bs].unsqueeze(0)
        else:
            posterior = sample(logits[:bs].unsqueeze(0), temperature)
        cache.length = start + bs
        return posterior, taps
