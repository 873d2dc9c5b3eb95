"""Test-side *target* models for the DFlash loop.  TEST INFRASTRUCTURE ONLY.

The target LLM is the caller's model: it sits outside the hot path (SURVEY.md
§8a "Calls out of the path").  Tests and the bench still need one, with seeded
random weights and no checkpoint, on CPU and on the GPU box:

* `TorchQwen3Target` — a small pure-torch Qwen3-style causal LM that offers
  exactly what `spec_generate` asks of `target` (model/dflash.py:210-255:
  callable with `position_ids`, `past_key_values`, `use_cache`,
  `logits_to_keep`, `output_hidden_states`; `.model.embed_tokens`, `.lm_head`,
  `.device`).  Same state-dict key names as HF `Qwen3ForCausalLM`.
* `HFTargetAdapter` — wraps a HF `Qwen3ForCausalLM` (transformers is third-party
  and present in the image) so the oracle can be driven with the very target
  the golden vectors were generated with.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .dflash_oracle import ListKVCache, apply_rotary_std, attention, rms_norm, rope_cos_sin, rope_inv_freq


class _Inner(nn.Module):
    def __init__(self, vocab: int, hidden: int, weight=None):
        super().__init__()
        self.embed_tokens = nn.Embedding(vocab, hidden, _weight=weight)


class TorchQwen3Target(nn.Module):
    def __init__(self, *, vocab_size: int, hidden_size: int, num_layers: int, num_heads: int,
                 num_kv_heads: int, head_dim: int, intermediate_size: int, rms_norm_eps: float = 1e-6,
                 rope_theta: float = 1e6, attn_impl: str = "eager", init_std: float = 0.02,
                 seed: Optional[int] = None, dtype=torch.float32, device="cpu", fill_fn=None):
        super().__init__()
        self.cfg = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size, num_hidden_layers=num_layers,
                                   num_attention_heads=num_heads, num_key_value_heads=num_kv_heads,
                                   head_dim=head_dim, intermediate_size=intermediate_size,
                                   rms_norm_eps=rms_norm_eps, rope_theta=rope_theta)
        self.attn_impl = attn_impl
        # fill_fn(shape) -> tensor: cheap deterministic fill for big timing-only models
        # (bench.py's CPU baseline); default is seeded N(0, init_std)
        self.model = _Inner(vocab_size, hidden_size, None if fill_fn is None else fill_fn((vocab_size, hidden_size)))
        self.lm_head = nn.Linear(hidden_size, vocab_size, bias=False, device="meta" if fill_fn else None)
        if fill_fn is not None:
            self.lm_head.weight = nn.Parameter(fill_fn((vocab_size, hidden_size)), requires_grad=False)
        H, D, I = hidden_size, head_dim, intermediate_size
        shapes = {"self_attn.q_proj.weight": (num_heads * D, H), "self_attn.k_proj.weight": (num_kv_heads * D, H),
                  "self_attn.v_proj.weight": (num_kv_heads * D, H), "self_attn.o_proj.weight": (H, num_heads * D),
                  "mlp.gate_proj.weight": (I, H), "mlp.up_proj.weight": (I, H), "mlp.down_proj.weight": (H, I)}
        ones = {"self_attn.q_norm.weight": (D,), "self_attn.k_norm.weight": (D,),
                "input_layernorm.weight": (H,), "post_attention_layernorm.weight": (H,)}
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        self.w = nn.ParameterDict()
        for i in range(num_layers):
            for k, s in shapes.items():
                t = fill_fn(s) if fill_fn is not None else torch.randn(s, generator=g) * init_std
                self.w[f"{i}|{k}".replace(".", "|")] = nn.Parameter(t, requires_grad=False)
            for k, s in ones.items():
                self.w[f"{i}|{k}".replace(".", "|")] = nn.Parameter(1.0 + 0.1 * torch.randn(s, generator=g))
        self.w["norm"] = nn.Parameter(1.0 + 0.1 * torch.randn(H, generator=g))
        if fill_fn is None:
            with torch.no_grad():
                self.model.embed_tokens.weight.copy_(torch.randn(vocab_size, H, generator=g) * init_std)
                self.lm_head.weight.copy_(torch.randn(vocab_size, H, generator=g) * init_std)
        self.requires_grad_(False)
        self.to(device=device, dtype=dtype)
        self.eval()

    @property
    def device(self):
        return self.lm_head.weight.device

    def _p(self, i, name):
        return self.w[f"{i}|{name}".replace(".", "|")]

    def new_cache(self):
        return ListKVCache()

    def load_hf_state_dict(self, sd: dict):
        """Copy weights from a HF Qwen3ForCausalLM state dict (same key names)."""
        with torch.no_grad():
            self.model.embed_tokens.weight.copy_(sd["model.embed_tokens.weight"])
            self.lm_head.weight.copy_(sd.get("lm_head.weight", sd["model.embed_tokens.weight"]))
            self.w["norm"].copy_(sd["model.norm.weight"])
            for key in list(self.w.keys()):
                if key == "norm":
                    continue
                i, rest = key.split("|", 1)
                self.w[key].copy_(sd[f"model.layers.{i}." + rest.replace("|", ".")])

    @torch.inference_mode()
    def forward(self, input_ids, position_ids=None, past_key_values: Optional[ListKVCache] = None,
                use_cache: bool = True, logits_to_keep: int = 0, output_hidden_states: bool = False, **_):
        c = self.cfg
        b, t = input_ids.shape
        h = self.model.embed_tokens(input_ids)
        past = past_key_values.get_seq_length() if past_key_values is not None else 0
        if position_ids is None:
            position_ids = torch.arange(past, past + t, device=h.device).unsqueeze(0)
        cos, sin = rope_cos_sin(position_ids, rope_inv_freq(c.head_dim, c.rope_theta).to(h.device), h.dtype)
        mask = None
        if t > 1:
            qi = torch.arange(t, device=h.device)[:, None] + past
            ki = torch.arange(past + t, device=h.device)[None, :]
            mask = torch.zeros(t, past + t, dtype=h.dtype, device=h.device)
            mask.masked_fill_(ki > qi, float("-inf"))
            mask = mask[None, None]
        hs = [h] if output_hidden_states else None
        for i in range(c.num_hidden_layers):
            res = h
            x = rms_norm(h, self._p(i, "input_layernorm.weight"), c.rms_norm_eps)
            q = F.linear(x, self._p(i, "self_attn.q_proj.weight")).view(b, t, -1, c.head_dim)
            k = F.linear(x, self._p(i, "self_attn.k_proj.weight")).view(b, t, -1, c.head_dim)
            v = F.linear(x, self._p(i, "self_attn.v_proj.weight")).view(b, t, -1, c.head_dim).transpose(1, 2)
            q = rms_norm(q, self._p(i, "self_attn.q_norm.weight"), c.rms_norm_eps).transpose(1, 2)
            k = rms_norm(k, self._p(i, "self_attn.k_norm.weight"), c.rms_norm_eps).transpose(1, 2)
            q, k = apply_rotary_std(q, k, cos, sin)
            if past_key_values is not None:
                k, v = past_key_values.update(k, v, i)
            o = attention(q, k, v, c.head_dim ** -0.5, self.attn_impl, causal_mask=mask)
            h = res + F.linear(o, self._p(i, "self_attn.o_proj.weight"))
            res = h
            x = rms_norm(h, self._p(i, "post_attention_layernorm.weight"), c.rms_norm_eps)
            x = F.linear(F.silu(F.linear(x, self._p(i, "mlp.gate_proj.weight"))) *
                         F.linear(x, self._p(i, "mlp.up_proj.weight")), self._p(i, "mlp.down_proj.weight"))
            h = res + x
            if output_hidden_states and i < c.num_hidden_layers - 1:
                hs.append(h)
        h = rms_norm(h, self.w["norm"], c.rms_norm_eps)
        if output_hidden_states:
            hs.append(h)  # HF: last entry is the post-norm state
        sl = slice(-logits_to_keep, None) if logits_to_keep else slice(None)
        logits = self.lm_head(h[:, sl, :])
        return SimpleNamespace(logits=logits, hidden_states=tuple(hs) if hs is not None else None)


class HFTargetAdapter:
    """Gives a HF causal LM the `new_cache()` hook the oracle loop uses; everything
    else is forwarded untouched (the reference passes a DynamicCache it created
    itself, model/dflash.py:214)."""

    def __init__(self, hf_model):
        self.hf = hf_model
        self.model = hf_model.model
        self.lm_head = hf_model.lm_head

    @property
    def device(self):
        return self.hf.device

    def new_cache(self):
        from transformers import DynamicCache
        return DynamicCache()

    def __call__(self, *a, **kw):
        return self.hf(*a, **kw)
