"""Draft-model hyper-parameters.

The reference keeps these on a HF `Qwen3Config` carrying three extra attributes
(`block_size`, `num_target_layers`, `dflash_config{mask_token_id,target_layer_ids}`,
model/dflash.py:157,162-163).  `DFlashConfig.from_any` accepts that object, a
plain dict (a parsed `config.json`) or another `DFlashConfig`, so the draft model
is constructible exactly like the reference's without importing transformers.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

from .utils import build_target_layer_ids


@dataclass
class DFlashConfig:
    hidden_size: int
    num_hidden_layers: int            # draft layers
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    num_target_layers: int
    block_size: int = 16
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_position_embeddings: int = 40960
    mask_token_id: Optional[int] = None
    target_layer_ids: list = field(default_factory=list)
    attention_bias: bool = False

    def __post_init__(self):
        if not self.target_layer_ids:
            self.target_layer_ids = build_target_layer_ids(self.num_target_layers, self.num_hidden_layers)
        if self.attention_bias:
            raise NotImplementedError("attention_bias=True drafts are not supported by the HIP path")

    @property
    def q_dim(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_key_value_heads * self.head_dim

    @property
    def fc_in(self) -> int:
        return len(self.target_layer_ids) * self.hidden_size

    @classmethod
    def from_any(cls, cfg: Any) -> "DFlashConfig":
        if isinstance(cfg, cls):
            return cfg
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        dfl = get("dflash_config") or {}
        rope = get("rope_parameters") or {}
        theta = get("rope_theta") or (rope.get("rope_theta") if isinstance(rope, dict) else None) or 1e6
        heads = get("num_attention_heads")
        # model/dflash.py:56,97: a layer whose layer_types entry is "sliding_attention" attends within config.sliding_window
        # only.  The kernels attend over the whole cached prefix: such a draft is rejected here, before anything is built.
        lt = get("layer_types") or ()
        if any(t == "sliding_attention" for t in lt) and get("sliding_window"):
            raise NotImplementedError("sliding-window attention drafts (layer_types 'sliding_attention') are not supported "
                                      "by the HIP path")
        return cls(
            hidden_size=get("hidden_size"), num_hidden_layers=get("num_hidden_layers"),
            num_attention_heads=heads, num_key_value_heads=get("num_key_value_heads", heads),
            head_dim=get("head_dim") or get("hidden_size") // heads,
            intermediate_size=get("intermediate_size"), vocab_size=get("vocab_size"),
            num_target_layers=get("num_target_layers"), block_size=get("block_size", 16),
            rms_norm_eps=get("rms_norm_eps", 1e-6), rope_theta=float(theta),
            max_position_embeddings=get("max_position_embeddings", 40960),
            mask_token_id=dfl.get("mask_token_id"),
            target_layer_ids=list(dfl.get("target_layer_ids") or []),
            attention_bias=bool(get("attention_bias", False)),
        )

    def state_dict_shapes(self) -> dict:
        """Parameter names and shapes — identical to the reference module's
        state dict, biases absent."""
        H, I = self.hidden_size, self.intermediate_size
        s = {"fc.weight": (H, self.fc_in), "hidden_norm.weight": (H,), "norm.weight": (H,)}
        for i in range(self.num_hidden_layers):
            p = f"layers.{i}."
            s[p + "self_attn.q_proj.weight"] = (self.q_dim, H)
            s[p + "self_attn.k_proj.weight"] = (self.kv_dim, H)
            s[p + "self_attn.v_proj.weight"] = (self.kv_dim, H)
            s[p + "self_attn.o_proj.weight"] = (H, self.q_dim)
            s[p + "self_attn.q_norm.weight"] = (self.head_dim,)
            s[p + "self_attn.k_norm.weight"] = (self.head_dim,)
            s[p + "mlp.gate_proj.weight"] = (I, H)
            s[p + "mlp.up_proj.weight"] = (I, H)
            s[p + "mlp.down_proj.weight"] = (H, I)
            s[p + "input_layernorm.weight"] = (H,)
            s[p + "post_attention_layernorm.weight"] = (H,)
        return s


# Architectures BASELINE.json names (public model-card values).
QWEN3_8B_DRAFT = dict(hidden_size=4096, num_hidden_layers=5, num_attention_heads=32, num_key_value_heads=8,
                      head_dim=128, intermediate_size=12288, vocab_size=151936, num_target_layers=36,
                      block_size=16, rope_theta=1e6, mask_token_id=151669)
QWEN3_8B_TARGET = dict(vocab_size=151936, hidden_size=4096, num_layers=36, num_heads=32, num_kv_heads=8,
                       head_dim=128, intermediate_size=12288, rope_theta=1e6)
QWEN3_4B_DRAFT = dict(hidden_size=2560, num_hidden_layers=5, num_attention_heads=32, num_key_value_heads=8,
                      head_dim=128, intermediate_size=9728, vocab_size=151936, num_target_layers=36,
                      block_size=16, rope_theta=1e6, mask_token_id=151669)
QWEN3_4B_TARGET = dict(vocab_size=151936, hidden_size=2560, num_layers=36, num_heads=32, num_kv_heads=8,
                       head_dim=128, intermediate_size=9728, rope_theta=1e6)
