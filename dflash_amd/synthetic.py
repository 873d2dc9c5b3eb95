"""Seeded random-init weights of a given architecture.

No checkpoints exist offline, so tests, the golden-vector
generator and bench.py all build weights from a seed.  Generation is on the CPU
generator in fp32 then cast, so the same seed yields bit-identical weights in
the build container and on the GPU box.
"""
from __future__ import annotations

import torch

from .config import DFlashConfig


def make_draft_state_dict(cfg: DFlashConfig, seed: int = 0, dtype=torch.bfloat16, std: float = 0.02,
                          device="cpu", norm_jitter: float = 0.1) -> dict:
    """State dict with the reference's key names.  Linear weights
    ~ N(0, std) (HF default init), norm weights ~ 1 + norm_jitter * N(0,1) so a
    norm-weight bug cannot hide behind all-ones."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in cfg.state_dict_shapes().items():
        if len(shape) == 1:
            t = 1.0 + norm_jitter * torch.randn(shape, generator=g)
        else:
            # big tensors row-chunked: keeps peak host memory low for 8B-shaped drafts
            t = torch.empty(shape, dtype=dtype)
            rows = max(1, (1 << 24) // shape[1])
            for r in range(0, shape[0], rows):
                n = min(rows, shape[0] - r)
                t[r:r + n] = (torch.randn((n, shape[1]), generator=g) * std).to(dtype)
        sd[name] = t.to(dtype).to(device)
    return sd


def make_hf_qwen3(dims: dict, device, dtype=torch.bfloat16, attn_impl: str = "sdpa"):
    """A HF `Qwen3ForCausalLM` (the reference's target class, benchmark.py:401) built from
    dimensions with HF default init, directly on `device`.  dims keys as in
    config.QWEN3_8B_TARGET."""
    from transformers import Qwen3Config, Qwen3ForCausalLM
    cfg = Qwen3Config(vocab_size=dims["vocab_size"], hidden_size=dims["hidden_size"],
                      intermediate_size=dims["intermediate_size"], num_hidden_layers=dims["num_layers"],
                      num_attention_heads=dims["num_heads"], num_key_value_heads=dims["num_kv_heads"],
                      head_dim=dims["head_dim"], max_position_embeddings=40960, rms_norm_eps=1e-6,
                      rope_parameters={"rope_type": "default", "rope_theta": dims["rope_theta"]},
                      tie_word_embeddings=False, attention_bias=False)
    cfg._attn_implementation = attn_impl
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            m = Qwen3ForCausalLM(cfg)
    finally:
        torch.set_default_dtype(prev)
    return m.eval()


def impose_greedy_walk(hf_model, seed: int = 1234) -> torch.Tensor:
    """Give a random-init target a noise-robust greedy rule without touching its
    architecture, byte count or FLOPs.  Plain random weights have ~zero top-2 logit
    margins, so the bf16 argmax flips between a 1-token and a 16-token forward and no
    greedy continuation survives a verify.  Here: seeded random weights, embedding std
    1.0, o_proj/down_proj scaled by 0.02 (the residual stream stays embedding-dominated),
    lm_head row perm[t] = 0.02 * embedding[t] for a seeded single-cycle permutation.
    The greedy next token of token t is then perm[t] with a logit margin of ~0.02*H.
    Returns perm (int64 [V], on the model's device)."""
    dev = hf_model.lm_head.weight.device
    V, H = hf_model.lm_head.weight.shape
    g = torch.Generator(device=dev).manual_seed(seed)
    cyc = torch.randperm(V, generator=g, device=dev)
    perm = torch.empty(V, dtype=torch.long, device=dev)
    perm[cyc] = torch.roll(cyc, -1)  # one cycle through the whole vocabulary
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(V, device=dev)
    with torch.no_grad():
        emb = torch.randn(V, H, generator=g, device=dev, dtype=torch.float32)
        hf_model.model.embed_tokens.weight.copy_(emb)
        hf_model.lm_head.weight.copy_(emb[inv] * 0.02)
        del emb
        for layer in hf_model.model.layers:
            layer.self_attn.o_proj.weight.mul_(0.02)
            if hasattr(layer.mlp, "experts"):      # sparse-MoE layer: the experts' fused down projections
                layer.mlp.experts.down_proj.mul_(0.02)
            else:
                layer.mlp.down_proj.weight.mul_(0.02)
    return perm


def greedy_walk(perm: torch.Tensor, prompt: torch.Tensor, n: int) -> torch.Tensor:
    """prompt ids followed by n tokens of the walk G[p+1] = perm[G[p]] (host loop)."""
    pc = perm.cpu()
    out = prompt.flatten().cpu().tolist()
    for _ in range(n):
        out.append(int(pc[out[-1]]))
    return torch.tensor(out, dtype=torch.long)
