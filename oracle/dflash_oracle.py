"""CPU oracle for the DFlash per-cycle hot path.  TEST INFRASTRUCTURE ONLY.

This file is a torch-CPU restatement of the reference's algorithm for the path
named in BASELINE.json (`model/dflash.py` + `model/utils.py:4-34` +
`benchmark.py:44-251` + `benchmark_dynamic_schedule.py:260-434`).  It imports
neither the reference nor `transformers`; every function cites the reference
lines (or, prefixed `tf:`, the transformers-5.15.0 lines the reference calls
into) that it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker.  The product (`dflash_amd/`) never
does: it fails loudly when the HIP library is missing.

Parity pin: `tests/golden/make_golden.py` (run in the build container, where
`/root/reference` is importable) drives the *reference* implementation on
seeded random-weight models and stores inputs/outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against those vectors
bit-for-bit on CPU (fp32 and bf16).  The reference itself ships no tests or
known-answer vectors, so those generated fixtures are the pin.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional, Sequence

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# leaf ops (third-party arithmetic the reference calls into)
# --------------------------------------------------------------------------

def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen3RMSNorm.forward — tf:models/qwen3/modeling_qwen3.py:59-64.

    fp32 normalise, cast back to the input dtype, THEN multiply by the weight
    (so the product is rounded once more in the storage dtype)."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)


def rope_inv_freq(head_dim: int, theta: float) -> torch.Tensor:
    """compute_default_rope_parameters — tf:...modeling_qwen3.py:105-123."""
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))


def rope_cos_sin(position_ids: torch.Tensor, inv_freq: torch.Tensor, dtype: torch.dtype):
    """Qwen3RotaryEmbedding.forward — tf:...modeling_qwen3.py:125-137.

    fp32 outer product, cat(freqs, freqs), cos/sin in fp32, cast to `dtype`.
    position_ids: [B, T] -> cos, sin: [B, T, head_dim]."""
    inv = inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
    pos = position_ids[:, None, :].float()
    freqs = (inv @ pos).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """tf:...modeling_qwen3.py:140-144."""
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rotary_dflash(q, k, cos, sin):
    """model/dflash.py:22-28 — q takes only the LAST q_len rows of cos/sin,
    k takes all of them (k = [ctx rows ; noise rows])."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    q_len = q.size(-2)
    q_out = (q * cos[..., -q_len:, :]) + (rotate_half(q) * sin[..., -q_len:, :])
    k_out = (k * cos) + (rotate_half(k) * sin)
    return q_out, k_out


def apply_rotary_std(q, k, cos, sin):
    """Standard HF rotary (target model) — tf:...modeling_qwen3.py:147-169."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    """tf:...modeling_qwen3.py:172-182."""
    if n_rep == 1:
        return x
    b, h, s, d = x.shape
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def attention(q, k, v, scale: float, impl: str, causal_mask: Optional[torch.Tensor] = None):
    """Attention backends the reference can dispatch to (model/dflash.py:86-99).

    "eager": tf:...modeling_qwen3.py:185-207 (fp32 softmax, scores in storage
    dtype).  "sdpa": tf:integrations/sdpa_attention.py:79-... which ends in
    torch SDPA with enable_gqa when no mask is given (:98-102).
    q [B,Hq,Tq,D], k/v [B,Hkv,Tk,D] -> [B,Tq,Hq*D]."""
    n_rep = q.shape[1] // k.shape[1]
    if impl == "eager":
        kk = repeat_kv(k, n_rep)
        vv = repeat_kv(v, n_rep)
        w = torch.matmul(q, kk.transpose(2, 3)) * scale
        if causal_mask is not None:
            w = w + causal_mask
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(w, vv)
    elif impl == "sdpa":
        if causal_mask is None:
            kw = {"enable_gqa": True} if n_rep > 1 else {}
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0,
                                               scale=scale, is_causal=False, **kw)
        else:
            kk = repeat_kv(k, n_rep)
            vv = repeat_kv(v, n_rep)
            o = F.scaled_dot_product_attention(q, kk, vv, attn_mask=causal_mask, dropout_p=0.0,
                                               scale=scale, is_causal=False)
    else:
        raise ValueError(impl)
    b, h, t, d = o.shape
    return o.transpose(1, 2).reshape(b, t, h * d)


def swiglu_mlp(x, w_gate, w_up, w_down):
    """Qwen3MLP.forward — tf:...modeling_qwen3.py:81-83."""
    return F.linear(F.silu(F.linear(x, w_gate)) * F.linear(x, w_up), w_down)


# --------------------------------------------------------------------------
# model/utils.py:4-34
# --------------------------------------------------------------------------

def build_target_layer_ids(num_target_layers: int, num_draft_layers: int) -> list[int]:
    """model/utils.py:4-14 (Python round() = banker's rounding)."""
    if num_draft_layers == 1:
        return [num_target_layers // 2]
    lo, hi = 1, num_target_layers - 3
    return [int(round(lo + (i * (hi - lo)) / (num_draft_layers - 1))) for i in range(num_draft_layers)]


def extract_context_feature(hidden_states: Sequence[torch.Tensor], layer_ids: Sequence[int]) -> torch.Tensor:
    """model/utils.py:16-25 — tap l is hidden_states[l + 1] (index 0 = embeddings)."""
    return torch.cat([hidden_states[l + 1] for l in layer_ids], dim=-1)


def sample(logits: torch.Tensor, temperature: float = 0.0) -> torch.Tensor:
    """model/utils.py:27-34.  T<1e-5: argmax (first max index).  Otherwise softmax
    in the logits' dtype and one torch.multinomial draw per row (global RNG)."""
    if temperature < 1e-5:
        return torch.argmax(logits, dim=-1)
    b, t, v = logits.shape
    probs = torch.softmax(logits.view(-1, v) / temperature, dim=-1)
    return torch.multinomial(probs, num_samples=1).view(b, t)


def acceptance_length(block_ids: torch.Tensor, posterior: torch.Tensor) -> int:
    """model/dflash.py:258 — number of leading draft tokens equal to the target's."""
    return int((block_ids[:, 1:] == posterior[:, :-1]).cumprod(dim=1).sum(dim=1)[0].item())


def accept_commit_ref(output_ids, block_ids, posterior, start: int):
    """model/dflash.py:258-261 on plain python lists / 1-D int tensors (integer-only).
    Returns (acc, new_start); mutates output_ids."""
    bs = len(block_ids)
    acc = 0
    while acc < bs - 1 and int(block_ids[acc + 1]) == int(posterior[acc]):
        acc += 1
    for i in range(acc + 1):
        output_ids[start + i] = int(block_ids[i])
    output_ids[start + acc + 1] = int(posterior[acc])
    return acc, start + acc + 1


# --------------------------------------------------------------------------
# draft KV cache with the DynamicCache semantics the loop relies on
# --------------------------------------------------------------------------

class ListKVCache:
    """tf:cache_utils.py:127-188 (DynamicLayer.update / get_seq_length / crop with
    the legacy positive-argument 'absolute length' form the reference uses at
    model/dflash.py:246,262)."""

    def __init__(self):
        self.k: list[torch.Tensor] = []
        self.v: list[torch.Tensor] = []

    def update(self, k, v, layer: int):
        if layer == len(self.k):
            self.k.append(k)
            self.v.append(v)
        else:
            self.k[layer] = torch.cat([self.k[layer], k], dim=-2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=-2)
        return self.k[layer], self.v[layer]

    def get_seq_length(self) -> int:
        return 0 if not self.k else int(self.k[0].shape[-2])

    def crop(self, max_length: int):
        if max_length <= 0 or max_length >= self.get_seq_length():
            return
        self.k = [t[..., :max_length, :] for t in self.k]
        self.v = [t[..., :max_length, :] for t in self.v]


# --------------------------------------------------------------------------
# draft model (model/dflash.py:30-190)
# --------------------------------------------------------------------------

class DraftConfig(SimpleNamespace):
    """hidden_size, num_hidden_layers, num_attention_heads, num_key_value_heads,
    head_dim, intermediate_size, rms_norm_eps, rope_theta, block_size,
    num_target_layers, mask_token_id, target_layer_ids, attn_impl."""


def draft_attention(w: dict, i: int, cfg, hidden, target_hidden, cos, sin, cache: Optional[ListKVCache]):
    """Qwen3DFlashAttention.forward — model/dflash.py:58-102."""
    p = f"layers.{i}.self_attn."
    b, q_len = hidden.shape[:-1]
    ctx_len = target_hidden.shape[1]
    d = cfg.head_dim
    q = F.linear(hidden, w[p + "q_proj.weight"]).view(b, q_len, -1, d)
    q = rms_norm(q, w[p + "q_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    k_ctx = F.linear(target_hidden, w[p + "k_proj.weight"])
    k_noise = F.linear(hidden, w[p + "k_proj.weight"])
    v_ctx = F.linear(target_hidden, w[p + "v_proj.weight"])
    v_noise = F.linear(hidden, w[p + "v_proj.weight"])
    k = torch.cat([k_ctx, k_noise], dim=1).view(b, ctx_len + q_len, -1, d)
    v = torch.cat([v_ctx, v_noise], dim=1).view(b, ctx_len + q_len, -1, d)
    k = rms_norm(k, w[p + "k_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    v = v.transpose(1, 2)
    q, k = apply_rotary_dflash(q, k, cos, sin)
    if cache is not None:
        k, v = cache.update(k, v, i)
    o = attention(q, k, v, d ** -0.5, cfg.attn_impl)
    return F.linear(o, w[p + "o_proj.weight"])


def draft_layer(w: dict, i: int, cfg, hidden, target_hidden, cos, sin, cache):
    """Qwen3DFlashDecoderLayer.forward — model/dflash.py:113-145."""
    p = f"layers.{i}."
    res = hidden
    h = rms_norm(hidden, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
    h = draft_attention(w, i, cfg, h, target_hidden, cos, sin, cache)
    hidden = res + h
    res = hidden
    h = rms_norm(hidden, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
    h = swiglu_mlp(h, w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"], w[p + "mlp.down_proj.weight"])
    return res + h


def draft_forward(w: dict, cfg, *, position_ids, noise_embedding, target_hidden,
                  cache: Optional[ListKVCache] = None, trace: Optional[dict] = None):
    """DFlashDraftModel.forward — model/dflash.py:166-190.  Returns the final-normed
    hidden states [B, q_len, H] (NOT logits)."""
    hidden = noise_embedding
    ctx = rms_norm(F.linear(target_hidden, w["fc.weight"]), w["hidden_norm.weight"], cfg.rms_norm_eps)
    inv = rope_inv_freq(cfg.head_dim, cfg.rope_theta)
    cos, sin = rope_cos_sin(position_ids, inv, hidden.dtype)
    if trace is not None:
        trace["ctx"] = ctx
    for i in range(cfg.num_hidden_layers):
        hidden = draft_layer(w, i, cfg, hidden, ctx, cos, sin, cache)
        if trace is not None:
            trace[f"layer{i}"] = hidden
    return rms_norm(hidden, w["norm.weight"], cfg.rms_norm_eps)


# --------------------------------------------------------------------------
# decode loops
# --------------------------------------------------------------------------

def _trim(output_ids, max_length, mask_token_id, stop_token_ids, num_input_tokens):
    """model/dflash.py:269-275."""
    output_ids = output_ids[:, :max_length]
    output_ids = output_ids[:, output_ids[0] != mask_token_id]
    if stop_token_ids is not None:
        st = torch.tensor(stop_token_ids, device=output_ids.device)
        idx = torch.isin(output_ids[0][num_input_tokens:], st).nonzero(as_tuple=True)[0]
        if idx.numel() > 0:
            output_ids = output_ids[:, : num_input_tokens + idx[0] + 1]
    return output_ids


def _stop_hit(output_ids, num_input_tokens, stop_token_ids) -> bool:
    """model/dflash.py:265-268."""
    return stop_token_ids is not None and any(
        s in output_ids[:, num_input_tokens:] for s in stop_token_ids)


@torch.inference_mode()
def spec_generate(w: dict, cfg, target, input_ids, max_new_tokens: int,
                  stop_token_ids, temperature: float, record: Optional[list] = None):
    """DFlashDraftModel.spec_generate — model/dflash.py:192-277."""
    n_in = input_ids.shape[1]
    max_length = n_in + max_new_tokens
    bs = cfg.block_size
    output_ids = torch.full((1, max_length + bs), cfg.mask_token_id, dtype=torch.long)
    position_ids = torch.arange(output_ids.shape[1]).unsqueeze(0)
    tcache = target.new_cache()
    dcache = ListKVCache()
    out = target(input_ids, position_ids=position_ids[:, :n_in], past_key_values=tcache,
                 use_cache=True, logits_to_keep=1, output_hidden_states=True)
    output_ids[:, :n_in] = input_ids
    output_ids[:, n_in:n_in + 1] = sample(out.logits, temperature)
    target_hidden = extract_context_feature(out.hidden_states, cfg.target_layer_ids)
    acceptance_lengths = []
    start = n_in
    while start < max_length:
        block = output_ids[:, start:start + bs].clone()
        block_pos = position_ids[:, start:start + bs]
        noise = target.model.embed_tokens(block)
        hid = draft_forward(w, cfg, target_hidden=target_hidden, noise_embedding=noise,
                            position_ids=position_ids[:, dcache.get_seq_length(): start + bs],
                            cache=dcache)
        draft_logits = target.lm_head(hid[:, -bs + 1:, :])
        dcache.crop(start)
        block[:, 1:] = sample(draft_logits)
        out = target(block, position_ids=block_pos, past_key_values=tcache, use_cache=True,
                     output_hidden_states=True)
        posterior = sample(out.logits, temperature)
        acc = acceptance_length(block, posterior)
        output_ids[:, start:start + acc + 1] = block[:, :acc + 1]
        output_ids[:, start + acc + 1] = posterior[:, acc]
        if record is not None:
            record.append({"start": start, "block": block.clone(), "posterior": posterior.clone(),
                           "acc": acc, "draft_len": dcache.get_seq_length()})
        start += acc + 1
        tcache.crop(start)
        target_hidden = extract_context_feature(out.hidden_states, cfg.target_layer_ids)[:, :acc + 1, :]
        acceptance_lengths.append(acc + 1)
        if _stop_hit(output_ids, n_in, stop_token_ids):
            break
    return _trim(output_ids, max_length, cfg.mask_token_id, stop_token_ids, n_in), acceptance_lengths


@torch.inference_mode()
def dflash_generate(w: dict, cfg, target, input_ids, mask_token_id: int, max_new_tokens: int,
                    block_size: int, stop_token_ids, temperature: float = 0.0, draft_steps: int = 1):
    """benchmark.py:44-251 without the wall-clock/event fields: tail clamp
    (:104-105), block_size==1 pure-target baseline (:77,82,108,157), draft_steps
    multi-pass refinement with no draft cache (:112-142)."""
    n_in = input_ids.shape[1]
    max_length = n_in + max_new_tokens
    output_ids = torch.full((1, max_length + block_size), mask_token_id, dtype=torch.long)
    position_ids = torch.arange(output_ids.shape[1]).unsqueeze(0)
    tcache = target.new_cache()
    dcache = ListKVCache()
    out = target(input_ids, position_ids=position_ids[:, :n_in], past_key_values=tcache, use_cache=True,
                 logits_to_keep=1, output_hidden_states=block_size > 1)
    output_ids[:, :n_in] = input_ids
    output_ids[:, n_in:n_in + 1] = sample(out.logits, temperature)
    target_hidden = None
    if block_size > 1:
        target_hidden = extract_context_feature(out.hidden_states, cfg.target_layer_ids)
    start = n_in
    taus = []
    while start < max_length:
        ebs = min(block_size, max_length - start)
        block = output_ids[:, start:start + ebs].clone()
        block_pos = position_ids[:, start:start + ebs]
        if ebs > 1:
            use_cache = draft_steps == 1
            for _ in range(draft_steps):
                noise = target.model.embed_tokens(block)
                if use_cache:
                    dpos = position_ids[:, dcache.get_seq_length(): start + ebs]
                    dpast = dcache
                else:
                    ctx_len = int(target_hidden.shape[1])
                    dpos = position_ids[:, max(0, start - ctx_len): start + ebs]
                    dpast = None
                hid = draft_forward(w, cfg, target_hidden=target_hidden, noise_embedding=noise,
                                    position_ids=dpos, cache=dpast)
                block[:, 1:] = sample(target.lm_head(hid[:, -ebs + 1:, :]))
            if use_cache:
                dcache.crop(start)
        out = target(block, position_ids=block_pos, past_key_values=tcache, use_cache=True,
                     output_hidden_states=ebs > 1)
        posterior = sample(out.logits, temperature)
        acc = acceptance_length(block, posterior)
        output_ids[:, start:start + acc + 1] = block[:, :acc + 1]
        output_ids[:, start + acc + 1] = posterior[:, acc]
        taus.append(acc + 1)
        start += acc + 1
        tcache.crop(start)
        if ebs > 1:
            target_hidden = extract_context_feature(out.hidden_states, cfg.target_layer_ids)[:, :acc + 1, :]
        if _stop_hit(output_ids, n_in, stop_token_ids):
            break
    output_ids = _trim(output_ids, max_length, mask_token_id, stop_token_ids, n_in)
    return SimpleNamespace(output_ids=output_ids, num_input_tokens=n_in,
                           num_output_tokens=output_ids.shape[1] - n_in, acceptance_lengths=taus)


@torch.inference_mode()
def dflash_generate_policy(w: dict, cfg, target, input_ids, mask_token_id: int, max_new_tokens: int,
                           stop_token_ids, temperature: float, block_sizes: Sequence[int]):
    """benchmark_dynamic_schedule.py:260-434 with the per-cycle block size taken
    from a precomputed list `block_sizes` (the scheduler's choices are wall-clock
    dependent, so the oracle replays a recorded schedule).  Differences from
    dflash_generate: bs chosen per cycle then clamped to `remaining` (:321-323),
    draft tokens sampled WITH temperature (:342), l_gen from the first drafted
    stop token (:344-349), output buffer sized by max candidate (:276-285)."""
    max_bs = max(block_sizes)
    n_in = input_ids.shape[1]
    max_length = n_in + max_new_tokens
    output_ids = torch.full((1, max_length + max_bs), mask_token_id, dtype=torch.long)
    position_ids = torch.arange(output_ids.shape[1]).unsqueeze(0)
    tcache = target.new_cache()
    dcache = ListKVCache()
    out = target(input_ids, position_ids=position_ids[:, :n_in], past_key_values=tcache, use_cache=True,
                 logits_to_keep=1, output_hidden_states=max_bs > 1)
    output_ids[:, :n_in] = input_ids
    output_ids[:, n_in:n_in + 1] = sample(out.logits, temperature)
    target_hidden = extract_context_feature(out.hidden_states, cfg.target_layer_ids) if max_bs > 1 else None
    stop_t = torch.tensor(stop_token_ids) if stop_token_ids else None
    start = n_in
    taus, used, lgens = [], [], []
    cyc = 0
    while start < max_length:
        chosen = block_sizes[min(cyc, len(block_sizes) - 1)]
        bs = max(1, min(chosen, max_length - start))
        l_gen = float(bs)
        block = output_ids[:, start:start + bs].clone()
        block_pos = position_ids[:, start:start + bs]
        if bs > 1:
            noise = target.model.embed_tokens(block)
            hid = draft_forward(w, cfg, target_hidden=target_hidden, noise_embedding=noise,
                                position_ids=position_ids[:, dcache.get_seq_length(): start + bs],
                                cache=dcache)
            dcache.crop(start)
            block[:, 1:] = sample(target.lm_head(hid[:, -bs + 1:, :]), temperature)
            if stop_t is not None:
                pos = torch.isin(block[0, 1:], stop_t).nonzero(as_tuple=True)[0]
                if pos.numel() > 0:
                    l_gen = float(min(int(pos[0].item()) + 1, bs))
        out = target(block, position_ids=block_pos, past_key_values=tcache, use_cache=True,
                     output_hidden_states=max_bs > 1)
        posterior = sample(out.logits, temperature)
        acc = acceptance_length(block, posterior)
        tau = acc + 1
        output_ids[:, start:start + tau] = block[:, :tau]
        output_ids[:, start + tau] = posterior[:, acc]
        taus.append(tau)
        used.append(bs)
        lgens.append(l_gen)
        start += tau
        tcache.crop(start)
        if max_bs > 1:
            target_hidden = extract_context_feature(out.hidden_states, cfg.target_layer_ids)[:, :tau, :]
        cyc += 1
        if _stop_hit(output_ids, n_in, stop_token_ids):
            break
    output_ids = _trim(output_ids, max_length, mask_token_id, stop_token_ids, n_in)
    return SimpleNamespace(output_ids=output_ids, num_input_tokens=n_in,
                           num_output_tokens=output_ids.shape[1] - n_in,
                           acceptance_lengths=taus, used_block_sizes=used, l_gen=lgens)
