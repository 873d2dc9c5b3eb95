#!/usr/bin/env python3
"""GPU box: the sparse-MoE MLP of ONE 30B-A3B-shaped layer (H 2048, 128 experts of 768, top-8) for the four 16-row tiles of
a ragged batch / a candidate pass: per-tile expert passes (NativeTarget.moe_mlp_tiles, round 3 first form) against the one
shared pass over the experts (the prefill's grouped kernels).  Random rows, the layer's own random-init router scaled so
that a tile's 128 slots spread over ~80 experts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import Qwen3MoeConfig, Qwen3MoeForCausalLM

from dflash_amd import NativeTarget, ops

dev = torch.device("cuda", 0)
cfg = Qwen3MoeConfig(vocab_size=2048, hidden_size=2048, intermediate_size=6144, moe_intermediate_size=768, num_hidden_layers=2,
                     num_attention_heads=32, num_key_value_heads=4, head_dim=128, num_experts=128, num_experts_per_tok=8,
                     decoder_sparse_step=1, norm_topk_prob=True, max_position_embeddings=4096, rms_norm_eps=1e-6,
                     tie_word_embeddings=False, rope_parameters={"rope_type": "default", "rope_theta": 1e7}, mlp_only_layers=[])
cfg._attn_implementation = "sdpa"
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    hf = Qwen3MoeForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
with torch.no_grad():
    for layer in hf.model.layers:
        layer.mlp.gate.weight.mul_(40.0)    # random rows against a random router: spread the routing
nt = NativeTarget(hf)
MT, H = 4, 2048
rows = (torch.randn(MT * 16, H, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16).to(dev)
xn = torch.zeros(MT, 16 * H, dtype=torch.bfloat16, device=dev)
for t in range(MT):
    ops.pack_rows(rows[16 * t:16 * t + 16], 16, xn[t])
dyn = torch.tensor([[0, 0, 16, 0, 0, 0, 0, 0]] * MT, dtype=torch.int32, device=dev)
part = torch.zeros(2 * MT * 16 * H, dtype=torch.float32, device=dev)
lw = nt.layers[0]


def timed(shared, R, n=30):
    nt.moe_shared_pass = shared
    for _ in range(3):
        nt.moe_mlp_tiles(lw, R, MT, dyn, xn, part)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        nt.moe_mlp_tiles(lw, R, MT, dyn, xn, part)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


nt.moe_shared_min = 2
for R in (2, 3, 4):
    a, b = timed(False, R), timed(True, R)
    used = int((nt._moe_sh["sc"]["cnt"] > 0).sum()) if nt._moe_sh else -1
    print(f"R = {R} tiles: per-tile expert passes {a:7.1f} us, one shared pass {b:7.1f} us ({used} experts used)")
