#!/usr/bin/env python3
"""GPU box: the 16-token verify of a Qwen3-Coder-30B-A3B-SHAPED sparse-MoE target (BASELINE configs[4]; H 2048, 32/4
heads, 128 experts, top-8, moe_intermediate 768, V 151936; random-init weights, LAYERS of the 48 layers) on the native
kernels vs through the HF forward (the reference's verify, model/dflash.py:249-255).  Prints one JSON line.
The router weights are scaled by GAIN: 8 concentrates a block's 128 routing slots on ~14 experts, 1 (random init as it
is) spreads them over ~80, the spread a trained router is balanced for.
SPREAD = 1 replaces the routing by a synthetic balanced one for the timing (random-init hidden rows are nearly parallel,
so their own routing never spreads): every row takes 8 experts drawn at random, weight 1/8 each — what a trained,
load-balanced router does to a 16-token block (~80 of the 128 experts active).  Numbers only, no numerics.
usage: bench_moe_verify.py [layers=8] [prefix=1024] [gain=8] [spread=0]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from transformers import DynamicCache, Qwen3MoeConfig, Qwen3MoeForCausalLM

from dflash_amd import NativeTarget

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
gain = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
spread = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
cfg = Qwen3MoeConfig(vocab_size=151936, hidden_size=2048, intermediate_size=6144, moe_intermediate_size=768,
                     num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=4, head_dim=128,
                     num_experts=128, num_experts_per_tok=8, decoder_sparse_step=1, norm_topk_prob=True,
                     max_position_embeddings=40960, rms_norm_eps=1e-6, tie_word_embeddings=False,
                     rope_parameters={"rope_type": "default", "rope_theta": 1e7}, mlp_only_layers=[])
cfg._attn_implementation = "sdpa"
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    hf = Qwen3MoeForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
with torch.no_grad():
    for layer in hf.model.layers:
        layer.mlp.gate.weight.mul_(gain)
nt = NativeTarget(hf)
nt.moe_pair_kernel = os.environ.get("DFL_MOE_PAIR", "1") != "0"   # A/B: dfl_moe_gate_up vs the general expert GEMM
g = torch.Generator().manual_seed(1)
prompt = torch.randint(0, 151000, (1, P), generator=g).to(dev)
block = torch.randint(0, 151000, (1, 16), generator=g).to(dev)
cache = nt.new_cache(P + 64)
nt.prefill(prompt, cache)

if spread:
    from dflash_amd import ops
    gs = torch.Generator().manual_seed(5)
    wt = torch.zeros(16, 128)
    for r in range(16):
        wt[r, torch.randperm(128, generator=gs)[:8]] = 0.125
    act = (wt.sum(0) > 0)
    lst = torch.nonzero(act).flatten().to(torch.int32)
    nt.ws["wt"][0].copy_(wt.to(torch.bfloat16))
    nt.ws["active"].copy_(act.to(torch.int32))
    nt.ws["elist"][:lst.numel()].copy_(lst)
    nt.ws["n_active"].fill_(int(lst.numel()))
    ops.moe_route = lambda *a, **k: None   # the buffers above stay as they are (the route launch, ~3 us, is not timed)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def native():
    cache.crop(P)
    nt.verify(block[0], P, cache)


rc = DynamicCache()
with torch.inference_mode():
    hf(prompt, past_key_values=rc, use_cache=True)


def through_hf():
    with torch.inference_mode():
        hf(block, position_ids=torch.arange(P, P + 16, device=dev)[None], past_key_values=rc, use_cache=True,
           output_hidden_states=True)
        rc.crop(P)


ms_n, ms_h = timed(native), timed(through_hf, 3)
n_act = int(nt.ws["n_active"])
per_layer_bytes = (n_act * 3 * 768 * 2048 + (32 + 2 * 4) * 128 * 2048 + 32 * 128 * 2048 + 128 * 2048) * 2
print(json.dumps({"workload": f"Qwen3-Coder-30B-A3B-shaped MoE target, {layers} of 48 layers, 16-token verify, prefix {P}",
                  "router_gain": gain, "synthetic_balanced_routing": bool(spread), "native_verify_ms": ms_n, "hf_verify_ms": ms_h, "speedup": ms_h / ms_n,
                  "active_experts_last_layer": n_act, "approx_weight_bytes_per_layer": per_layer_bytes,
                  "native_ms_per_layer": (ms_n - 0.19) / layers,
                  "approx_layer_GBps": per_layer_bytes / ((ms_n - 0.19) / layers * 1e-3) / 1e9}))
