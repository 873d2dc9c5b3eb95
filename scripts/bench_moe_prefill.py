#!/usr/bin/env python3
"""GPU box: prefill of a Qwen3-Coder-30B-A3B-SHAPED sparse-MoE target (BASELINE configs[4]; H 2048, 32/4 heads, 128
experts, top-8, moe_intermediate 768; random-init weights, LAYERS of the 48 layers) on the kernels (rows sorted by expert,
grouped MFMA GEMMs: csrc/prefill.hip) vs through the HF forward (model/dflash.py:218-225), same box, same weights.
SPREAD = 1 replaces the router logits by random ones for the timing (random-init hidden rows are nearly parallel, so their
own routing piles every row on the same few experts; a trained, load-balanced router gives each expert ~P*8/128 rows).
usage: bench_moe_prefill.py [layers=8] [P=1024] [spread=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from transformers import Qwen3MoeConfig, Qwen3MoeForCausalLM

from dflash_amd import NativeTarget, ops

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spread = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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
prompt = torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(1)).to(dev)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


nt = NativeTarget(hf)
assert nt.native_prefill and nt.is_moe
cache = nt.new_cache(P + 64)
if spread:
    rnd = (torch.randn(ops.prefill_rows_padded(P), 128, generator=torch.Generator().manual_seed(2)) * 3).to(torch.bfloat16).to(dev)
    real = ops.prefill_gemm_rows

    def patched(wp, xf, p, n, k, out):
        if n == 128 and k == 2048:     # the router GEMM: random logits instead (timing only)
            out.copy_(rnd)
        else:
            real(wp, xf, p, n, k, out)
    ops.prefill_gemm_rows = patched
ms = timed(lambda: nt.prefill(prompt, cache, output_hidden_states=True, tap_layers=[1]))
sc = nt._pf_moe
cnt = sc["cnt"].cpu()
print(f"MoE prefill P={P} layers={layers} native: {ms:7.2f} ms ({ms / layers * 1e3:.0f} us per layer); last layer's routing: "
      f"{int((cnt > 0).sum())} experts used, rows per expert min {int(cnt.min())} max {int(cnt.max())}, "
      f"{int(sc['n_items'][0])} work items / {int(sc['n_items'][1])} tiles", flush=True)
if spread:
    ops.prefill_gemm_rows = real
nt2 = NativeTarget(hf, prefill="hf")
c2 = nt2.new_cache(P + 64)
ms2 = timed(lambda: nt2.prefill(prompt, c2, output_hidden_states=True), n=3)
print(f"MoE prefill P={P} layers={layers} through the HF forward (its own routing): {ms2:7.2f} ms ({ms2 / layers * 1e3:.0f} us per layer)")
