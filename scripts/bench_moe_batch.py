#!/usr/bin/env python3
"""GPU box: the 16-token verify of FOUR requests on a Qwen3-Coder-30B-A3B-shaped sparse-MoE target (LAYERS of its 48 layers,
prefix 1024) through BatchedDecoder.verify: attention and dense projections batched; the expert MLP as one pass per request
tile (round 3, first form) against ONE shared pass over the experts (NativeTarget._moe_mlp_shared).  Random-init weights;
the routers scaled so that a tile's 128 slots spread over ~80 experts.
usage: bench_moe_batch.py [layers=8] [prefix=1024]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import Qwen3MoeConfig, Qwen3MoeForCausalLM

from dflash_amd import DFlashDraftModel, NativeTarget
from dflash_amd.batch import BatchedDecoder
from dflash_amd.config import DFlashConfig
from dflash_amd.synthetic import make_draft_state_dict

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
cfg = Qwen3MoeConfig(vocab_size=151936, hidden_size=2048, intermediate_size=6144, moe_intermediate_size=768,
                     num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=4, head_dim=128, num_experts=128,
                     num_experts_per_tok=8, decoder_sparse_step=1, norm_topk_prob=True, max_position_embeddings=40960,
                     rms_norm_eps=1e-6, tie_word_embeddings=False, rope_parameters={"rope_type": "default", "rope_theta": 1e7},
                     mlp_only_layers=[])
cfg._attn_implementation = "sdpa"
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    hf = Qwen3MoeForCausalLM(cfg).eval()
torch.set_default_dtype(torch.float32)
nt = NativeTarget(hf, keep_hf=False)
dcfg = DFlashConfig(hidden_size=2048, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=4, head_dim=128,
                    intermediate_size=6144, vocab_size=151936, num_target_layers=layers, block_size=16, rope_theta=1e7,
                    mask_token_id=151669)
m = DFlashDraftModel(dcfg, device=dev)
m.load_state_dict(make_draft_state_dict(dcfg, seed=3, dtype=torch.bfloat16))
R = 4
dec = BatchedDecoder(m, nt, R, max_rows=P + 256, out_len=P + 256, mask_token_id=dcfg.mask_token_id)
for r in range(R):
    dec.admit(r, torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(1 + r)).to(dev))
dec.block[:, 1:] = torch.randint(0, 151000, (dec.block.shape[0], 15), generator=torch.Generator().manual_seed(9)).to(dev)


def timed(shared, n=10):
    nt.moe_shared_pass = shared
    for _ in range(3):
        dec.verify()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dec.verify()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b = timed(False), timed(True)
print(f"MoE target, {layers} layers, 4 requests x 16 rows, prefix {P}: verify {a:.3f} ms with a pass per request tile "
      f"({(a - 0.25) / layers * 1e3:.0f} us per layer), {b:.3f} ms with one shared pass over the experts "
      f"({(b - 0.25) / layers * 1e3:.0f} us per layer)")
