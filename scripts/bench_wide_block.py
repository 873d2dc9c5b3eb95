"""GPU box: time one draft forward (+ lm_head) and one native verify of a wide block (17..32 rows) on the Qwen3-8B
shapes, one pass over the weights (ragged-batch GEMMs, R = 2) against two passes (single-request GEMMs per 16-row
tile), next to the 16-row block.  usage: bench_wide_block.py [target_layers]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from dflash_amd import NativeTarget, ops
from dflash_amd.harness_cli import _synthetic

dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 36
target, draft, _ = _synthetic("qwen3-8b", L, dev)
nt = NativeTarget(target)
P = 1024
lm = ops.pack_weight(target.lm_head.weight.detach().to(torch.bfloat16).contiguous())
emb = target.model.embed_tokens.weight.detach().to(torch.bfloat16).contiguous()
ids = torch.randint(0, 1000, (1, P + 64), device=dev)


def timed(fn, n=12):
    for _ in range(3):   # one-off work of the first calls (weight packing, code-object loads, lazy workspaces)
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.inference_mode():
    tc = nt.new_cache(P + 128)
    nt.prefill(ids[:, :P], tc)
    dc = draft.new_cache(P + 128)
    draft.prefill_context(dc, torch.randn(P, draft.config.fc_in, device=dev).to(torch.bfloat16), 0)
    th = torch.randn(16, draft.config.fc_in, device=dev).to(torch.bfloat16)
    for bs in (16, 20, 24, 32):
        for one in ((True,) if bs <= 16 else (True, False)):
            nt.wide_one_pass = draft.wide_one_pass = one
            blk = ids[0, P:P + bs].clone()

            def v():
                tc.length = P
                nt.verify(blk, P, tc, tap_layers=list(draft.target_layer_ids))

            def d():
                dc.length = P
                rows = draft.draft_block(dc, th_rows=th[:8], tau=8, bs=bs, pos0=P, block_ids=blk, embed=emb, append=False)
                draft.draft_tokens(rows, lm, bs, blk)

            print(f"bs {bs:2d} {'one pass ' if one else 'two passes'}: draft+lm_head {timed(d):6.3f} ms   verify {timed(v):6.3f} ms", flush=True)
