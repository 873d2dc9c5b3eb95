"""GPU box: a long single-request generation on the Qwen3-8B shapes (prompt 4096, 2048 new tokens: the prefix grows
from 4k to 6k keys, across the switch to two query heads per attention workgroup) with scripted acceptance; the
committed ids must be the target's closed-form greedy walk for the whole run.  Prints cycles, tokens/s and the check."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from dflash_amd import NativeTarget, dflash_generate
from dflash_amd.harness_cli import _synthetic, _tau_hook
from dflash_amd.synthetic import greedy_walk

dev = torch.device("cuda", 0)
P, N, bs = int(os.environ.get("P", "4096")), int(os.environ.get("N", "2048")), 16
target, draft, perm = _synthetic("qwen3-8b", int(os.environ.get("LAYERS", "36")), dev)
nt = NativeTarget(target)
prompt = torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(3)).to(dev)
G = greedy_walk(perm, prompt, N + 64).to(dev)
hook = _tau_hook(perm, prompt, N, bs, 7.3, 11, 151000)
torch.cuda.synchronize()
t0 = time.perf_counter()
r = dflash_generate(draft, nt, prompt, draft.config.mask_token_id, N, bs, None, 0.0, draft_token_hook=hook)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = r.output_ids[0]
ok = bool((out == G[:out.numel()]).all())
print(f"prompt {P}, {out.numel() - P} new tokens in {len(r.acceptance_lengths)} cycles, {dt:.2f} s wall (prefill included), "
      f"TPOT {1e3 * r.time_per_output_token:.3f} ms, mean tau {sum(r.acceptance_lengths) / len(r.acceptance_lengths):.2f}, "
      f"ids == greedy walk: {ok}")
sys.exit(0 if ok and out.numel() == P + N else 1)
