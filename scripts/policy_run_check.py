"""GPU box: the dynamic block-size loop (benchmark_dynamic_schedule.py's dflash_generate_policy, EWMA scheduler over
{8, 12, 16}; BASELINE configs[4]'s schedule) on the Qwen3-8B shapes with scripted acceptance: prints TPOT, the block
sizes used and checks the committed ids against the target's closed-form greedy walk."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from dflash_amd import EWMAPerformanceScheduler, NativeTarget, dflash_generate_policy
from dflash_amd.harness_cli import _synthetic, _tau_hook
from dflash_amd.synthetic import greedy_walk

dev = torch.device("cuda", 0)
P, N = 1024, int(os.environ.get("N", "768"))
target, draft, perm = _synthetic("qwen3-8b", int(os.environ.get("LAYERS", "36")), dev)
nt = NativeTarget(target)
prompt = torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(3)).to(dev)
G = greedy_walk(perm, prompt, N + 64).to(dev)
hook = _tau_hook(perm, prompt, N, 16, 7.3, 11, 151000)
sched = EWMAPerformanceScheduler(candidates=[8, 12, 16], scheduler_mode="ewma", warmup_cycles=6, ewma_alpha=0.25,
                                 switch_margin=0.03, required_streak=2, cooldown_cycles=2, probe_interval=5,
                                 low_accept_threshold=0.2, low_accept_streak=3, adl_rho=0.3, adl_delta=1.0, adl_k_min=8,
                                 adl_k_max=16, adl_neighborhood=4)
r = dflash_generate_policy(model=draft, target=nt, input_ids=prompt, mask_token_id=draft.config.mask_token_id,
                           max_new_tokens=N, stop_token_ids=None, temperature=0.0, scheduler=sched, draft_token_hook=hook)
out = r.output_ids[0]
ok = bool((out == G[:out.numel()]).all())
print(f"policy loop, candidates {{8, 12, 16}}: {out.numel() - P} new tokens in {len(r.acceptance_lengths)} cycles, "
      f"TPOT {1e3 * r.time_per_output_token:.3f} ms, mean tau {sum(r.acceptance_lengths) / len(r.acceptance_lengths):.2f}, "
      f"block sizes used {dict(sorted(Counter(r.used_block_sizes).items()))}, ids == greedy walk: {ok}")
sys.exit(0 if ok else 1)
