"""`python -m dflash_amd.harness_cli` — the driver of the harness wire format:
what `benchmark.py:301-623` does around `dflash_generate`, on this package's loop.

Per prompt it runs the bs = 1 baseline and the speculative run (benchmark.py:446-470), shards the
prompts `range(rank, n, world)` (:445), gathers to rank 0 (:536-551), prints the stat lines
`run_block_sweep.sh:199-212` greps (:553-604) and writes the per-sample JSONL (:503-534, :606-612)
and the per-cycle trace JSONL (:481-497, :614-620).

Offline there are no checkpoints, tokenizers or datasets, so the inputs are either
HF checkpoint directories (`--model-name-or-path DIR --draft-name-or-path DIR`, both
`local_files_only`) or `--synthetic {tiny,qwen3-4b,qwen3-8b}`: seeded random-init weights of that
architecture with a large-margin greedy rule imposed on the target (dflash_amd.synthetic), prompts
of `--prompt-len` seeded random ids, and — because random draft weights never agree with the target —
an optional scripted acceptance (`--scripted-tau`).  Without a tokenizer `output_text` is the
space-joined ids.
"""
from __future__ import annotations

import argparse
import atexit
import random
import sys
import time
from itertools import chain

import numpy as np
import torch

from . import distributed as dist
from . import harness
from .generate import cuda_time, dflash_generate

SYNTHETIC = {
    "tiny": (dict(vocab_size=2048, hidden_size=512, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=128,
                  intermediate_size=1024, rope_theta=1e6),
             dict(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                  intermediate_size=1024, vocab_size=2048, num_target_layers=6, block_size=16, rope_theta=1e6,
                  mask_token_id=2047)),
}


def _synthetic(name: str, target_layers, dev):
    from .config import (DFlashConfig, QWEN3_4B_DRAFT, QWEN3_4B_TARGET, QWEN3_8B_DRAFT, QWEN3_8B_TARGET)
    from .model import DFlashDraftModel
    from .synthetic import impose_greedy_walk, make_hf_qwen3
    table = dict(SYNTHETIC)
    table["qwen3-8b"] = (QWEN3_8B_TARGET, QWEN3_8B_DRAFT)
    table["qwen3-4b"] = (QWEN3_4B_TARGET, QWEN3_4B_DRAFT)
    tdims, ddims = table[name]
    if target_layers:
        tdims = {**tdims, "num_layers": target_layers}
        ddims = {**ddims, "num_target_layers": target_layers}
    target = make_hf_qwen3(tdims, dev)
    perm = impose_greedy_walk(target, seed=1234)
    cfg = DFlashConfig(**ddims)
    draft = DFlashDraftModel(cfg, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    sd = {k: (torch.randn(s, generator=g, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
          if len(s) == 2 else torch.ones(s, device=dev, dtype=torch.bfloat16)
          for k, s in cfg.state_dict_shapes().items()}
    draft.load_state_dict(sd)
    return target, draft, perm


def _tau_hook(perm, prompt, n_new, bs, mean_tau, seed, vocab):
    """Scripted acceptance: after the timed draft forward + argmax the draft tokens are
    overwritten with k tokens of the target's greedy walk followed by a wrong id, k + 1 ~ truncated
    geometric with the requested mean."""
    from .synthetic import greedy_walk
    G = greedy_walk(perm, prompt, n_new + 4 * bs).to(prompt.device)
    lo, hi = 0.0, 1.0
    for _ in range(60):
        p = 0.5 * (lo + hi)
        m = 1.0 + sum(p ** j for j in range(1, bs))
        lo, hi = (p, hi) if m < mean_tau else (lo, p)
    gen = torch.Generator().manual_seed(seed)
    plan = (torch.rand(n_new + 8, bs - 1, generator=gen) < p).long().cumprod(dim=1).sum(dim=1).tolist()

    def hook(blk, start, call):
        k = min(plan[call % len(plan)], blk.shape[1] - 1)
        if k:
            blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            wrong = G[start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == wrong, (wrong + 1) % (vocab - 64), blk[0, k + 1])

    return hook


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m dflash_amd.harness_cli")
    ap.add_argument("--model-name-or-path", type=str, default=None)       # benchmark.py:303
    ap.add_argument("--draft-name-or-path", type=str, default=None)       # :304
    ap.add_argument("--synthetic", choices=["tiny", "qwen3-4b", "qwen3-8b"], default=None)
    ap.add_argument("--target-layers", type=int, default=None, help="synthetic: fewer target layers (quick runs)")
    ap.add_argument("--block-size", type=int, default=None)               # :305
    ap.add_argument("--dataset", type=str, default="synthetic")           # :306 (only `synthetic` exists offline)
    ap.add_argument("--max-samples", type=int, default=4)                 # :307
    ap.add_argument("--max-new-tokens", type=int, default=128)            # :308
    ap.add_argument("--temperature", type=float, default=0.0)             # :309
    ap.add_argument("--skip-baseline", action="store_true")               # :310-314
    ap.add_argument("--local-files-only", action="store_true")            # :315-319 (always true offline)
    ap.add_argument("--save-outputs-path", type=str, default=None)        # :325-330
    ap.add_argument("--save-cycle-trace-path", type=str, default=None)    # :331-336
    ap.add_argument("--collect-profile", action="store_true")             # :337-341
    ap.add_argument("--draft-steps", type=int, default=1)                 # :342-347
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--scripted-tau", type=float, default=None,
                    help="synthetic only: mean acceptance length to script (random draft weights give tau = 1)")
    ap.add_argument("--hf-verify", action="store_true", help="verify through the HF forward instead of NativeTarget")
    args = ap.parse_args(argv)
    if args.draft_steps < 1:
        raise ValueError("--draft-steps must be >= 1")
    if args.dataset != "synthetic":
        raise SystemExit(f"dataset `{args.dataset}` needs a download; offline only `synthetic` exists (SURVEY.md §8c)")
    if (args.synthetic is None) == (args.model_name_or_path is None):
        raise SystemExit("give either --synthetic NAME or --model-name-or-path DIR --draft-name-or-path DIR")
    collect_profile = args.collect_profile or (args.save_cycle_trace_path is not None)
    t0 = time.perf_counter()

    def setup_log(msg):
        if dist.is_main():
            print(f"[setup][rank{dist.rank()}] +{time.perf_counter() - t0:.2f}s {msg}", flush=True)

    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    if not torch.cuda.is_available():
        raise SystemExit("harness_cli needs a GPU: the product has no CPU path")
    torch.cuda.manual_seed_all(0)
    dist.init()
    atexit.register(dist.destroy)
    torch.cuda.set_device(dist.local_rank())
    dev = torch.device("cuda", dist.local_rank())
    setup_log(f"distributed ready (world_size={dist.size()}, device={dev})")

    perm = None
    if args.synthetic:
        target, draft, perm = _synthetic(args.synthetic, args.target_layers, dev)
        vocab = target.config.vocab_size
        stop_ids = None
    else:
        from transformers import AutoModelForCausalLM
        from .model import DFlashDraftModel
        target = AutoModelForCausalLM.from_pretrained(args.model_name_or_path, dtype=torch.bfloat16,
                                                      local_files_only=True).to(dev).eval()
        draft = DFlashDraftModel.from_pretrained(args.draft_name_or_path, device=dev)
        vocab = target.config.vocab_size
        eos = getattr(target.config, "eos_token_id", None)
        stop_ids = None if eos is None else ([eos] if isinstance(eos, int) else list(eos))
    hf_target = target
    if not args.hf_verify:
        from .target import NativeTarget
        target = NativeTarget(hf_target)
    block_size = args.block_size if args.block_size is not None else draft.block_size
    setup_log(f"models ready; effective block_size={block_size}")

    n = args.max_samples
    responses, output_records, trace_records = [], [], []
    baseline_enabled = not args.skip_baseline
    for idx in dist.shard_indices(n):
        ids = torch.randint(0, vocab - 64, (1, args.prompt_len), generator=torch.Generator().manual_seed(1000 + idx))
        input_ids = ids.to(dev)
        response = {}
        for bs in dict.fromkeys([block_size] if args.skip_baseline else [1, block_size]):
            hook = None
            if bs > 1 and perm is not None and args.scripted_tau:
                hook = _tau_hook(perm, ids, args.max_new_tokens, bs, args.scripted_tau, 7 + idx, vocab)
            t_call = cuda_time()
            response[bs] = dflash_generate(model=draft, target=target, input_ids=input_ids,
                                           mask_token_id=draft.mask_token_id, max_new_tokens=args.max_new_tokens,
                                           block_size=bs, stop_token_ids=stop_ids, temperature=args.temperature,
                                           collect_profile=collect_profile, draft_steps=args.draft_steps,
                                           draft_token_hook=hook)
            response[bs].wall_time_s = cuda_time() - t_call
            response[bs].output_ids = response[bs].output_ids.cpu()      # gather_object pickles these
        text = {bs: " ".join(str(t) for t in r.output_ids[0, r.num_input_tokens:].tolist())
                for bs, r in response.items()}
        responses.append(response)
        if args.save_cycle_trace_path:
            for mode, mbs in (("baseline", 1), ("speculative", block_size)):
                if mbs in response:
                    trace_records += list(harness.cycle_trace_records(
                        response[mbs], rank=dist.rank(), dataset=args.dataset, dataset_row_idx=idx, turn_index=0,
                        mode=mode, block_size=mbs))
        prompt_text = " ".join(str(t) for t in ids[0].tolist())
        output_records.append(harness.output_record(
            rank=dist.rank(), dataset_row_idx=idx, turn_index=0, dataset=args.dataset, prompt_text=prompt_text,
            input_text=prompt_text, block_size=block_size, draft_steps=args.draft_steps,
            speculative=response[block_size], speculative_text=text[block_size],
            baseline=response.get(1) if baseline_enabled else None, baseline_text=text.get(1)))

    if dist.size() > 1:
        responses = dist.gather(responses)
        output_records = dist.gather(output_records)
        trace_records = dist.gather(trace_records)
        if not dist.is_main():
            return 0
        responses, output_records = list(chain(*responses)), list(chain(*output_records))
        trace_records = list(chain(*trace_records))

    for line in harness.stat_lines(responses, block_size, draft_steps=args.draft_steps, baseline=baseline_enabled,
                                   collect_profile=collect_profile, gpu_name=torch.cuda.get_device_name(dev),
                                   runtime_version=torch.version.cuda, torch_version=torch.__version__,
                                   world_size=dist.size()):
        print(line)
    if args.save_outputs_path:
        harness.write_jsonl(args.save_outputs_path, output_records)
        print(f"Saved per-sample outputs to: {args.save_outputs_path}")
    if args.save_cycle_trace_path:
        harness.write_jsonl(args.save_cycle_trace_path, trace_records)
        print(f"Saved per-cycle trace to: {args.save_cycle_trace_path}")
    sys.stdout.flush()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
