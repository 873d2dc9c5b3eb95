"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference and transformers).  The
reference's code is imported, never copied: what is stored is inputs and the
outputs the reference computed for them.  `benchmark.py` /
`benchmark_dynamic_schedule.py` import `loguru`, `rich`, `tqdm` at module top
(absent here) — those three are replaced by empty stand-in modules before the
import; nothing on the measured path uses them.

    python tests/golden/make_golden.py

Fixtures (SURVEY.md §8c list):
  G1 draft_forward_<cfg>.npz   multi-cycle DFlashDraftModel.forward with cache
  G2 argmax.npz                sample(logits, 0) incl. engineered exact ties
  G3 accept.json               acceptance scan / commit on synthetic id pairs
  G4 e2e_<name>.json           spec_generate / dflash_generate / _policy ids
  G5 scheduler.json            EWMAPerformanceScheduler decision traces
  G6 sample_t.npz              sample(logits, 0.7) under torch.manual_seed
  G7 harness.json              summarize_mode / summarize_profile of benchmark.py
  G9 candidates.json           candidate builders + budget rule of benchmark_candidate_solutions.py
  G8 draft_forward_*_wide.npz, e2e_wide.json   block sizes 17..32 (forward vectors; loop runs at 20 / 24 / 32, policy)
"""
import json
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch

import importlib.util  # noqa: E402

for name in ("loguru", "rich", "tqdm"):
    if importlib.util.find_spec(name) is None:
        m = types.ModuleType(name)
        m.logger = types.SimpleNamespace(warning=print, info=print)
        m.print = print
        m.tqdm = lambda x, **k: x
        sys.modules[name] = m

from transformers import Qwen3Config  # noqa: E402
from model import DFlashDraftModel, sample as ref_sample  # noqa: E402  (reference)
import benchmark as ref_bench  # noqa: E402  (reference)
import benchmark_dynamic_schedule as ref_dyn  # noqa: E402  (reference)

import helpers as H  # noqa: E402
from dflash_amd.config import DFlashConfig  # noqa: E402

torch.set_num_threads(8)


def ref_draft(cfg: DFlashConfig, sd: dict, dtype, attn_impl: str) -> DFlashDraftModel:
    hf = Qwen3Config(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                     head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, max_position_embeddings=cfg.max_position_embeddings,
                     rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
                     attention_bias=False, tie_word_embeddings=False)
    hf.block_size = cfg.block_size
    hf.num_target_layers = cfg.num_target_layers
    hf.dflash_config = {"mask_token_id": cfg.mask_token_id, "target_layer_ids": list(cfg.target_layer_ids)}
    hf._attn_implementation = attn_impl
    # Instantiate under the default dtype, as `from_pretrained(dtype=...)` does
    # (benchmark.py:411-416): parameters come out in `dtype` while the rotary
    # `inv_freq` buffer stays fp32.  A blanket `.to(bf16)` would round inv_freq.
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        m = DFlashDraftModel(hf)
    finally:
        torch.set_default_dtype(prev)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    assert m.rotary_emb.inv_freq.dtype == torch.float32 and m.fc.weight.dtype == dtype
    return m.eval()


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


# ---------------------------------------------------------------- G1
def gen_draft_forward(tag, cfg, dtype, attn_impl, prompt_len, steps, seed):
    from transformers import DynamicCache
    sd = H.draft_weights(cfg, dtype=dtype)
    model = ref_draft(cfg, sd, dtype, attn_impl)
    g = torch.Generator().manual_seed(seed)
    cache = DynamicCache()
    out = {"prompt_len": prompt_len, "steps": np.array(steps), "dtype": str(dtype), "attn_impl": attn_impl,
           "w_checksum": float(sum(v.float().sum() for v in sd.values()))}
    start = prompt_len
    ctx_rows = prompt_len
    with torch.inference_mode():
        for c, (bs, tau_next) in enumerate(steps):
            th = (torch.randn(1, ctx_rows, cfg.fc_in, generator=g) * 1.5).to(dtype)
            ne = (torch.randn(1, bs, cfg.hidden_size, generator=g) * 0.05).to(dtype)
            pos = torch.arange(cache.get_seq_length(), start + bs).unsqueeze(0)
            hid = model(target_hidden=th, noise_embedding=ne, position_ids=pos, past_key_values=cache,
                        use_cache=True, is_causal=False)
            cache.crop(start)
            out[f"th{c}"], out[f"ne{c}"], out[f"hid{c}"] = f32(th), f32(ne), f32(hid)
            out[f"start{c}"] = start
            start += tau_next
            ctx_rows = tau_next
        # final cache content of layer 0 / last layer (post-norm, post-RoPE K; V)
        for li in (0, cfg.num_hidden_layers - 1):
            out[f"k_l{li}"] = f32(cache.layers[li].keys)
            out[f"v_l{li}"] = f32(cache.layers[li].values)
    np.savez_compressed(os.path.join(HERE, f"draft_forward_{tag}.npz"), **out)
    print("G1", tag, "done")


# ---------------------------------------------------------------- G2
def gen_argmax():
    g = torch.Generator().manual_seed(11)
    logits = (torch.randn(1, 15, 4096 + 48, generator=g) * 2).to(torch.bfloat16)
    # engineered exact ties: the row maximum repeated at several columns; first index must win
    for r in range(15):
        mx = logits[0, r].max()
        cols = torch.randint(0, logits.shape[2], (3,), generator=g)
        logits[0, r, cols] = mx
    logits[0, 3, :] = 0.5                      # whole row tied -> index 0
    logits[0, 4, -1] = 100.0                   # max in the last column
    ids = ref_sample(logits, 0.0)
    lf = torch.randn(1, 7, 1000, generator=g)  # fp32 logits path
    lf[0, 2, 10] = lf[0, 2, 500] = 9.0
    np.savez_compressed(os.path.join(HERE, "argmax.npz"), logits_bf16=f32(logits), ids_bf16=ids.numpy(),
                        logits_f32=lf.numpy(), ids_f32=ref_sample(lf, 0.0).numpy())
    print("G2 done")


# ---------------------------------------------------------------- G3
def gen_accept():
    g = torch.Generator().manual_seed(5)
    cases = []
    for bs in (2, 8, 12, 16):
        for acc_want in sorted({0, 1, bs // 2, bs - 2, bs - 1} & set(range(bs))):
            block = torch.randint(0, 1000, (1, bs), generator=g)
            post = torch.randint(1000, 2000, (1, bs), generator=g)      # disjoint ranges: no accidental match
            post[0, :acc_want] = block[0, 1:acc_want + 1]
            if acc_want + 2 < bs:                                         # a later coincidence must not count
                post[0, acc_want + 1] = block[0, acc_want + 2]
            # the reference's expression, model/dflash.py:258
            acc = (block[:, 1:] == post[:, :-1]).cumprod(dim=1).sum(dim=1)[0].item()
            start = int(torch.randint(5, 50, (1,), generator=g))
            out = torch.full((1, start + bs + 4), 9999, dtype=torch.long)
            out[:, start] = block[0, 0]
            out[:, start:start + acc + 1] = block[:, :acc + 1]            # :259
            out[:, start + acc + 1] = post[:, acc]                         # :260
            cases.append({"bs": bs, "block": block[0].tolist(), "posterior": post[0].tolist(), "start": start,
                          "acc": int(acc), "new_start": start + acc + 1, "out": out[0].tolist()})
    json.dump(cases, open(os.path.join(HERE, "accept.json"), "w"))
    print("G3", len(cases), "cases")


# ---------------------------------------------------------------- G4
def _scripted(dtype, attn, plan_seed, tape_seed, cfg, total):
    base = H.tiny_target(dtype=dtype, attn_impl=attn)
    tape = H.make_tape(total + 64, cfg.vocab_size, tape_seed, forbid=(cfg.mask_token_id,))
    return H.ScriptedTarget(base, tape, H.make_plan(64, cfg.block_size, plan_seed))


def gen_e2e():
    cfg = H.tiny_cfg()
    res = {}
    ref_bench.cuda_time = time.perf_counter
    clock = {"t": 0.0}

    def fake_time():
        clock["t"] += 1e-3
        return clock["t"]
    ref_dyn.cuda_time = fake_time

    for name, dtype, attn in (("f32_eager", torch.float32, "eager"), ("bf16_sdpa", torch.bfloat16, "sdpa")):
        sd = H.draft_weights(cfg, dtype=dtype)
        model = ref_draft(cfg, sd, dtype, attn)
        g = torch.Generator().manual_seed(21)
        prompt = torch.randint(0, 2000, (1, 37), generator=g)

        # (a) spec_generate, scripted acceptance, no stop ids
        tgt = _scripted(dtype, attn, 1, 2, cfg, 37 + 90)
        ids = model.spec_generate(target=tgt, input_ids=prompt, max_new_tokens=90, stop_token_ids=None,
                                  temperature=0.0)
        res[f"{name}/spec"] = {"prompt": prompt[0].tolist(), "max_new_tokens": 90, "ids": ids[0].tolist(),
                               "verify": tgt.verify_log, "plan_seed": 1, "tape_seed": 2}

        # (b) spec_generate with a stop token sitting in the tape
        tgt = _scripted(dtype, attn, 3, 4, cfg, 37 + 90)
        stop_id = int(tgt.tape[37 + 29])
        ids = model.spec_generate(target=tgt, input_ids=prompt, max_new_tokens=90, stop_token_ids=[stop_id, 5],
                                  temperature=0.0)
        res[f"{name}/spec_stop"] = {"prompt": prompt[0].tolist(), "max_new_tokens": 90, "ids": ids[0].tolist(),
                                    "verify": tgt.verify_log, "plan_seed": 3, "tape_seed": 4,
                                    "stop_token_ids": [stop_id, 5]}

        # (c) natural run (no scripting): random weights -> tau == 1, output == greedy AR
        base = H.tiny_target(dtype=dtype, attn_impl=attn)
        ids = model.spec_generate(target=base, input_ids=prompt, max_new_tokens=24, stop_token_ids=None,
                                  temperature=0.0)
        res[f"{name}/natural"] = {"prompt": prompt[0].tolist(), "max_new_tokens": 24, "ids": ids[0].tolist()}

        # (d) harness form: benchmark.dflash_generate — bs=1 baseline, bs=12 with tail clamp, draft_steps=2
        for key, bs, steps, mnt in (("gen_bs1", 1, 1, 12), ("gen_bs12", 12, 1, 50), ("gen_bs16", 16, 1, 61),
                                    ("gen_steps2", 16, 2, 40)):
            tgt = _scripted(dtype, attn, 6, 7, cfg, 37 + mnt)
            r = ref_bench.dflash_generate(model, tgt, prompt, cfg.mask_token_id, mnt, bs, None, 0.0,
                                          collect_profile=False, draft_steps=steps)
            res[f"{name}/{key}"] = {"prompt": prompt[0].tolist(), "max_new_tokens": mnt, "block_size": bs,
                                    "draft_steps": steps, "ids": r.output_ids[0].tolist(),
                                    "acceptance_lengths": [int(a) for a in r.acceptance_lengths],
                                    "num_output_tokens": int(r.num_output_tokens), "plan_seed": 6, "tape_seed": 7}

        # (e) variable block size per cycle: benchmark_dynamic_schedule.dflash_generate_policy
        sched = ref_dyn.EWMAPerformanceScheduler(
            candidates=[8, 12, 16], scheduler_mode="ewma", warmup_cycles=6, ewma_alpha=0.25, switch_margin=0.03,
            required_streak=2, cooldown_cycles=2, probe_interval=5, low_accept_threshold=0.2, low_accept_streak=3,
            adl_rho=0.3, adl_delta=1.0, adl_k_min=8, adl_k_max=16, adl_neighborhood=4)
        tgt = _scripted(dtype, attn, 8, 9, cfg, 37 + 120)
        stop_id = int(tgt.tape[37 + 100])
        r = ref_dyn.dflash_generate_policy(model=model, target=tgt, input_ids=prompt,
                                           mask_token_id=cfg.mask_token_id, max_new_tokens=120,
                                           stop_token_ids=[stop_id], temperature=0.0, scheduler=sched)
        res[f"{name}/policy"] = {"prompt": prompt[0].tolist(), "max_new_tokens": 120, "stop_token_ids": [stop_id],
                                 "ids": r.output_ids[0].tolist(),
                                 "acceptance_lengths": [int(a) for a in r.acceptance_lengths],
                                 "used_block_sizes": [int(b) for b in r.used_block_sizes],
                                 "chosen_block_sizes": [int(t["chosen_block_size"]) for t in r.cycle_trace],
                                 "l_gen": [float(t["l_gen"]) for t in r.cycle_trace],
                                 "plan_seed": 8, "tape_seed": 9}
        print("G4", name, "done")
    json.dump(res, open(os.path.join(HERE, "e2e.json"), "w"))


# ---------------------------------------------------------------- G5
def gen_scheduler():
    traces = []
    g = np.random.default_rng(0)
    param_sets = [
        dict(candidates=[8, 12, 16], scheduler_mode="ewma", warmup_cycles=6, ewma_alpha=0.25, switch_margin=0.03,
             required_streak=2, cooldown_cycles=2, probe_interval=5, low_accept_threshold=0.2, low_accept_streak=3,
             adl_rho=0.3, adl_delta=1.0, adl_k_min=8, adl_k_max=16, adl_neighborhood=4),
        dict(candidates=[16, 4, 8, 24], scheduler_mode="adl_ewma", warmup_cycles=3, ewma_alpha=0.5,
             switch_margin=0.0, required_streak=1, cooldown_cycles=0, probe_interval=0, low_accept_threshold=0.3,
             low_accept_streak=2, adl_rho=0.4, adl_delta=2.0, adl_k_min=4, adl_k_max=24, adl_neighborhood=8),
        dict(candidates=[8, 16], scheduler_mode="adl_ewma", warmup_cycles=0, ewma_alpha=1.0, switch_margin=0.1,
             required_streak=3, cooldown_cycles=4, probe_interval=3, low_accept_threshold=0.5, low_accept_streak=1,
             adl_rho=1.0, adl_delta=0.0, adl_k_min=8, adl_k_max=16, adl_neighborhood=0),
    ]
    for params in param_sets:
        s = ref_dyn.EWMAPerformanceScheduler(**params)
        steps = []
        for cyc in range(120):
            chosen = s.select(cyc)
            bs = chosen if cyc % 17 != 16 else 1       # sprinkle clamped tail cycles
            quality = 0.15 if 40 <= cyc < 60 else 0.6   # a stretch of poor acceptance
            tau = int(min(bs, max(1, g.binomial(bs, quality))))
            cycle_s = float(0.04 + 0.0008 * bs + g.normal(0, 0.002))
            l_gen = float(bs if cyc % 5 else max(1, bs // 2))
            s.update(tau=tau, cycle_s=cycle_s, effective_bs=bs, cycle_idx=cyc, l_gen=l_gen)
            steps.append({"chosen": int(chosen), "bs": int(bs), "tau": tau, "cycle_s": cycle_s, "l_gen": l_gen,
                          "current": int(s.current), "cooldown_left": int(s.cooldown_left),
                          "pending_target": int(s.pending_target), "pending_streak": int(s.pending_streak),
                          "tau_hat": {str(k): v for k, v in s.tau_hat.items()},
                          "score_hat": {str(k): v for k, v in s.score_hat.items()},
                          "adl_target_k": int(s.adl_target_k), "adl_target_bs": int(s.adl_target_bs),
                          "adl_lgen_hat": s.adl_lgen_hat, "adl_lacc_hat": s.adl_lacc_hat})
        traces.append({"params": params, "steps": steps})
    json.dump(traces, open(os.path.join(HERE, "scheduler.json"), "w"))
    print("G5 done")


# ---------------------------------------------------------------- G6
def gen_sample_t():
    g = torch.Generator().manual_seed(13)
    logits = (torch.randn(1, 16, 2048, generator=g) * 3).to(torch.bfloat16)
    torch.manual_seed(0)
    ids = ref_sample(logits, 0.7)
    block = torch.randint(0, 2048, (1, 16), generator=g)
    block[0, 1:6] = ids[0, :5]
    acc = (block[:, 1:] == ids[:, :-1]).cumprod(dim=1).sum(dim=1)[0].item()
    np.savez_compressed(os.path.join(HERE, "sample_t.npz"), logits=f32(logits), ids=ids.numpy(),
                        block=block.numpy(), acc=np.array(acc), temperature=np.array(0.7), seed=np.array(0))
    print("G6 done")


# ---------------------------------------------------------------- G7
def gen_harness():
    """summarize_mode / summarize_profile of the reference harness on synthetic samples."""
    g = np.random.default_rng(3)
    samples = []
    for i in range(5):
        n_out = int(g.integers(50, 300))
        dec = float(g.uniform(0.5, 3.0))
        samples.append(dict(num_input_tokens=int(g.integers(10, 80)), num_output_tokens=n_out,
                            time_to_first_token=float(g.uniform(0.05, 0.2)), time_per_output_token=dec / n_out,
                            wall_time_s=dec + 0.3, acceptance_lengths=[int(x) for x in g.integers(1, 17, size=12)],
                            profile_summary={"target_prefill_s": float(g.uniform(0.05, 0.2)),
                                             "target_decode_s": float(g.uniform(0.5, 2.0)),
                                             "draft_decode_s": float(g.uniform(0.05, 0.4)),
                                             "cycle_decode_s_sum": float(g.uniform(0.6, 2.5)),
                                             "decode_wall_s": dec, "profiled_cycles": 12}))
    ns = [types.SimpleNamespace(**s) for s in samples]
    json.dump({"samples": samples, "summarize_mode": ref_bench.summarize_mode(ns),
               "summarize_profile": ref_bench.summarize_profile(ns)},
              open(os.path.join(HERE, "harness.json"), "w"))
    print("G7 done")


# ---------------------------------------------------------------- G8: blocks wider than 16 rows
def gen_wide():
    """Block sizes 17..32 (results.md:11-16 sweeps 20 and 24; the scheduler takes any candidate >= 2,
    benchmark_dynamic_schedule.py:44-51): G1-style forward vectors and G4-style loop runs."""
    tiny = H.tiny_cfg()
    # (bs, tau_next): context rows of the next cycle go up to the block size
    gen_draft_forward("tiny_bf16_sdpa_wide", tiny, torch.bfloat16, "sdpa", 40,
                      [(24, 5), (20, 20), (32, 17), (17, 2), (24, 24), (16, 9)], 103)
    cfg = tiny
    res = {}
    ref_bench.cuda_time = time.perf_counter
    clock = {"t": 0.0}

    def fake_time():
        clock["t"] += 1e-3
        return clock["t"]
    ref_dyn.cuda_time = fake_time
    dtype, attn = torch.bfloat16, "sdpa"
    sd = H.draft_weights(cfg, dtype=dtype)
    model = ref_draft(cfg, sd, dtype, attn)
    prompt = torch.randint(0, 2000, (1, 37), generator=torch.Generator().manual_seed(21))

    def scripted(plan_seed, tape_seed, total, plan_bs):
        base = H.tiny_target(dtype=dtype, attn_impl=attn)
        tape = H.make_tape(total + 64, cfg.vocab_size, tape_seed, forbid=(cfg.mask_token_id,))
        return H.ScriptedTarget(base, tape, H.make_plan(64, plan_bs, plan_seed))

    for key, bs, mnt in (("gen_bs20", 20, 90), ("gen_bs24", 24, 100), ("gen_bs32", 32, 120)):
        tgt = scripted(16, 17, 37 + mnt, bs)
        r = ref_bench.dflash_generate(model, tgt, prompt, cfg.mask_token_id, mnt, bs, None, 0.0,
                                      collect_profile=False, draft_steps=1)
        res[f"bf16_sdpa/{key}"] = {"prompt": prompt[0].tolist(), "max_new_tokens": mnt, "block_size": bs,
                                   "draft_steps": 1, "ids": r.output_ids[0].tolist(),
                                   "acceptance_lengths": [int(a) for a in r.acceptance_lengths],
                                   "num_output_tokens": int(r.num_output_tokens), "plan_seed": 16, "tape_seed": 17,
                                   "plan_bs": bs}
    sched = ref_dyn.EWMAPerformanceScheduler(
        candidates=[12, 20, 24], scheduler_mode="ewma", warmup_cycles=6, ewma_alpha=0.25, switch_margin=0.03,
        required_streak=2, cooldown_cycles=2, probe_interval=5, low_accept_threshold=0.2, low_accept_streak=3,
        adl_rho=0.3, adl_delta=1.0, adl_k_min=8, adl_k_max=24, adl_neighborhood=4)
    tgt = scripted(18, 19, 37 + 150, 24)
    r = ref_dyn.dflash_generate_policy(model=model, target=tgt, input_ids=prompt, mask_token_id=cfg.mask_token_id,
                                       max_new_tokens=150, stop_token_ids=None, temperature=0.0, scheduler=sched)
    res["bf16_sdpa/policy_wide"] = {"prompt": prompt[0].tolist(), "max_new_tokens": 150, "stop_token_ids": None,
                                    "ids": r.output_ids[0].tolist(),
                                    "acceptance_lengths": [int(a) for a in r.acceptance_lengths],
                                    "used_block_sizes": [int(b) for b in r.used_block_sizes],
                                    "chosen_block_sizes": [int(t["chosen_block_size"]) for t in r.cycle_trace],
                                    "l_gen": [float(t["l_gen"]) for t in r.cycle_trace],
                                    "plan_seed": 18, "tape_seed": 19, "plan_bs": 24, "candidates": [12, 20, 24]}
    json.dump(res, open(os.path.join(HERE, "e2e_wide.json"), "w"))
    print("G8 done")


# ---------------------------------------------------------------- G9: multi-candidate verify
def _distinct_bf16_logits(g, rows, V, spread):
    """[1, rows, V] bf16 logits whose values are pairwise distinct inside a row, so that torch.topk / argsort have
    ONE answer (bf16 lm_head outputs tie often, and topk's order among ties is an implementation detail)."""
    bits = torch.arange(0x3C00, 0x4180, dtype=torch.int32)               # bf16 patterns of 0.0078 .. 16: all distinct
    pool = torch.cat([bits, bits | 0x8000]).to(torch.int16).view(torch.bfloat16)
    pool = pool[pool.float().abs() <= spread]
    assert pool.numel() >= V, (pool.numel(), V)
    out = torch.empty(1, rows, V, dtype=torch.bfloat16)
    for r in range(rows):
        out[0, r] = pool[torch.randperm(pool.numel(), generator=g)[:V]]
    return out


def gen_candidates():
    """Candidate builders of benchmark_candidate_solutions.py (:84-414) — pure functions of (block ids, draft logits,
    parameters) — run as imported; the loop around them (:570-618) cannot run on transformers 5.15
    (DynamicCache.to_legacy_cache is gone), so what is stored is every builder's output plus the budget rule."""
    import benchmark_candidate_solutions as ref_cand  # noqa: E402  (reference)
    g = torch.Generator().manual_seed(77)
    V = 2048
    cases = []

    def meta_list(meta):
        return [{k: (v if not isinstance(v, float) else float(v)) for k, v in m.items()} for m in meta]

    for ci, (bs, spread) in enumerate(((16, 8.0), (16, 2.0), (12, 8.0), (5, 4.0), (2, 8.0), (16, 16.0))):
        logits = _distinct_bf16_logits(g, bs - 1, V, spread)
        block = torch.randint(0, V, (1, bs), generator=g)
        block[:, 1:] = ref_sample(logits)                       # the loop's greedy fill (:529)
        case = {"bs": bs, "block": block[0].tolist(),
                "logits_bits": logits.view(torch.int16)[0].numpy().astype(np.int16).tolist(), "runs": []}
        for depth, thr, topk, maxc in ((6, -1.0, 2, 4), (3, -1.0, 3, 8), (6, 0.3, 2, 4), (6, 0.02, 2, 4), (1, -1.0, 2, 1)):
            pos = ref_cand.select_branch_positions(logits, bs, depth, thr)
            cands, meta = ref_cand.build_candidate_blocks(block, logits, pos, topk, maxc)
            case["runs"].append({"mode": "branch_beam", "branch_depth": depth, "margin_threshold": thr,
                                 "branch_top_k": topk, "max_candidates": maxc, "selected_positions": pos,
                                 "candidates": [c[0].tolist() for c in cands], "meta": meta_list(meta)})
        for fpl, topk, maxc in ((5, 4, 4), (2, 2, 8), (20, 4, 4), (5, 1, 4), (1, 8, 8), (5, 4, 3)):
            cands, meta, pos = ref_cand.build_fixed_prefix_rank_candidates(block, logits, fpl, topk, maxc)
            case["runs"].append({"mode": "fixed_prefix_rank", "fixed_prefix_len": fpl, "branch_top_k": topk,
                                 "max_candidates": maxc, "selected_positions": pos,
                                 "candidates": [c[0].tolist() for c in cands], "meta": meta_list(meta)})
        for fpl, topk, maxc, smp, thr in ((5, 4, 4, 4, -1.0), (2, 3, 8, 2, -1.0), (5, 4, 4, 4, 0.3), (5, 4, 4, 4, 1e-9),
                                          (1, 2, 6, 8, -1.0), (5, 1, 4, 4, -1.0)):
            cands, meta, pos = ref_cand.build_uncertainty_sparse_rank_candidates(block, logits, fpl, topk, maxc, smp, thr)
            case["runs"].append({"mode": "uncertainty_sparse_rank", "fixed_prefix_len": fpl, "branch_top_k": topk,
                                 "max_candidates": maxc, "sparse_max_positions": smp, "margin_threshold": thr,
                                 "selected_positions": pos, "candidates": [c[0].tolist() for c in cands],
                                 "meta": meta_list(meta)})
        cases.append(case)
    budget = []
    for enabled in (False, True):
        for cyc in (0, 1, 2, 3, 5, 8, 10):
            for ratio in (None, 0.9, 0.85, 0.7, 0.65, 0.3):
                for maxc in (8, 2):
                    kw = dict(enabled=enabled, max_candidates=maxc, cycle_idx=cyc, last_accept_ratio=ratio,
                              budgets=(1, 4, 8), accept_thresholds=(0.85, 0.65), warmup_cycles=2, probe_interval=5)
                    budget.append({**{k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                                   "out": ref_cand.resolve_cycle_max_candidates(**kw)})
    json.dump({"vocab": V, "cases": cases, "budget": budget}, open(os.path.join(HERE, "candidates.json"), "w"))
    print("G9 done", sum(len(c["runs"]) for c in cases), "builder runs")


if __name__ == "__main__":
    if sys.argv[1:] == ["candidates"]:
        gen_candidates()
        sys.exit(0)
    if sys.argv[1:] == ["wide"]:     # later additions are generated alone: the round-1 fixtures stay byte-identical
        gen_wide()
        sys.exit(0)
    tiny, mid = H.tiny_cfg(), H.mid_cfg()
    # (bs, tau_next): block size this cycle, tokens committed after it (= next cycle's ctx rows)
    steps = [(16, 1), (16, 7), (16, 16), (12, 3), (8, 8), (16, 2)]
    gen_draft_forward("tiny_f32_eager", tiny, torch.float32, "eager", 40, steps, 101)
    gen_draft_forward("tiny_bf16_eager", tiny, torch.bfloat16, "eager", 40, steps, 101)
    gen_draft_forward("tiny_bf16_sdpa", tiny, torch.bfloat16, "sdpa", 40, steps, 101)
    gen_draft_forward("mid_bf16_sdpa", mid, torch.bfloat16, "sdpa", 300, steps, 102)
    gen_argmax()
    gen_accept()
    gen_e2e()
    gen_scheduler()
    gen_sample_t()
    gen_harness()
    gen_wide()
    gen_candidates()
