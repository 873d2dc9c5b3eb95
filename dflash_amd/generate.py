"""Decode loops over the HIP hot path, with the reference's call signatures and
return fields:

* `dflash_generate`        — benchmark.py:44-251 (fixed block, tail clamp,
                             block_size==1 baseline, draft_steps, per-cycle profile)
* `dflash_generate_policy` — benchmark_dynamic_schedule.py:260-434 (block size from
                             `EWMAPerformanceScheduler` every cycle)
* `DFlashDraftModel.spec_generate` (model.py) — model/dflash.py:192-277

All three share `run_decode`.  Per cycle the host enqueues: draft forward + fused
lm_head/argmax (kernels), the caller's target forward (PyTorch-ROCm, outside the
path), the posterior argmax kernel and the accept/commit kernel; the only
device->host read is the 4-int accept result the loop needs to slice the
target's hidden states and roll its cache back.
"""
from __future__ import annotations

import time
from types import SimpleNamespace
from typing import Callable, Optional, Sequence

import torch

from . import ops
from .utils import extract_context_feature, sample


def cuda_time() -> float:
    torch.cuda.synchronize()
    return time.perf_counter()


def _new_target_cache(target):
    if hasattr(target, "new_cache"):
        return target.new_cache()
    from transformers import DynamicCache  # the reference's choice (model/dflash.py:214)
    return DynamicCache()


def _trim(output_ids, max_length, mask_token_id, stop_token_ids, n_in):
    """model/dflash.py:269-275."""
    output_ids = output_ids[:, :max_length]
    output_ids = output_ids[:, output_ids[0] != mask_token_id]
    if stop_token_ids is not None:
        st = torch.tensor(stop_token_ids, device=output_ids.device)
        idx = torch.isin(output_ids[0][n_in:], st).nonzero(as_tuple=True)[0]
        if idx.numel() > 0:
            output_ids = output_ids[:, : n_in + idx[0] + 1]
    return output_ids


@torch.inference_mode()
def run_decode(model, target, input_ids: torch.Tensor, *, mask_token_id: int, max_new_tokens: int,
               block_size: int, stop_token_ids, temperature: float, clamp_tail: bool,
               draft_steps: int = 1, collect_profile: bool = False, scheduler=None,
               draft_temperature: float = 0.0, draft_token_hook: Optional[Callable] = None,
               max_block_size: Optional[int] = None) -> SimpleNamespace:
    dev = model.device
    if not input_ids.is_cuda:
        raise RuntimeError("dflash_amd: input_ids must be on the GPU")
    if input_ids.shape[0] != 1:
        raise NotImplementedError("batch = 1 per call, as in the reference; shard requests over ranks/streams")
    max_bs = max_block_size or block_size
    n_in = input_ids.shape[1]
    max_length = n_in + max_new_tokens
    output_ids = torch.full((1, max_length + max_bs), mask_token_id, dtype=torch.long, device=dev)
    position_ids = torch.arange(output_ids.shape[1], device=dev).unsqueeze(0)
    tcache = _new_target_cache(target)
    use_draft = max_bs > 1
    dcache = model.new_cache(max_length + 2 * max_bs) if use_draft else None
    embed_w = _bf16_table(target.model.embed_tokens.weight, dev)
    lm_wp = model.packed_lm_head(target.lm_head) if use_draft else None
    stop_t = torch.tensor(stop_token_ids, dtype=torch.long, device=dev) if stop_token_ids else None
    stop_always = stop_token_ids is not None and mask_token_id in stop_token_ids  # reference scans mask slots too
    result = torch.zeros(4, dtype=torch.int32, device=dev)
    block = torch.empty(1, max_bs, dtype=torch.long, device=dev)

    # ---- prefill (model/dflash.py:218-229)
    t_prefill = cuda_time()
    out = target(input_ids, position_ids=position_ids[:, :n_in], past_key_values=tcache, use_cache=True,
                 logits_to_keep=1, output_hidden_states=use_draft)
    output_ids[:, :n_in] = input_ids
    output_ids[:, n_in:n_in + 1] = sample(out.logits, temperature)
    target_hidden = _taps(out.hidden_states, model.target_layer_ids) if use_draft else None
    time_to_first_token = cuda_time() - t_prefill

    decode_start = cuda_time()
    start = n_in
    taus, used_bs, cycle_trace, lgens = [], [], [], []
    draft_prefill = True
    hook_calls = 0
    cyc = 0
    while start < max_length:
        cycle_t0 = cuda_time() if scheduler is not None else None
        ev = None
        if collect_profile:
            ev = {k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for k in ("cycle", "draft", "target")}
            ev["cycle"][0].record()
        chosen = block_size if scheduler is None else scheduler.select(cyc)
        remaining = max_length - start
        bs = max(1, min(chosen, remaining)) if (clamp_tail or scheduler is not None) else chosen
        l_gen = float(bs)
        blk = block[:, :bs]
        blk.copy_(output_ids[:, start:start + bs])
        if bs > 1:
            if collect_profile:
                ev["draft"][0].record()
            if draft_steps == 1:
                ctx = target_hidden[0]
                S = dcache.get_seq_length()
                if ctx.shape[0] > 16:  # cycle 0: the prompt's context rows, 16 at a time
                    head = ctx.shape[0] - 16
                    model.prefill_context(dcache, ctx[:head], S)
                    ctx, S = ctx[head:], S + head
                hid = model.draft_block(dcache, th_rows=ctx, tau=ctx.shape[0], bs=bs, pos0=S, block_ids=blk[0],
                                        embed=embed_w)
                _draft_ids(model, hid, lm_wp, bs, blk, draft_temperature)
                if draft_token_hook is not None:
                    draft_token_hook(blk, start, hook_calls)
                hook_calls += 1
            else:
                # benchmark.py:112-142: k full passes, each re-embedding the whole block, no
                # draft cache, positions rebuilt as [start-ctx_len, start+bs)
                ctx = target_hidden[0]
                ctx_len = ctx.shape[0]
                for _ in range(draft_steps):
                    tmp = model.new_cache(ctx_len + bs)
                    p0 = max(0, start - ctx_len)
                    c2 = ctx
                    if ctx_len > 16:
                        model.prefill_context(tmp, ctx[:ctx_len - 16], p0)
                        c2, p0 = ctx[ctx_len - 16:], p0 + ctx_len - 16
                    hid = model.draft_block(tmp, th_rows=c2, tau=c2.shape[0], bs=bs, pos0=p0, block_ids=blk[0],
                                            embed=embed_w, append=False)
                    _draft_ids(model, hid, lm_wp, bs, blk, 0.0)
                    if draft_token_hook is not None:
                        draft_token_hook(blk, start, hook_calls)
                    hook_calls += 1
            if scheduler is not None and stop_t is not None:
                pos = torch.isin(blk[0, 1:], stop_t).nonzero(as_tuple=True)[0]
                if pos.numel() > 0:
                    l_gen = float(min(int(pos[0].item()) + 1, bs))
            if collect_profile:
                ev["draft"][1].record()
            if draft_prefill:
                draft_prefill = False
                decode_start = cuda_time()

        # ---- target verify (outside the path; model/dflash.py:249-255)
        if collect_profile:
            ev["target"][0].record()
        want_hidden = use_draft if (scheduler is not None or not clamp_tail) else bs > 1
        out = target(blk, position_ids=position_ids[:, start:start + bs], past_key_values=tcache, use_cache=True,
                     output_hidden_states=want_hidden)
        if collect_profile:
            ev["target"][1].record()
        posterior = sample(out.logits, temperature)

        # ---- accept scan + commit + bookkeeping on the device (:258-268)
        dyn = dcache.dyn if dcache is not None else _scratch_dyn(dev)
        ops.set_dyn(dyn, 0, 0, bs, start)  # start word = pos0 + tau = start
        ops.accept_commit(blk[0], posterior[0].contiguous(), bs, output_ids[0], dyn, stop_t, result)
        res = result.tolist()  # the cycle's one device->host read (synchronises the stream)
        acc = res[0]
        tau = acc + 1
        taus.append(tau)
        used_bs.append(bs)
        lgens.append(l_gen)
        gen_before = start - n_in
        if scheduler is not None:
            cycle_s = cuda_time() - cycle_t0
            scheduler.update(tau=tau, cycle_s=cycle_s, effective_bs=bs, cycle_idx=cyc, l_gen=l_gen)
            cycle_trace.append({
                "cycle_idx": cyc, "start_idx": int(start), "block_size": int(bs), "chosen_block_size": int(chosen),
                "tau": int(tau), "l_gen": float(l_gen), "acceptance_ratio": float(tau / max(1, bs)),
                "cycle_s": float(cycle_s), "tau_hat": scheduler.tau_hat.get(bs),
                "cycle_hat": scheduler.cycle_hat.get(bs), "score_hat": scheduler.score_hat.get(bs),
                "current_block_size": int(scheduler.current), "adl_lgen_hat": scheduler.adl_lgen_hat,
                "adl_lacc_hat": scheduler.adl_lacc_hat, "adl_target_k": int(scheduler.adl_target_k),
                "adl_target_bs": int(scheduler.adl_target_bs)})
        elif collect_profile:
            ev["cycle"][1].record()
            cycle_trace.append({"cycle_idx": cyc, "generated_tokens_before": int(gen_before),
                                "effective_block_size": int(bs), "tau": int(tau),
                                "acceptance_ratio": float(tau / max(1, bs)), "_events": ev})
        start += tau
        tcache.crop(start)
        if want_hidden and use_draft:
            target_hidden = _taps(out.hidden_states, model.target_layer_ids)[:, :tau, :]
        cyc += 1
        if stop_always or res[2]:
            break

    output_ids = _trim(output_ids, max_length, mask_token_id, stop_token_ids, n_in)
    num_output_tokens = output_ids.shape[1] - n_in
    total_decode_time = cuda_time() - decode_start
    profile_summary = None
    if collect_profile and scheduler is None:
        profile_summary = _resolve_profile(cycle_trace, time_to_first_token, total_decode_time)
    return SimpleNamespace(output_ids=output_ids, num_input_tokens=n_in, num_output_tokens=num_output_tokens,
                           time_to_first_token=time_to_first_token,
                           time_per_output_token=total_decode_time / max(1, num_output_tokens),
                           acceptance_lengths=taus, used_block_sizes=used_bs, l_gen=lgens,
                           cycle_trace=cycle_trace, profile_summary=profile_summary)


_DYN = {}
_TABLES = {}


def _taps(hidden_states, layer_ids):
    """extract_context_feature (model/utils.py:16-25); the kernels take bf16 rows, which
    is what a bf16 target (the reference's dtype, benchmark.py:404) already returns."""
    t = extract_context_feature(hidden_states, layer_ids)
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


def _bf16_table(w, dev):
    if w.dtype == torch.bfloat16 and w.device == dev and w.is_contiguous():
        return w.detach()
    key = (w.data_ptr(), tuple(w.shape), w._version)
    if key not in _TABLES:
        _TABLES.clear()
        _TABLES[key] = w.detach().to(device=dev, dtype=torch.bfloat16).contiguous()
    return _TABLES[key]


def _scratch_dyn(dev):
    if dev not in _DYN:
        _DYN[dev] = torch.zeros(8, dtype=torch.int32, device=dev)
    return _DYN[dev]


def _draft_ids(model, hid_frag, lm_wp, bs, blk, draft_temperature):
    """blk[0, 1:bs] <- draft tokens.  Greedy (every loop but the policy one at T>0):
    fused lm_head GEMM + argmax.  benchmark_dynamic_schedule.py:342 samples the draft
    with the temperature: then the logits are materialised by the same GEMM and the
    reference's softmax + multinomial is applied to them."""
    if draft_temperature < 1e-5:
        model.draft_tokens(hid_frag, lm_wp, bs, blk[0])
        return
    V = model.config.vocab_size
    logits = torch.empty(16, V, dtype=torch.bfloat16, device=model.device)
    model.draft_tokens(hid_frag, lm_wp, bs, blk[0], logits=logits)
    blk[:, 1:bs] = sample(logits[1:bs].unsqueeze(0), draft_temperature)


def _resolve_profile(cycle_trace, ttft, decode_wall):
    """benchmark.py:208-240: turn the recorded event pairs into seconds."""
    torch.cuda.synchronize()
    tot = {"draft": 0.0, "target": 0.0, "cycle": 0.0}
    for row in cycle_trace:
        ev = row.pop("_events")
        for k in tot:
            a, b = ev[k]
            s = a.elapsed_time(b) / 1000.0 if (k != "draft" or row["effective_block_size"] > 1) else 0.0
            row[f"{k}_s"] = float(s)
            tot[k] += s
    den = max(1e-12, tot["draft"] + tot["target"])
    return {"target_prefill_s": float(ttft), "target_decode_s": float(tot["target"]),
            "draft_decode_s": float(tot["draft"]), "cycle_decode_s_sum": float(tot["cycle"]),
            "decode_wall_s": float(decode_wall), "profiled_cycles": int(len(cycle_trace)),
            "draft_share_decode": float(tot["draft"] / den), "target_share_decode": float(tot["target"] / den)}


def dflash_generate(model, target, input_ids: torch.Tensor, mask_token_id: int, max_new_tokens: int,
                    block_size: int, stop_token_ids, temperature: float = 0.0, collect_profile: bool = False,
                    draft_steps: int = 1, draft_token_hook=None) -> SimpleNamespace:
    """benchmark.py:44-55 signature; returns the namespace of :242-251."""
    r = run_decode(model, target, input_ids, mask_token_id=mask_token_id, max_new_tokens=max_new_tokens,
                   block_size=block_size, stop_token_ids=stop_token_ids, temperature=temperature, clamp_tail=True,
                   draft_steps=draft_steps, collect_profile=collect_profile, draft_token_hook=draft_token_hook)
    return SimpleNamespace(output_ids=r.output_ids, num_input_tokens=r.num_input_tokens,
                           num_output_tokens=r.num_output_tokens, time_to_first_token=r.time_to_first_token,
                           time_per_output_token=r.time_per_output_token, acceptance_lengths=r.acceptance_lengths,
                           cycle_trace=r.cycle_trace, profile_summary=r.profile_summary)


def dflash_generate_policy(*, model, target, input_ids: torch.Tensor, mask_token_id: int, max_new_tokens: int,
                           stop_token_ids, temperature: float, fixed_block_size: Optional[int] = None,
                           scheduler=None, draft_token_hook=None) -> SimpleNamespace:
    """benchmark_dynamic_schedule.py:260-272 signature; returns the namespace of :425-434."""
    if fixed_block_size is None and scheduler is None:
        raise ValueError("Either fixed_block_size or scheduler must be provided.")
    max_bs = fixed_block_size if fixed_block_size is not None else max(scheduler.candidates)
    sched = scheduler if fixed_block_size is None else _Fixed(fixed_block_size)
    r = run_decode(model, target, input_ids, mask_token_id=mask_token_id, max_new_tokens=max_new_tokens,
                   block_size=max_bs, stop_token_ids=stop_token_ids, temperature=temperature, clamp_tail=True,
                   scheduler=sched, draft_temperature=temperature, max_block_size=max_bs,
                   draft_token_hook=draft_token_hook)
    if fixed_block_size is not None:
        for row in r.cycle_trace:
            for k in ("tau_hat", "cycle_hat", "score_hat", "current_block_size", "adl_lgen_hat", "adl_lacc_hat",
                      "adl_target_k", "adl_target_bs"):
                row[k] = None
    return SimpleNamespace(output_ids=r.output_ids, num_input_tokens=r.num_input_tokens,
                           num_output_tokens=r.num_output_tokens, time_to_first_token=r.time_to_first_token,
                           time_per_output_token=r.time_per_output_token, acceptance_lengths=r.acceptance_lengths,
                           used_block_sizes=r.used_block_sizes, cycle_trace=r.cycle_trace)


class _Fixed:
    """fixed_block_size path of dflash_generate_policy: a scheduler that never moves."""
    candidates = ()
    tau_hat = cycle_hat = score_hat = {}
    current = adl_target_k = adl_target_bs = 0
    adl_lgen_hat = adl_lacc_hat = None

    def __init__(self, bs):
        self.bs = bs
        self.current = self.adl_target_k = self.adl_target_bs = bs

    def select(self, cyc):
        return self.bs

    def update(self, **_):
        pass
