#!/usr/bin/env python3
"""bench.py — accepted tokens/s of the DFlash decode cycle on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): Qwen3-8B-shaped target (36 layers,
HF `Qwen3ForCausalLM`, PyTorch-ROCm — outside the hot path) + 5-layer DFlash-b16 draft,
seeded random-init bf16 weights, one request per GPU, 1024 random prompt ids, block 16,
temperature 0.  A *step* is one decode cycle: draft block forward + fused lm_head/argmax,
target verify (NativeTarget: the same HIP kernels; `--hf-verify` sends it through the HF forward
instead), posterior argmax + accept/commit.

Random weights never agree (tau == 1), so acceptance is scripted as SURVEY.md §8d
prescribes: the target's greedy continuation G is known beforehand and, after the
fully timed draft forward + argmax, the draft tokens are overwritten with
G[start+1 : start+k] followed by a wrong id, k drawn from a seeded truncated-geometric
law whose mean tau matches the published 7.3 (the K timed cycles are conditioned on it; the
overlay rows are laid out before the timed loop, inside it they cost one 16-id copy per cycle).
`value` is committed tokens / wall time over all ranks; `raw_tau1_value` is the same cycles
counted at tau = 1.  Setup before the W warmup steps: prefill, cycle 0 (it carries the one-off
projection of the prompt's 1024 context rows into the draft cache) and the first steady-state
cycle (one-off code-object loads, ~60 ms) — so that even `--warmup 0` times steady-state cycles.

A plainly random-init bf16 target has near-zero top-2 logit margins: its argmax flips
between a 1-token and a 16-token forward (measured: 65 % of tokens reproduced), so no
greedy tape survives the verify.  The synthetic target is therefore given a large-margin
greedy rule WITHOUT changing its architecture, byte count or FLOPs: seeded random
weights, embedding std 1.0, o_proj/down_proj scaled by 0.02 (the residual stream stays
embedding-dominated through all 36 layers), lm_head = 0.02 * embedding rows permuted by
a seeded single-cycle permutation.  Its greedy next token is perm[token] with a logit
margin ~70, so G is a closed-form walk and `lossless_fraction` (committed ids == G)
must read 1.0.

N > 1: every rank runs its own request (weak scaling, no collective on the accept path;
one all-reduce of the timing/token scalars after the timed region).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

LM_HEAD_BYTES = 151936 * 4096 * 2
DRAFT_WEIGHT_BYTES = 2_097_252_864  # SURVEY.md §8d probe of the 8B-shaped draft


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_hf_target(dev, layers=36):
    from dflash_amd.config import QWEN3_8B_TARGET as T
    from dflash_amd.synthetic import impose_greedy_walk, make_hf_qwen3
    m = make_hf_qwen3({**T, "num_layers": layers}, dev)
    return m, impose_greedy_walk(m, seed=1234)   # large-margin greedy rule, see module docstring


def tau_plan(n_pre, steps, bs, seed, mean_tau=7.3, extra=8):
    """k_c = number of agreeing draft tokens per cycle; tau = k + 1 = 1 + min(Geom(p), bs-1) with
    p solving E[tau] = mean_tau.  The first n_pre cycles (cycle 0 + warmup) and `extra` spare
    ones are free draws; the `steps` TIMED cycles are one seeded draw conditioned on their taus
    summing to round(steps * mean_tau), so the timed window has the published mean acceptance
    length whatever --steps the driver picks (a free window of 48 cycles is off by +-1-2 %)."""
    lo, hi = 0.0, 1.0
    for _ in range(60):
        p = 0.5 * (lo + hi)
        m = 1.0 + sum(p ** j for j in range(1, bs))
        lo, hi = (p, hi) if m < mean_tau else (lo, p)
    g = torch.Generator().manual_seed(seed)

    def draw(n):
        u = torch.rand(n, bs - 1, generator=g)
        return (u < p).long().cumprod(dim=1).sum(dim=1)

    pre = draw(n_pre).tolist()
    want = round(steps * mean_tau)
    k = draw(steps)
    for _ in range(20000):
        if int(k.sum()) + steps == want:
            break
        k = draw(steps)
    else:  # never seen; nudge single cycles so the sum still matches
        k = k.clone()
        i = 0
        while int(k.sum()) + steps != want:
            d = 1 if int(k.sum()) + steps < want else -1
            if 0 <= int(k[i % steps]) + d <= bs - 1:
                k[i % steps] += d
            i += 1
    return pre + k.tolist() + draw(extra).tolist()


def gpu_leg(args, rank, world, dev):
    from dflash_amd import DFlashConfig, DFlashDraftModel
    from dflash_amd.config import QWEN3_8B_DRAFT
    from dflash_amd.generate import DecodeSession

    torch.manual_seed(0)
    t0 = time.time()
    target, perm = make_hf_target(dev, layers=args.target_layers)
    if not args.hf_verify:
        from dflash_amd import NativeTarget
        # verify AND prefill on the kernels; the wrapped HF model is dropped after packing (one copy of
        # the target's weights in memory) unless --hf-prefill asks for the round-2 configuration
        target = NativeTarget(target, attn_impl=args.attn_impl, keep_hf=args.hf_prefill,
                              prefill="hf" if args.hf_prefill else "native")
        target.fuse_oproj = args.fuse_oproj
    cfg = DFlashConfig(**{**QWEN3_8B_DRAFT, "num_target_layers": args.target_layers})
    draft = DFlashDraftModel(cfg, device=dev)
    draft.attn_impl = args.attn_impl
    draft.fuse_oproj = args.fuse_oproj
    # seeded init directly on the GPU (CPU generation of 1e9 normals costs a minute)
    g = torch.Generator(device=dev).manual_seed(0)
    sd = {k: (torch.randn(s, generator=g, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
          if len(s) == 2 else torch.ones(s, device=dev, dtype=torch.bfloat16)
          for k, s in cfg.state_dict_shapes().items()}
    draft.load_state_dict(sd)
    del sd
    torch.cuda.synchronize()
    log(f"[rank {rank}] models ready in {time.time() - t0:.1f}s")

    # diagnostic knob (DESIGN.md "second stream / captured graph" A/B; never set by the driver): put the process into one
    # of the states profiles/r3_graph_ab.txt blamed for slower kernels before anything is timed
    extra = os.environ.get("DFL_BENCH_EXTRA_STREAM", "")
    if extra:
        keep = gpu_leg.__dict__.setdefault("_keep", {})
        if extra in ("idle", "used"):
            keep["s2"] = torch.cuda.Stream()
        if extra == "used":
            with torch.cuda.stream(keep["s2"]):
                keep["t"] = torch.zeros(64, device=dev) + 1
            keep["s2"].synchronize()
        if extra == "graph1":      # a one-kernel torch graph, captured and never replayed
            g1, y = torch.cuda.CUDAGraph(), torch.zeros(64, device=dev)
            with torch.cuda.graph(g1):
                y += 1
            keep["g1"] = (g1, y)
        torch.cuda.synchronize()

    if args.requests_per_gpu > 1:
        return batched_leg(args, rank, dev, draft, target, perm, cfg)

    bs, P = 16, args.prefix
    prompt = torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    ncyc = args.warmup + args.steps + 2
    plan = tau_plan(2 + args.warmup, args.steps, bs, seed=100 + rank)
    need = sum(k + 1 for k in plan[:ncyc]) + 2 * bs
    mask_id = cfg.mask_token_id

    # ---- the target's greedy continuation G in closed form: G[p+1] = perm[G[p]]
    from dflash_amd.synthetic import greedy_walk
    G = greedy_walk(perm, prompt, need + 2 * bs).to(dev)

    # The scripted overlay, laid out before the timed loop so that it costs ONE small copy per cycle inside it (the
    # scripted acceptance lengths fix every cycle's start): row c = k agreeing tokens of G, then a token that is NOT
    # the target's (the draft forward, lm_head and argmax still run in full; their ids are overwritten).
    rep_rows = torch.zeros(len(plan), bs, dtype=torch.long)
    Gc, st = G.cpu(), P
    for c, k in enumerate(plan[:ncyc + 2]):
        if st + bs + 1 >= Gc.numel():
            break
        rep_rows[c, 1:k + 1] = Gc[st + 1:st + k + 1]
        if k + 1 < bs:
            rep_rows[c, k + 1] = (Gc[st + k + 1] + 1) % 151000
        st += k + 1
    rep_rows = rep_rows.to(dev)

    def hook(blk, start, call):
        k = plan[call]
        n = min(k + 2, bs)
        if n > 1:
            blk[0, 1:n] = rep_rows[call, 1:n]

    for rep in range(2):   # rep 0 pays the one-off costs (code-object loads, allocator, HF lazy init): rep 1 is reported
        s = DecodeSession(draft, target, prompt, mask_token_id=mask_id, max_new_tokens=need, max_block_size=bs,
                          stop_token_ids=None, temperature=0.0, draft_token_hook=hook)
        torch.cuda.synchronize()
        t_pf = time.perf_counter()
        s.prefill()
        torch.cuda.synchronize()
        t_c0 = time.perf_counter()
        s.cycle(bs)                  # cycle 0: carries the one-off 1024-row draft-context prefill
        torch.cuda.synchronize()
        ttft_side = {"target_prefill_ms": 1e3 * (t_c0 - t_pf), "cycle0_ms": 1e3 * (time.perf_counter() - t_c0),
                     "note": "second request on warm code (outside the timed region): the target prefill runs "
                             + ("through the wrapped HF model" if (args.hf_verify or args.hf_prefill) else
                                "on the kernels (csrc/prefill.hip: MFMA GEMMs on the packed weights, causal attention, norm / RoPE / cache write)")
                             + f"; cycle 0 = projection of the {P} prompt context rows into the draft cache "
                             "(model/dflash.py:73-85, 64 rows per pass) + one decode cycle"}
    s.cycle(bs, ahead_ok=True)       # first steady-state cycle: one-off code-object loads (60 ms) — setup, like cycle 0
    for _ in range(args.warmup):
        s.cycle(bs, ahead_ok=True)

    ev_all = []
    # HIP events right around the lm_head GEMM launch itself (dfl_gemm_argmax_timed records them on the launch stream):
    # created, and recorded once so that their handles exist, before the timed region
    lm_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in lm_ev:
        a.record()
        b.record()
    # the same around ONE gate/up GEMM launch of the target verify per timed cycle (layer = cycle index mod layers): the
    # kernel with the largest share of the cycle (36 + 5 launches; rocprof: 34 % of the GPU time)
    gu_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in gu_ev:
        a.record()
        b.record()
    native = not args.hf_verify
    if args.graph:   # A/B: the steady-state cycle as two hipGraph replays (DecodeSession.capture), captured before the timed region
        s.capture(bs)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    draft.lm_head_events_log = []
    # Instrumented cycles are SAMPLED (every --event-every-th timed cycle): an event record is a barrier + timestamp packet
    # on the stream, 2.6 us by the kernel trace, and ten per cycle (phase marks, the lm_head pair, the gate/up pair) were
    # ~0.9 % of the cycle they were meant to observe.  The cycle after an instrumented one only collects the pairs of
    # the draft forward that was enqueued ahead of it.
    E = max(1, args.event_every)
    if args.graph:
        E = max(E, args.steps)   # ONE eager, instrumented cycle; the others are replayed
    s.host_times = []
    for i in range(args.steps):
        instr = i % E == 0
        s.events = {} if (instr or (i - 1) % E == 0) else None
        s.record_events = instr
        draft.lm_head_events = lm_ev[i] if instr else None
        if native:
            target.gu_events = (i % args.target_layers, gu_ev[i][0], gu_ev[i][1]) if instr else None
        # fixed block size: the next cycle's draft is enqueued behind this cycle's accept
        r = s.cycle_graph(bs) if (args.graph and s.events is None) else s.cycle(bs, ahead_ok=True)
        if s.events:
            ev_all.append(s.events)
        tokens += r.tau
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    s.events, s.record_events = None, True
    host_enq = sum(a for a, _ in s.host_times) / max(1, len(s.host_times))
    host_wait = sum(b for _, b in s.host_times) / max(1, len(s.host_times))
    s.host_times = None
    draft.lm_head_events = None
    if native:
        target.gu_events = None

    def avg_ms(key):   # (the first timed cycle's run-ahead draft was enqueued by the last warmup cycle: no pair for it)
        have = [e for e in ev_all if key in e and None not in e[key]]
        return sum(e[key][0].elapsed_time(e[key][1]) for e in have) / max(1, len(have))

    draft_ms, target_ms = avg_ms("draft"), avg_ms("target")
    lm_used = draft.lm_head_events_log or [lm_ev[i] for i in range(0, args.steps, E)]   # (the pairs actually recorded)
    draft.lm_head_events_log = None
    lm_ms = sum(a.elapsed_time(b) for a, b in lm_used) / len(lm_used)
    gu_used = [gu_ev[i] for i in range(0, args.steps, E)]
    gu_ms = sum(a.elapsed_time(b) for a, b in gu_used) / len(gu_used) if native else None
    # committed ids must be the target's own greedy continuation (losslessness)
    n_ok = int((s.output_ids[0, P:s.start] == G[P:s.start]).sum())
    lossless = n_ok / max(1, s.start - P)

    from dflash_amd import distributed as D
    dt_max, tok_sum = D.reduce_timing(dt, float(tokens), device=dev)
    _, cyc_sum = D.reduce_timing(dt, float(args.steps), device=dev)
    # PMC passes cannot run inside the timed bench (rocprofv3 --pmc serialises and slows the run): `traffic` is the
    # committed per-launch HBM byte count of these very kernels — accepted only while the hash of the kernels' sources
    # stored beside it still matches (scripts/pmc_kernels_json.py); a changed kernel reports null until re-profiled
    traffic, rp_us, traffic_note = {}, {}, "no PMC summary for the current kernel sources"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from pmc_kernels_json import source_hash
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if fn.endswith("_pmc_kernels.json"):
                rec = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if rec.get("kernel_source_sha256_16") == source_hash():
                    traffic = {k: v.get("hbm_bytes_per_launch") for k, v in rec.get("kernels", {}).items()}
                    rp_us = {k: v.get("rocprof_avg_us") for k, v in rec.get("kernels", {}).items()}
                    traffic_note = f"profiles/{fn} (kernel sources unchanged since)"
                    break
    except Exception as e:   # never let bookkeeping break the measurement
        traffic_note = f"PMC summary not read: {type(e).__name__}"
    GATE_UP_BYTES = 2 * 12288 * 4096 * 2   # Qwen3-8B: gate and up, [12288][4096] bf16 each
    ev_note = ("achieved/avg_ms from HIP events recorded on the launch stream right before and right after the GEMM launch, "
               "every timed cycle (an event pair also spans the dependent launch boundary, ~2-3 us: rocprof_avg_ms is the "
               "kernel's own duration in the committed rocprofv3 --kernel-trace --stats run of this command, same kernel "
               "sources); traffic = 2*FETCH_SIZE+WRITE_SIZE bytes per launch: " + traffic_note)

    def rp(key, nbytes):   # the committed profile's figure for the same kernel, while the sources are unchanged
        us = rp_us.get(key)
        return {"rocprof_avg_ms": us / 1e3, "frac_rocprof": nbytes / (us * 1e-6) / 8e12} if us else {}
    lm_entry = {"kernel": "k_gemm<1,false,EPI_ARGMAX> (lm_head GEMM + fused argmax; 2 launches per cycle)", "bound": "hbm",
                "achieved": LM_HEAD_BYTES / (lm_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": LM_HEAD_BYTES / (lm_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic.get("lm_head"),
                "bytes_per_launch": LM_HEAD_BYTES, "avg_ms": lm_ms, **rp("lm_head", LM_HEAD_BYTES), "note": ev_note}
    if gu_ms:
        # the roofline object names the kernel with the largest share of the timed cycle: the gate/up GEMM
        # with the fused SiLU*up epilogue, one launch per layer of target and draft; the lm_head GEMM rides beside it
        roofline = {"kernel": "k_gemm<1,false,EPI_SILU> (gate/up GEMM + SiLU*up epilogue; 41 launches per cycle, the largest "
                              "share of the cycle)", "bound": "hbm",
                    "achieved": GATE_UP_BYTES / (gu_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": GATE_UP_BYTES / (gu_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic.get("gate_up"),
                    "bytes_per_launch": GATE_UP_BYTES, "avg_ms": gu_ms, **rp("gate_up", GATE_UP_BYTES), "note": ev_note,
                    "also": [lm_entry]}
    else:
        roofline = lm_entry
    kv_bytes = 20480 * (P + 16)
    hot_bytes = DRAFT_WEIGHT_BYTES + LM_HEAD_BYTES + kv_bytes
    return dict(
        value=tok_sum / dt_max, ms_per_step=1000.0 * dt_max / args.steps, mean_tau=tok_sum / cyc_sum,
        raw_tau1_value=cyc_sum / dt_max, lossless_fraction=lossless,
        roofline=roofline,
        hot_path={"draft_plus_lm_head_ms_per_cycle": draft_ms, "target_verify_ms_per_cycle": target_ms,
                  "algorithmic_bytes_per_cycle": hot_bytes,
                  # (a run too short to see a run-ahead draft's event pair — it is collected one cycle later — has none)
                  "achieved_GBps": hot_bytes / (draft_ms * 1e-3) / 1e9 if draft_ms > 0 else None,
                  "frac_of_8TBps": hot_bytes / (draft_ms * 1e-3) / 1e9 / 8000.0 if draft_ms > 0 else None},
        ttft_side=ttft_side,
        host_side={"enqueue_ms_per_cycle": 1e3 * host_enq, "poll_wait_ms_per_cycle": 1e3 * host_wait,
                   "note": "host share of a timed cycle (time.perf_counter inside DecodeSession.cycle): Python + ctypes enqueueing "
                           "the cycle's ~215 launches (with the run-ahead draft: the NEXT cycle's draft forward included), then "
                           "polling the pinned result word; the GPU is the bottleneck while enqueue < ms_per_step and the poll "
                           "wait is the rest of it"},
    )


def batched_leg(args, rank, dev, draft, target, perm, cfg):
    """--requests-per-gpu R > 1 (BASELINE.json configs[2]: 4 requests per GPU): the R requests of
    this rank advance together, one pass over the weights per cycle (dflash_amd.batch)."""
    from dflash_amd import distributed as D
    from dflash_amd.batch import BatchedDecoder
    from dflash_amd.synthetic import greedy_walk
    R, bs, P = args.requests_per_gpu, 16, args.prefix
    ncyc = args.warmup + args.steps + 2
    plans = [tau_plan(2 + args.warmup, args.steps, bs, seed=100 + rank * 16 + r) for r in range(R)]
    need = max(sum(k + 1 for k in pl[:ncyc]) for pl in plans) + 2 * bs
    prompts = [torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(1 + rank * 16 + r)).to(dev)
               for r in range(R)]
    Gs = [greedy_walk(perm, p, need + 2 * bs).to(dev) for p in prompts]
    dec = BatchedDecoder(draft, target, R, max_rows=P + need + 3 * bs, out_len=P + need + bs,
                         mask_token_id=cfg.mask_token_id)
    for r, p in enumerate(prompts):
        dec.admit(r, p)

    # scripted overlay rows laid out before the loop: one small copy per request and cycle inside it (single-request leg)
    reps = []
    for r in range(R):
        rr, Gc, st = torch.zeros(len(plans[r]), bs, dtype=torch.long), Gs[r].cpu(), P
        for c, k in enumerate(plans[r][:ncyc + 2]):
            if st + bs + 1 >= Gc.numel():
                break
            rr[c, 1:k + 1] = Gc[st + 1:st + k + 1]
            if k + 1 < bs:
                rr[c, k + 1] = (Gc[st + k + 1] + 1) % 151000
            st += k + 1
        reps.append(rr.to(dev))

    def hook(r, blk, start, call):
        n = min(plans[r][call] + 2, bs)
        if n > 1:
            blk[0, 1:n] = reps[r][call, 1:n]

    dec.cycle(hook)
    dec.cycle(hook)                  # setup, as in the single-request leg: cycle 0 and the first steady-state cycle
    step = dec.cycle   # (cycle(ahead_ok=True), the run-ahead draft, measured no gain here: 5.99 vs 6.01 ms — the batched
    #                    cycle's launches keep ahead of the GPU as they are)
    if args.graph:
        dec.capture()
        step = dec.cycle_graph
    for _ in range(args.warmup):
        step(hook)
    ev_all = []
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    for i in range(args.steps):   # (phase marks on every --event-every-th cycle, as in the single-request leg)
        dec.events = {} if i % max(1, args.event_every) == 0 else None
        out = step(hook)
        if dec.events:
            ev_all.append(dec.events)
        tokens += sum(o[0] for o in out)
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    dec.events = None

    def avg_ms(key):   # (the first timed cycle's run-ahead draft was enqueued by the last warmup cycle: no pair for it)
        have = [e for e in ev_all if key in e and None not in e[key]]
        return sum(e[key][0].elapsed_time(e[key][1]) for e in have) / max(1, len(have))

    lm_ms, draft_ms, target_ms = avg_ms("lm_head"), avg_ms("draft"), avg_ms("target")
    n_ok = n_all = 0
    for r in range(R):
        n_ok += int((dec.output_ids[r, P:dec.start[r]] == Gs[r][P:dec.start[r]]).sum())
        n_all += dec.start[r] - P
    dt_max, tok_sum = D.reduce_timing(dt, float(tokens), device=dev)
    _, cyc_sum = D.reduce_timing(dt, float(args.steps * R), device=dev)
    # lm_head launch: weights once + the fp32 partial tiles of the 2 K parts written and read back
    part = 2 * (151936 // 16) * 4 * 1024 * 2
    traffic_b, traffic_note = None, "no PMC summary for the current kernel sources"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:    # committed per-launch HBM bytes of this very kernel, accepted while gemm_batch.hip is unchanged (hash)
        from pmc_kernels_json import BATCH_SOURCES, source_hash
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if fn.endswith("_pmc_batch_kernels.json"):
                rec = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if rec.get("kernel_source_sha256_16") == source_hash(BATCH_SOURCES) and R == 4:
                    traffic_b = rec.get("kernels", {}).get("lm_head", {}).get("hbm_bytes_per_launch")
                    traffic_note = f"profiles/{fn} (kernel sources unchanged since)"
                    break
    except Exception as e:
        traffic_note = f"PMC summary not read: {type(e).__name__}"
    kv_bytes = 20480 * (P + 16)
    hot_bytes = DRAFT_WEIGHT_BYTES + LM_HEAD_BYTES + R * kv_bytes
    return dict(
        value=tok_sum / dt_max, ms_per_step=1000.0 * dt_max / args.steps, mean_tau=tok_sum / cyc_sum,
        raw_tau1_value=cyc_sum / dt_max, lossless_fraction=n_ok / max(1, n_all),
        roofline={"kernel": "k_gemm_b<4,EPI_ARGMAX> (lm_head GEMM + fused argmax, 4 request tiles)", "bound": "hbm",
                  "achieved": LM_HEAD_BYTES / (lm_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                  "frac": LM_HEAD_BYTES / (lm_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic_b,
                  "bytes_per_launch": LM_HEAD_BYTES, "avg_ms": lm_ms,
                  "note": f"algorithmic bytes = the weights; the kernel also moves {part} B of fp32 partial tiles "
                          "(K parts meet through HBM)"},
        hot_path={"draft_plus_lm_head_ms_per_cycle": draft_ms, "target_verify_ms_per_cycle": target_ms,
                  "algorithmic_bytes_per_cycle": hot_bytes,
                  "achieved_GBps": hot_bytes / (draft_ms * 1e-3) / 1e9 if draft_ms > 0 else None,
                  "frac_of_8TBps": hot_bytes / (draft_ms * 1e-3) / 1e9 / 8000.0 if draft_ms > 0 else None},
    )


def cpu_leg(args, mean_tau):
    """The oracle (CPU restatement of the reference loop) timed on this box's host cores:
    a bounded sample of the same workload — `n` steady-state cycles at prefix 1024 (draft
    forward with a 7-row context + 15-row lm_head + argmax + 16-token target verify +
    accept), on synthetic prefix KV state, weights filled from a tiled random block."""
    from oracle import dflash_oracle as O
    from oracle.torch_target import TorchQwen3Target
    from dflash_amd.config import DFlashConfig, QWEN3_8B_DRAFT, QWEN3_8B_TARGET

    threads = torch.get_num_threads()
    t_build = time.time()
    blk = (torch.randn(1 << 22, generator=torch.Generator().manual_seed(0)) * 0.02).to(torch.bfloat16)

    def fill(shape):
        n = 1
        for d in shape:
            n *= d
        t = torch.empty(n, dtype=torch.bfloat16)
        for o in range(0, n, blk.numel()):
            m = min(blk.numel(), n - o)
            t[o:o + m] = blk[:m]
        return t.view(shape)

    cfg = DFlashConfig(**{**QWEN3_8B_DRAFT, "num_target_layers": args.target_layers})
    w = {k: (fill(s) if len(s) == 2 else torch.ones(s, dtype=torch.bfloat16)) for k, s in
         cfg.state_dict_shapes().items()}
    tgt = TorchQwen3Target(**{**QWEN3_8B_TARGET, "num_layers": args.target_layers}, dtype=torch.bfloat16,
                           attn_impl="sdpa", fill_fn=fill)
    oc = O.DraftConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                       num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                       head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size, rms_norm_eps=cfg.rms_norm_eps,
                       rope_theta=cfg.rope_theta, block_size=16, num_target_layers=cfg.num_target_layers,
                       mask_token_id=cfg.mask_token_id, target_layer_ids=list(cfg.target_layer_ids),
                       attn_impl="sdpa")
    P, bs, tau = args.prefix, 16, 7
    g = torch.Generator().manual_seed(3)
    tc, dc = tgt.new_cache(), O.ListKVCache()
    for _ in range(args.target_layers):
        tc.k.append(torch.randn(1, 8, P, 128, generator=g).to(torch.bfloat16))
        tc.v.append(torch.randn(1, 8, P, 128, generator=g).to(torch.bfloat16))
    for _ in range(cfg.num_hidden_layers):
        dc.k.append(torch.randn(1, 8, P - tau, 128, generator=g).to(torch.bfloat16))
        dc.v.append(torch.randn(1, 8, P - tau, 128, generator=g).to(torch.bfloat16))
    build_s = time.time() - t_build
    th = (torch.randn(1, tau, cfg.fc_in, generator=g)).to(torch.bfloat16)
    block = torch.randint(0, 151000, (1, bs), generator=g)
    pos = torch.arange(P + 64)[None]
    times = []
    with torch.inference_mode():
        for c in range(args.cpu_cycles + 1):
            t0 = time.perf_counter()
            noise = tgt.model.embed_tokens(block)
            hid = O.draft_forward(w, oc, target_hidden=th, noise_embedding=noise,
                                  position_ids=pos[:, dc.get_seq_length(): P + bs], cache=dc)
            block[:, 1:] = O.sample(tgt.lm_head(hid[:, -bs + 1:, :]))
            dc.crop(P)
            out = tgt(block, position_ids=pos[:, P:P + bs], past_key_values=tc, use_cache=True,
                      output_hidden_states=True)
            post = O.sample(out.logits, 0.0)
            O.acceptance_length(block, post)
            tc.crop(P)
            dc.crop(P - tau)
            th = O.extract_context_feature(out.hidden_states, oc.target_layer_ids)[:, :tau, :]
            if c:  # first pass warms caches / thread pools
                times.append(time.perf_counter() - t0)
    sec = sum(times) / len(times)
    return {"value": mean_tau / sec, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} steady-state cycles at prefix {P} (draft fwd ctx=7 + 15-row lm_head+argmax + "
                      f"16-token verify of the {args.target_layers}-layer target + accept), synthetic prefix KV, "
                      f"oracle loop on torch-CPU bf16/sdpa; {sec:.2f} s/cycle x the GPU run's mean tau; "
                      f"build {build_s:.0f}s untimed"}


def launch_ranks(n: int) -> int:
    """One fresh child process per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, as torch.distributed.run would),
    started BEFORE this process makes any GPU call and never by re-exec'ing a process that did.  Rank 0 prints the
    JSON line; the launcher only forwards exit codes (run_benchmark.sh:120-133 launches torchrun the same way)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    try:
        for p in procs:
            code = p.wait()
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in procs:     # one rank failed: the others would wait in a collective for ever
                    if q.poll() is None:
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def selftest_cpu(args, rank, world, use_pg):
    """--selftest-cpu: the N > 1 control flow of main() without kernels (gloo on the host)."""
    import torch.distributed as dist
    from dflash_amd import distributed as D
    if use_pg:
        dist.init_process_group("gloo")
        world = dist.get_world_size()
        dist.barrier()
    dt, units = D.reduce_timing(0.001 * (rank + 1), float(args.steps))
    if rank == 0:
        print(json.dumps({"metric": "accepted_tokens_per_sec", "value": None, "value_per_gpu": None, "unit": "tokens/s",
                          "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "selftest": "cpu launcher plumbing, no kernels",
                          "cycles_all_ranks": units, "config": {"parallelism": f"dp{world}",
                                                                "requests": world * args.requests_per_gpu}}),
              flush=True)
    if use_pg:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--prefix", type=int, default=1024)
    ap.add_argument("--event-every", type=int, default=4,
                    help="phase marks and the GEMM event pairs are recorded on every N-th timed cycle (an event is a "
                         "barrier packet on the stream: ten of them per cycle cost ~0.9 %% of the cycle)")
    ap.add_argument("--target-layers", type=int, default=36)
    ap.add_argument("--cpu-cycles", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--requests-per-gpu", type=int, default=1,
                    help="R > 1: R requests per GPU decode as one ragged batch sharing the weight stream "
                         "(BASELINE.json configs[2] uses 4); needs the native verify")
    ap.add_argument("--graph", action="store_true",
                    help="replay the steady-state cycle from captured hipGraphs instead of ~215 (N = 1: two graphs) / ~300 "
                         "(--requests-per-gpu > 1: three graphs) launches per cycle: the host's share of a cycle drops from "
                         "~2.2 ms to ~0.2 ms; the GPU time is the same or slightly longer (DESIGN.md), so eager is the default")
    ap.add_argument("--hf-verify", action="store_true",
                    help="verify through the HF/PyTorch target forward (round-1 configuration) instead of "
                         "dflash_amd.NativeTarget")
    ap.add_argument("--hf-prefill", action="store_true",
                    help="A/B: target prefill through the wrapped HF model (round-2 configuration; keeps both weight copies)")
    ap.add_argument("--fuse-oproj", action="store_true",
                    help="A/B: attention stage and o_proj as ONE launch (dfl_attn_head_oproj; measured slower, off by default)")
    ap.add_argument("--attn-impl", choices=["head", "fused"], default="head",
                    help="attention stage of the native verify: head = dfl_attn_head (round 2), fused = dfl_attn_fused (round 1)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launcher plumbing only (tests/test_distributed_cpu.py): ranks rendezvous over gloo, run no "
                         "kernels and report value null; never a measurement")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: this process becomes the launcher and never touches a GPU
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N does it itself)")
    use_pg = "RANK" in os.environ   # launched by torch.distributed.run or by launch_ranks (also with one rank)
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world, use_pg)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # DFL_BENCH_SHARE_GPU=1: REHEARSAL of the N-rank control flow on a one-GPU box — every rank on cuda:0, rendezvous and
    # timing scalars over gloo (RCCL refuses two ranks on one device).  Its numbers mean nothing and the line says so.
    rehearsal = os.environ.get("DFL_BENCH_SHARE_GPU") == "1"
    if rehearsal:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} GPUs are visible")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if use_pg:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)  # RCCL; timing scalars only
        world = torch.distributed.get_world_size()                    # n_gpus = the RCCL world actually seen

    res = gpu_leg(args, rank, world, dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_leg(args, res["mean_tau"])
        except Exception as e:  # the baseline is reported beside the result, never instead of it
            cpu = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        line = {
            "metric": "accepted_tokens_per_sec", "value": res["value"], "value_per_gpu": res["value"] / world,
            "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Qwen3-8B-shaped target (" + ("HF/PyTorch-ROCm verify" if args.hf_verify else
                                   ("HF prefill, " if args.hf_prefill else "NativeTarget prefill and ") + "verify on the kernels") + ") + DFlash-b16 5-layer draft, "
                                   f"block=16, temp=0, batch={args.requests_per_gpu} per GPU, prefix={args.prefix}, random-init weights, "
                                   "scripted acceptance (seeded truncated-geometric, mean tau 7.3 over the timed cycles)",
                       "target_layers": args.target_layers, "requests": world * args.requests_per_gpu,
                       "target_verify": "hf" if args.hf_verify else "native", "parallelism": f"dp{world}"},
            "mean_acceptance_length": res["mean_tau"], "raw_tau1_value": res["raw_tau1_value"],
            "lossless_fraction": res["lossless_fraction"], "roofline": res["roofline"], "hot_path": res["hot_path"],
            "ttft_side": res.get("ttft_side"), "host_side": res.get("host_side"), "cpu_baseline": cpu,
        }
        if rehearsal:
            line["rehearsal"] = "DFL_BENCH_SHARE_GPU=1: all ranks shared cuda:0 over gloo — control flow only, not a measurement"
        print(json.dumps(line), flush=True)
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
