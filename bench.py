#!/usr/bin/env python3
"""bench.py — accepted tokens/s of the DFlash decode cycle on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json configs[1]): Qwen3-8B-shaped target (36 layers) + 5-layer DFlash-b16
draft, seeded random-init bf16 weights, one request per GPU, 1024 random prompt ids, block 16, temperature 0.  A *step*
is one decode cycle: draft block forward + fused lm_head/argmax, target verify (NativeTarget: the same HIP kernels;
`--hf-verify` sends it through the HF forward instead), posterior argmax + accept/commit.  `value` is that leg.
The same line carries a `batch4` object: BASELINE configs[2]'s per-GPU leg — four requests per GPU as one ragged batch
sharing every weight byte (dflash_amd.batch) — timed right after the headline with the same models, K steps as well.

Other workloads (builder-run lines, `--workload`):
  llama31-8b       BASELINE configs[3]: Llama-3.1-8B shapes (32 layers, FFN 14336, V 128256, llama3 RoPE), temperature
                   0.7 — the stochastic acceptance path (materialised 16 x V logits, softmax + multinomial, accept iff
                   the draft token equals the sampled one)
  qwen3-30b-a3b    BASELINE configs[4]: Qwen3-Coder-30B-A3B shapes (48 layers, 128 experts top-8, sparse-MoE target) under
                   the dynamic block-size loop of benchmark_dynamic_schedule.py (EWMAPerformanceScheduler over 8,12,16,
                   the reference's default flags); a step is one cycle of that loop, its wall time fed back

Random weights never agree (tau == 1), so acceptance is scripted as SURVEY.md §8d prescribes: the target's greedy
continuation G is known beforehand and, after the fully timed draft forward + argmax, the draft tokens are overwritten
with G[start+1 : start+k] followed by a wrong id, k drawn from a seeded truncated-geometric law whose mean tau matches
the published 7.3 (the K timed cycles are conditioned on it; the overlay rows are laid out before the timed loop, inside
it they cost one 16-id copy per cycle).  `value` is committed tokens / wall time over all ranks; `raw_tau1_value` is the
same cycles counted at tau = 1.  Setup before the W warmup steps: prefill, cycle 0 (it carries the one-off projection of
the prompt's 1024 context rows into the draft cache) and the first steady-state cycle (one-off code-object loads, ~60
ms) — so that even `--warmup 0` times steady-state cycles.

A plainly random-init bf16 target has near-zero top-2 logit margins: its argmax flips between a 1-token and a 16-token
forward (measured: 65 % of tokens reproduced), so no greedy tape survives the verify.  The synthetic target is therefore
given a large-margin greedy rule WITHOUT changing its architecture, byte count or FLOPs: seeded random weights,
embedding std 1.0, o_proj/down_proj scaled by 0.02 (the residual stream stays embedding-dominated through all layers),
lm_head = 0.02 * embedding rows permuted by a seeded single-cycle permutation.  Its greedy next token is perm[token]
with a logit margin ~70, so G is a closed-form walk and `lossless_fraction` (committed ids == G) must read 1.0 — at
temperature 0.7 too: softmax(logits / 0.7) of a margin-70 row is one-hot, the draw still runs.

Steady-state cycles are replayed from hipGraphs (DecodeSession.capture / BatchedDecoder.capture; `--eager`: ~215 / ~300
ctypes launches per cycle instead): same GPU time, the host's share drops from ~2.2 ms to ~0.2 ms per cycle (DESIGN.md
section 5: what rounds 2-3 read as "kernels are slower once a graph exists" was the driver clearing the VRAM that
torch.cuda.graph()'s empty_cache() hands back).  Every --event-every-th cycle runs eagerly with HIP events around its
phases and GEMMs.

N > 1: every rank runs its own request(s) (weak scaling, no collective on the accept path).  Ranks rendezvous and
barrier over gloo; the RCCL communicator (torch's "nccl") is created only AFTER the timed regions, for the world size
and the all-reduce of the timing scalars — no RCCL stream or proxy thread is alive while cycles are timed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ----------------------------------------------------------------------------------------------- workloads
def workload_spec(name):
    from dflash_amd.config import QWEN3_8B_DRAFT, QWEN3_8B_TARGET
    if name == "qwen3-8b":
        return dict(kind="qwen3", target=dict(QWEN3_8B_TARGET), draft=dict(QWEN3_8B_DRAFT), temperature=0.0, schedule=None,
                    label="Qwen3-8B-shaped target", baseline_config="configs[1]")
    if name == "llama31-8b":   # draft: the Qwen3-style module of model/dflash.py at the target's widths (depth unknown offline: 5)
        return dict(kind="llama",
                    target=dict(vocab_size=128256, hidden_size=4096, num_layers=32, num_heads=32, num_kv_heads=8, head_dim=128,
                                intermediate_size=14336, rope_theta=500000.0),
                    draft=dict(hidden_size=4096, num_hidden_layers=5, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
                               intermediate_size=14336, vocab_size=128256, num_target_layers=32, block_size=16, rope_theta=500000.0,
                               rms_norm_eps=1e-5, mask_token_id=128255),
                    temperature=0.7, schedule=None, label="Llama-3.1-8B-shaped target", baseline_config="configs[3]")
    if name == "qwen3-30b-a3b":
        return dict(kind="qwen3moe",
                    target=dict(vocab_size=151936, hidden_size=2048, num_layers=48, num_heads=32, num_kv_heads=4, head_dim=128,
                                intermediate_size=6144, moe_intermediate_size=768, num_experts=128, top_k=8, rope_theta=1e7),
                    draft=dict(hidden_size=2048, num_hidden_layers=5, num_attention_heads=32, num_key_value_heads=4, head_dim=128,
                               intermediate_size=6144, vocab_size=151936, num_target_layers=48, block_size=16, rope_theta=1e7,
                               mask_token_id=151669),
                    temperature=0.0, schedule=(8, 12, 16), label="Qwen3-Coder-30B-A3B-shaped sparse-MoE target (128 experts, top-8)",
                    baseline_config="configs[4]")
    raise SystemExit(f"unknown workload {name}")


def make_hf_target(spec, dev, layers):
    """The HF class the reference would load for this workload (benchmark.py:401), random-init on the GPU, with the
    large-margin greedy rule of the module docstring imposed (same architecture, bytes and FLOPs)."""
    from dflash_amd.synthetic import impose_greedy_walk, make_hf_qwen3
    T = {**spec["target"], "num_layers": layers}
    if spec["kind"] == "qwen3":
        m = make_hf_qwen3(T, dev)
    else:
        import transformers as tf
        if spec["kind"] == "llama":
            cfg = tf.LlamaConfig(vocab_size=T["vocab_size"], hidden_size=T["hidden_size"], intermediate_size=T["intermediate_size"],
                                 num_hidden_layers=layers, num_attention_heads=T["num_heads"], num_key_value_heads=T["num_kv_heads"],
                                 head_dim=128, rms_norm_eps=1e-5, max_position_embeddings=131072, tie_word_embeddings=False,
                                 attention_bias=False, mlp_bias=False,
                                 rope_parameters={"rope_type": "llama3", "rope_theta": T["rope_theta"], "factor": 8.0,
                                                  "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                                  "original_max_position_embeddings": 8192})
            cls = tf.LlamaForCausalLM
        else:
            cfg = tf.Qwen3MoeConfig(vocab_size=T["vocab_size"], hidden_size=T["hidden_size"], intermediate_size=T["intermediate_size"],
                                    moe_intermediate_size=T["moe_intermediate_size"], num_hidden_layers=layers,
                                    num_attention_heads=T["num_heads"], num_key_value_heads=T["num_kv_heads"], head_dim=128,
                                    num_experts=T["num_experts"], num_experts_per_tok=T["top_k"], decoder_sparse_step=1,
                                    norm_topk_prob=True, max_position_embeddings=40960, rms_norm_eps=1e-6, tie_word_embeddings=False,
                                    rope_parameters={"rope_type": "default", "rope_theta": T["rope_theta"]}, mlp_only_layers=[])
            cls = tf.Qwen3MoeForCausalLM
        cfg._attn_implementation = "sdpa"
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.bfloat16)
        try:
            with torch.device(dev):
                m = cls(cfg).eval()
        finally:
            torch.set_default_dtype(prev)
    return m, impose_greedy_walk(m, seed=1234)


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


def barrier():
    if torch.distributed.is_initialized():
        torch.distributed.barrier()     # gloo: a host rendezvous, no device stream


def pmc_traffic(pattern, batch=False):
    """Committed per-launch HBM bytes (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes) and rocprof durations of the
    kernels, accepted only while the hash of the kernels' sources stored beside them still matches — PMC passes cannot
    run inside the timed bench (rocprofv3 --pmc serialises and slows the run); a changed kernel reports null until
    re-profiled (scripts/pmc_kernels_json.py)."""
    out, note = {}, "no PMC summary for the current kernel sources"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import pmc_kernels_json as PK
        want = PK.source_hash(PK.BATCH_SOURCES if batch else None)
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if fn.endswith(pattern):
                rec = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if rec.get("kernel_source_sha256_16") == want:
                    out = rec.get("kernels", {})
                    note = f"profiles/{fn} (kernel sources unchanged since)"
                    break
    except Exception as e:   # never let bookkeeping break the measurement
        note = f"PMC summary not read: {type(e).__name__}"
    return out, note


# ----------------------------------------------------------------------------------------------- models
def build_models(args, spec, rank, dev):
    from dflash_amd import DFlashConfig, DFlashDraftModel
    torch.manual_seed(0)
    t0 = time.time()
    layers = args.target_layers or spec["target"]["num_layers"]
    target, perm = make_hf_target(spec, dev, layers)
    if not args.hf_verify:
        from dflash_amd import NativeTarget
        # verify AND prefill on the kernels; the wrapped HF model is dropped after packing (one copy of
        # the target's weights in memory) unless --hf-prefill asks for the round-2 configuration
        target = NativeTarget(target, attn_impl=args.attn_impl, keep_hf=args.hf_prefill,
                              prefill="hf" if args.hf_prefill else "native")
        target.fuse_oproj = args.fuse_oproj
    cfg = DFlashConfig(**{**spec["draft"], "num_target_layers": layers})
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
    # The dropped HF weights (~17 GB at 8B) sit in torch's caching allocator.  Hand them back to the driver NOW: the
    # driver clears released VRAM in the background for a few hundred ms, and every kernel that runs meanwhile is 2-4 %
    # slower — far from any timed region here, and nothing later calls empty_cache().
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    log(f"[rank {rank}] models ready in {time.time() - t0:.1f}s")
    n = lambda s: s[0] * s[1] if len(s) == 2 else s[0]   # noqa: E731
    T = spec["target"]
    H, V = T["hidden_size"], T["vocab_size"]
    q_dim, kv_dim = T["num_heads"] * 128, T["num_kv_heads"] * 128
    meta = dict(layers=layers, H=H, V=V, lm_head_bytes=V * H * 2,
                draft_weight_bytes=2 * sum(n(s) for s in cfg.state_dict_shapes().values()),
                kv_row_bytes_draft=cfg.num_hidden_layers * 2 * cfg.kv_dim * 2,
                kv_row_bytes_target=layers * 2 * kv_dim * 2,
                attn_bytes_per_layer=((q_dim + 2 * kv_dim) * H + H * q_dim) * 2)
    if spec["kind"] == "qwen3moe":
        meta.update(expert_gu_bytes=2 * T["moe_intermediate_size"] * H * 2, expert_bytes=3 * T["moe_intermediate_size"] * H * 2,
                    router_bytes=T["num_experts"] * H * 2)
    else:
        meta.update(gate_up_bytes=2 * T["intermediate_size"] * H * 2, mlp_bytes=3 * T["intermediate_size"] * H * 2)
    return target, draft, cfg, perm, meta


def diag_state(dev):
    """diagnostic knob (scripts/probes/stream_ab_bench*.sh; never set by the driver): put the process
    into one of the states rounds 2-3 blamed for slower kernels before anything is timed"""
    extra = os.environ.get("DFL_BENCH_EXTRA_STREAM", "")
    if not extra:
        return
    keep = diag_state.__dict__.setdefault("_keep", {})
    if extra in ("idle", "used"):
        keep["s2"] = torch.cuda.Stream()
    if extra == "used":
        with torch.cuda.stream(keep["s2"]):
            keep["t"] = torch.zeros(64, device=dev) + 1
        keep["s2"].synchronize()
    if extra in ("graph1", "graph1_del"):      # a one-kernel torch graph, captured and never replayed
        g1, y = torch.cuda.CUDAGraph(), torch.zeros(64, device=dev)
        with torch.cuda.graph(g1):
            y += 1
        keep["g1"] = (g1, y)
        if extra == "graph1_del":
            del keep["g1"], g1
    if extra in ("emptycache", "graph1_del"):  # what torch.cuda.graph() does before it begins a capture, without the capture
        import gc
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
    if extra == "refill":   # GB of freed blocks back into torch's cache, then emptied: the driver's clear right before the timing
        blocks = [torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1)
                  for _ in range(int(os.environ.get("DFL_BENCH_REFILL_GB", "17")))]
        torch.cuda.synchronize()
        del blocks
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    time.sleep(float(os.environ.get("DFL_BENCH_EXTRA_SLEEP", "0")))


# ----------------------------------------------------------------------------------------------- one request per GPU
def single_leg(args, spec, rank, dev, target, draft, cfg, perm, meta):
    from dflash_amd.generate import DecodeSession, cuda_time
    from dflash_amd.synthetic import greedy_walk
    bs, P, V = 16, args.prefix, meta["V"]
    T = spec["temperature"] if args.temperature is None else args.temperature
    sched_c = spec["schedule"] if args.schedule is None else tuple(int(x) for x in args.schedule.split(",") if x)
    native = not args.hf_verify
    prompt = torch.randint(0, V - 1000, (1, P), generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    ncyc = args.warmup + args.steps + 2
    plan = tau_plan(2 + args.warmup, args.steps, bs, seed=100 + rank)
    need = sum(k + 1 for k in plan[:ncyc]) + 2 * bs
    mask_id = cfg.mask_token_id
    # ---- the target's greedy continuation G in closed form: G[p+1] = perm[G[p]]
    G = greedy_walk(perm, prompt, need + 2 * bs).to(dev)

    if sched_c:
        # block sizes come from the scheduler, so a cycle's start is not known beforehand: the overlay is written from G
        # at the cycle's own start (two small device copies per cycle instead of one)
        def hook(blk, start, call):
            k = min(plan[call], blk.shape[1] - 1)
            if k > 0:
                blk[0, 1:k + 1] = G[start + 1:start + k + 1]
            if k + 1 < blk.shape[1]:
                blk[0, k + 1] = (G[start + k + 1] + 1) % (V - 1000)
    else:
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
                rep_rows[c, k + 1] = (Gc[st + k + 1] + 1) % (V - 1000)
            st += k + 1
        rep_rows = rep_rows.to(dev)

        def hook(blk, start, call):
            n = min(plan[call] + 2, bs)
            if n > 1:
                blk[0, 1:n] = rep_rows[call, 1:n]

    sched = None
    if sched_c:
        from dflash_amd.scheduler import EWMAPerformanceScheduler
        # the reference's default flags (benchmark_dynamic_schedule.py:462-481)
        sched = EWMAPerformanceScheduler(candidates=list(sched_c), scheduler_mode="ewma", warmup_cycles=2, ewma_alpha=0.10,
                                         switch_margin=0.05, required_streak=2, cooldown_cycles=2, probe_interval=12,
                                         low_accept_threshold=0.35, low_accept_streak=2, adl_rho=0.30, adl_delta=2.0,
                                         adl_k_min=min(sched_c), adl_k_max=max(sched_c), adl_neighborhood=4)
    for rep in range(2):   # rep 0 pays the one-off costs (code-object loads, allocator, HF lazy init): rep 1 is reported
        s = DecodeSession(draft, target, prompt, mask_token_id=mask_id, max_new_tokens=need, max_block_size=bs,
                          stop_token_ids=None, temperature=T, draft_temperature=T if sched else 0.0, draft_token_hook=hook)
        torch.cuda.synchronize()
        t_pf = time.perf_counter()
        s.prefill()
        torch.cuda.synchronize()
        t_c0 = time.perf_counter()
        s.cycle(bs)                  # cycle 0: carries the one-off draft-context prefill of the prompt rows
        torch.cuda.synchronize()
        hf_pf = args.hf_verify or args.hf_prefill or not getattr(target, "native_prefill", False)
        ttft_side = {"target_prefill_ms": 1e3 * (t_c0 - t_pf), "cycle0_ms": 1e3 * (time.perf_counter() - t_c0),
                     "note": "second request on warm code (outside the timed region): the target prefill runs "
                             + ("through the wrapped HF model" if hf_pf else
                                "on the kernels (csrc/prefill.hip: MFMA GEMMs on the packed weights, causal attention, norm / RoPE / cache write)")
                             + f"; cycle 0 = projection of the {P} prompt context rows into the draft cache "
                             "(model/dflash.py:73-85) + one decode cycle"}
    cyc, used = [0], []

    def step(instr=False, allow_graph=False):
        """one cycle of the loop this workload times; returns tau"""
        if sched is None:
            if allow_graph and not instr and getattr(s, "_graph_bs", None):
                return s.cycle_graph(bs).tau
            return s.cycle(bs, ahead_ok=True).tau
        # benchmark_dynamic_schedule.py:319-379: block size from the scheduler, the cycle's wall time fed back
        t_c = cuda_time()
        chosen = sched.select(cyc[0])
        b = max(1, min(chosen, s.max_length - s.start))
        # replay of the size's graph pair (DecodeSession.capture_sizes) unless this cycle carries event pairs
        r = s.cycle_sized(b) if (allow_graph and not instr) else s.cycle(b)
        sched.update(tau=r.tau, cycle_s=cuda_time() - t_c, effective_bs=b, cycle_idx=cyc[0], l_gen=float(b))
        used.append(b)
        cyc[0] += 1
        return r.tau

    step()       # first steady-state cycle: one-off code-object loads (60 ms) — setup, like cycle 0
    for _ in range(args.warmup):
        step()
    use_graph = (not args.eager) and T < 1e-5 and native and args.attn_impl == "head" and not args.fuse_oproj \
        and (sched is None or (min(sched_c) >= 2 and max(sched_c) <= 16))
    if use_graph and sched is None:   # the steady-state cycle as two hipGraph replays (DecodeSession.capture), captured before the timed region
        s.capture(bs)
    elif use_graph:                   # a scheduler picks the size every cycle: one pair of graphs per candidate size, no run-ahead draft
        s.capture_sizes(sched_c)

    ev_all, lm_ev, gu_ev, moe_log = [], [], [], None
    E = max(1, args.event_every)
    n_instr = (args.steps + E - 1) // E
    # HIP events right around the lm_head GEMM launch itself (dfl_gemm_argmax_timed records them on the launch stream) and
    # around ONE gate/up GEMM launch of the target verify per instrumented cycle (layer = cycle index mod layers):
    # created, and recorded once so that their handles exist, before the timed region
    for lst in (lm_ev, gu_ev):
        for _ in range(n_instr):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            lst.append((a, b))
    is_moe = native and getattr(target, "is_moe", False)
    if is_moe:
        moe_log = torch.zeros(n_instr, dtype=torch.int32, device=dev)
    used.clear()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    draft.lm_head_events_log = []
    # Instrumented cycles are SAMPLED (every --event-every-th timed cycle): an event record is a barrier + timestamp packet
    # on the stream, 2.6 us by the kernel trace, and ten per cycle (phase marks, the lm_head pair, the gate/up pair) were
    # ~0.9 % of the cycle they were meant to observe.  The cycle after an instrumented one only collects the pairs of
    # the draft forward that was enqueued ahead of it.  With graph replay the instrumented cycles run eagerly.
    s.host_times = []
    for i in range(args.steps):
        instr = i % E == 0
        j = i // E
        s.events = {} if (instr or (i - 1) % E == 0) else None
        s.record_events = instr
        draft.lm_head_events = lm_ev[j] if instr else None
        if native:
            target.gu_events = (i % meta["layers"], gu_ev[j][0], gu_ev[j][1]) if (instr and not is_moe) else None
            target.moe_events = (i % meta["layers"], gu_ev[j][0], gu_ev[j][1], moe_log[j:j + 1]) if (instr and is_moe) else None
        tokens += step(instr, allow_graph=use_graph)
        if s.events:
            ev_all.append(s.events)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    s.events, s.record_events = None, True
    host_enq = sum(a for a, _ in s.host_times) / max(1, len(s.host_times))
    host_wait = sum(b for _, b in s.host_times) / max(1, len(s.host_times))
    s.host_times = None
    draft.lm_head_events = None
    if native:
        target.gu_events = target.moe_events = None

    def avg_ms(key):   # (the first timed cycle's run-ahead draft was enqueued by the last warmup cycle: no pair for it)
        have = [e for e in ev_all if key in e and None not in e[key]]
        return sum(e[key][0].elapsed_time(e[key][1]) for e in have) / max(1, len(have))

    draft_ms, target_ms = avg_ms("draft"), avg_ms("target")
    lm_used = draft.lm_head_events_log or lm_ev   # (the pairs actually recorded)
    draft.lm_head_events_log = None
    lm_ms = sum(a.elapsed_time(b) for a, b in lm_used) / len(lm_used)
    gu_ms = sum(a.elapsed_time(b) for a, b in gu_ev) / len(gu_ev) if native else None
    # committed ids must be the target's own greedy continuation (losslessness)
    n_ok = int((s.output_ids[0, P:s.start] == G[P:s.start]).sum())
    lossless = n_ok / max(1, s.start - P)

    traffic, traffic_note = pmc_traffic("_pmc_kernels.json")
    ev_note = ("achieved/avg_ms from HIP events recorded on the launch stream right before and right after the GEMM launch, on "
               "every --event-every-th timed cycle (an event pair also spans the dependent launch boundary, ~2-3 us: "
               "rocprof_avg_ms is the kernel's own duration in the committed rocprofv3 --kernel-trace --stats run of this "
               "command, same kernel sources); traffic = 2*FETCH_SIZE+WRITE_SIZE bytes per launch: " + traffic_note)

    def entry(kernel, key, nbytes, ms, extra=None):
        k = traffic.get(key, {}) if spec["kind"] == "qwen3" else {}
        us = k.get("rocprof_avg_us")
        e = {"kernel": kernel, "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
             "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0, "traffic": k.get("hbm_bytes_per_launch"), "bytes_per_launch": nbytes,
             "avg_ms": ms, "note": ev_note}
        if us:
            e.update(rocprof_avg_ms=us / 1e3, frac_rocprof=nbytes / (us * 1e-6) / 8e12)
        if extra:
            e.update(extra)
        return e

    lm_entry = entry("k_gemm<1,false,EPI_ARGMAX> (lm_head GEMM + fused argmax; 2 launches per cycle)", "lm_head",
                     meta["lm_head_bytes"], lm_ms)
    if gu_ms and is_moe:
        # active-expert bytes: the launch streams the gate/up weights of the experts the block's rows x top-8 picked
        n_act = moe_log.float().mean().item()
        roofline = entry("k_moe_gate_up (gate/up + SiLU of every ACTIVE expert of a layer in one launch; 1 launch per target "
                         "layer and cycle)", "moe_gate_up", n_act * meta["expert_gu_bytes"], gu_ms,
                         {"active_experts_mean": n_act, "bytes_per_expert": meta["expert_gu_bytes"],
                          "bytes_note": "algorithmic bytes per launch = active experts of the instrumented layer (read back "
                                        "from the routing kernel's count, one int per instrumented cycle) x 2 x 768 x 2048 x 2 B",
                          "also": [lm_entry]})
    elif gu_ms:
        # the roofline object names the kernel with the largest share of the timed cycle: the gate/up GEMM
        # with the fused SiLU*up epilogue, one launch per layer of target and draft; the lm_head GEMM rides beside it
        roofline = entry(f"k_gemm<1,false,EPI_SILU> (gate/up GEMM + SiLU*up epilogue; {meta['layers'] + cfg.num_hidden_layers} "
                         "launches per cycle, the largest share of the cycle)", "gate_up", meta["gate_up_bytes"], gu_ms,
                         {"also": [lm_entry]})
    else:
        roofline = lm_entry
    hot_bytes = meta["draft_weight_bytes"] + meta["lm_head_bytes"] + meta["kv_row_bytes_draft"] * (P + 16)
    return dict(
        dt=dt, tokens=float(tokens), cycles=float(args.steps), lossless_fraction=lossless, roofline=roofline,
        hot_path={"draft_plus_lm_head_ms_per_cycle": draft_ms, "target_verify_ms_per_cycle": target_ms,
                  "algorithmic_bytes_per_cycle": hot_bytes,
                  # (a run too short to see a run-ahead draft's event pair — it is collected one cycle later — has none)
                  "achieved_GBps": hot_bytes / (draft_ms * 1e-3) / 1e9 if draft_ms > 0 else None,
                  "frac_of_8TBps": hot_bytes / (draft_ms * 1e-3) / 1e9 / 8000.0 if draft_ms > 0 else None},
        ttft_side=ttft_side,
        host_side={"enqueue_ms_per_cycle": 1e3 * host_enq, "poll_wait_ms_per_cycle": 1e3 * host_wait,
                   "mode": ("hipGraph replay (two graphs per cycle); every --event-every-th cycle eager" if use_graph
                            else "eager launches"),
                   "note": "host share of a timed cycle (time.perf_counter inside DecodeSession.cycle / cycle_graph): enqueueing the "
                           "cycle's launches (eager: Python + ctypes for ~215 launches, the NEXT cycle's run-ahead draft included; "
                           "replay: two graph launches), then polling the pinned result word; the GPU is the bottleneck while "
                           "enqueue < ms_per_step and the poll wait is the rest of it"},
        temperature=T, schedule=list(sched_c) if sched_c else None,
        used_block_sizes={str(b): used.count(b) for b in sorted(set(used))} if used else None,
    )


# ----------------------------------------------------------------------------------------------- R requests per GPU
def batched_leg(args, rank, dev, draft, target, perm, cfg, meta, R):
    """R > 1 requests per GPU (BASELINE.json configs[2]: 4 requests per GPU): the R requests of
    this rank advance together, one pass over the weights per cycle (dflash_amd.batch)."""
    from dflash_amd.batch import BatchedDecoder
    from dflash_amd.synthetic import greedy_walk
    bs, P, V = 16, args.prefix, meta["V"]
    ncyc = args.warmup + args.steps + 2
    plans = [tau_plan(2 + args.warmup, args.steps, bs, seed=100 + rank * 16 + r) for r in range(R)]
    need = max(sum(k + 1 for k in pl[:ncyc]) for pl in plans) + 2 * bs
    prompts = [torch.randint(0, V - 1000, (1, P), generator=torch.Generator().manual_seed(1 + rank * 16 + r)).to(dev)
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
                rr[c, k + 1] = (Gc[st + k + 1] + 1) % (V - 1000)
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
    use_graph = not args.eager
    if use_graph:
        dec.capture()
        step = dec.cycle_graph
    for _ in range(args.warmup):
        step(hook)
    ev_all = []
    barrier()
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
    barrier()
    dt = time.perf_counter() - t0
    dec.events = None

    def avg_ms(key):
        have = [e for e in ev_all if key in e and None not in e[key]]
        return sum(e[key][0].elapsed_time(e[key][1]) for e in have) / max(1, len(have))

    lm_ms, draft_ms, target_ms = avg_ms("lm_head"), avg_ms("draft"), avg_ms("target")
    n_ok = n_all = 0
    for r in range(R):
        n_ok += int((dec.output_ids[r, P:dec.start[r]] == Gs[r][P:dec.start[r]]).sum())
        n_all += dec.start[r] - P
    traffic, traffic_note = pmc_traffic("_pmc_batch_kernels.json", batch=True)
    k = traffic.get("lm_head", {}) if R == 4 else {}
    us = k.get("rocprof_avg_us")
    roofline = {"kernel": f"dfl_k_gemm_r<4,1,1,16,2,EPI_ARGMAX> (lm_head GEMM + fused argmax, {R} request tiles on one pass over "
                          "the weights: a wave owns a column tile over the whole K, the activations stream through an LDS ring)",
                "bound": "hbm", "achieved": meta["lm_head_bytes"] / (lm_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": meta["lm_head_bytes"] / (lm_ms * 1e-3) / 1e9 / 8000.0, "traffic": k.get("hbm_bytes_per_launch"),
                "bytes_per_launch": meta["lm_head_bytes"], "avg_ms": lm_ms,
                "note": "avg_ms: HIP event pair around the draft's lm_head launch pair (the GEMM + its 4 us finish kernel), every "
                        "--event-every-th cycle; traffic: " + traffic_note}
    if us:
        roofline.update(rocprof_avg_ms=us / 1e3, frac_rocprof=meta["lm_head_bytes"] / (us * 1e-6) / 8e12)
    hot_bytes = meta["draft_weight_bytes"] + meta["lm_head_bytes"] + R * meta["kv_row_bytes_draft"] * (P + 16)
    cyc_bytes = (hot_bytes + meta["lm_head_bytes"] + meta["layers"] * (meta["attn_bytes_per_layer"] + meta.get("mlp_bytes", 0))
                 + R * meta["kv_row_bytes_target"] * (P + 16))
    return dict(
        dt=dt, tokens=float(tokens), cycles=float(args.steps * R), steps=args.steps, lossless_fraction=n_ok / max(1, n_all),
        roofline=roofline,
        hot_path={"draft_plus_lm_head_ms_per_cycle": draft_ms, "target_verify_ms_per_cycle": target_ms,
                  "algorithmic_bytes_per_cycle": hot_bytes,
                  "achieved_GBps": hot_bytes / (draft_ms * 1e-3) / 1e9 if draft_ms > 0 else None,
                  "frac_of_8TBps": hot_bytes / (draft_ms * 1e-3) / 1e9 / 8000.0 if draft_ms > 0 else None},
        cycle_bytes=cyc_bytes, mode="hipGraph replay (three graphs per cycle)" if use_graph else "eager launches")


def cpu_leg(args, spec, mean_tau):
    """The oracle (CPU restatement of the reference loop) timed on this box's host cores:
    a bounded sample of the same workload — `n` steady-state cycles at prefix 1024 (draft
    forward with a 7-row context + 15-row lm_head + argmax + 16-token target verify +
    accept), on synthetic prefix KV state, weights filled from a tiled random block."""
    from oracle import dflash_oracle as O
    from oracle.torch_target import TorchQwen3Target
    from dflash_amd.config import DFlashConfig

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

    layers = args.target_layers or spec["target"]["num_layers"]
    cfg = DFlashConfig(**{**spec["draft"], "num_target_layers": layers})
    w = {k: (fill(s) if len(s) == 2 else torch.ones(s, dtype=torch.bfloat16)) for k, s in
         cfg.state_dict_shapes().items()}
    tgt = TorchQwen3Target(**{**spec["target"], "num_layers": layers}, dtype=torch.bfloat16,
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
    for _ in range(layers):
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
                      f"16-token verify of the {layers}-layer target + accept), synthetic prefix KV, "
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
    ap.add_argument("--workload", choices=["qwen3-8b", "llama31-8b", "qwen3-30b-a3b"], default="qwen3-8b",
                    help="qwen3-8b = BASELINE configs[1] (+ the configs[2] leg as `batch4`); llama31-8b = configs[3] (T = 0.7); "
                         "qwen3-30b-a3b = configs[4] (sparse-MoE target, dynamic block-size schedule)")
    ap.add_argument("--temperature", type=float, default=None, help="override the workload's temperature")
    ap.add_argument("--schedule", type=str, default=None,
                    help="candidate block sizes of the dynamic schedule, e.g. 8,12,16 ('' = fixed block 16); default: the workload's")
    ap.add_argument("--event-every", type=int, default=4,
                    help="phase marks and the GEMM event pairs are recorded on every N-th timed cycle (an event is a "
                         "barrier packet on the stream: ten of them per cycle cost ~0.9 %% of the cycle)")
    ap.add_argument("--target-layers", type=int, default=0, help="0 = the workload's own depth")
    ap.add_argument("--cpu-cycles", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--requests-per-gpu", type=int, default=1,
                    help="R > 1: the HEADLINE becomes R requests per GPU as one ragged batch sharing the weight stream "
                         "(BASELINE.json configs[2] uses 4); needs the native verify.  The default line carries that leg at "
                         "R = 4 as the `batch4` object anyway")
    ap.add_argument("--no-batch4", action="store_true", help="skip the configs[2] leg (the `batch4` object) of the default line")
    ap.add_argument("--eager", action="store_true",
                    help="~215 (N = 1) / ~300 (ragged batch) ctypes launches per cycle instead of hipGraph replays: same GPU time, "
                         "the host's share of a cycle is ~2.2 ms instead of ~0.2 ms (DESIGN.md section 5)")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)   # (round-3 flag: replay is the default now)
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
    # DFL_BENCH_SHARE_GPU=1: REHEARSAL of the N-rank control flow on a one-GPU box — every rank on cuda:0 (RCCL refuses two
    # ranks on one device: the scalars are then reduced over gloo).  Its numbers mean nothing and the line says so.
    rehearsal = os.environ.get("DFL_BENCH_SHARE_GPU") == "1"
    if rehearsal:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} GPUs are visible")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if use_pg:
        # rendezvous + barriers around the timed regions over gloo (host side).  The RCCL communicator comes AFTER the
        # timing (below): its streams / proxy threads do not exist while cycles are timed.  (distributed.py:18-22: the
        # reference inits NCCL up front — it has no kernels of its own to protect.)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("gloo")

    spec = workload_spec(args.workload)
    R = args.requests_per_gpu
    if R > 1 and (args.hf_verify or args.workload != "qwen3-8b"):
        raise SystemExit("--requests-per-gpu needs the native verify and the default workload")
    target, draft, cfg, perm, meta = build_models(args, spec, rank, dev)
    diag_state(dev)
    b4 = None
    if R > 1:
        res = batched_leg(args, rank, dev, draft, target, perm, cfg, meta, R)
    else:
        res = single_leg(args, spec, rank, dev, target, draft, cfg, perm, meta)
        if args.workload == "qwen3-8b" and not args.no_batch4 and not args.hf_verify:
            b4 = batched_leg(args, rank, dev, draft, target, perm, cfg, meta, 4)

    # ---- all timing is done: NOW the RCCL communicator (the world size it sees is the one reported) and the reductions
    from dflash_amd import distributed as D
    grp, rccl_note = None, None
    if use_pg:
        world = torch.distributed.get_world_size()
    if use_pg and not rehearsal:
        try:
            grp = torch.distributed.new_group(backend="nccl", device_id=dev)   # RCCL
            world = torch.distributed.get_world_size(grp)
            rccl_note = "RCCL group created after the timed regions; timing scalars all-reduced over it"
        except Exception as e:   # the measurement stands without it: the scalars then travel over the gloo group
            grp, rccl_note = None, f"RCCL group not created ({type(e).__name__}: {e}); timing scalars reduced over gloo"

    def reduce(leg):
        dt_max, tok = D.reduce_timing(leg["dt"], leg["tokens"], device=dev, group=grp)
        _, cyc = D.reduce_timing(leg["dt"], leg["cycles"], device=dev, group=grp)
        return dt_max, tok, cyc

    dt_max, tok_sum, cyc_sum = reduce(res)
    value, mean_tau = tok_sum / dt_max, tok_sum / cyc_sum
    b4_obj = None
    if b4 is not None:
        bd, bt, bc = reduce(b4)
        ms = 1000.0 * bd / b4["steps"]
        b4_obj = {"workload": "BASELINE configs[2], per-GPU leg: 4 requests per GPU decode as ONE ragged batch sharing every weight "
                              "byte of draft, lm_head and target verify (dflash_amd.batch; benchmark.py:445-470 runs them one after "
                              "another); same models, prompts of 1024 ids, block 16, T = 0, scripted mean tau 7.3 per request",
                  "value": bt / bd, "value_per_gpu": bt / bd / world, "unit": "tokens/s", "requests": 4 * world,
                  "ms_per_step": ms, "steps": b4["steps"], "mean_acceptance_length": bt / bc,
                  "lossless_fraction": b4["lossless_fraction"], "cycle_algorithmic_bytes": b4["cycle_bytes"],
                  "cycle_frac_of_8TBps": b4["cycle_bytes"] / (ms * 1e-3) / 8e12,
                  "roofline": b4["roofline"], "hot_path": b4["hot_path"], "mode": b4["mode"]}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            if spec["kind"] != "qwen3":
                raise RuntimeError("the CPU oracle's target is the dense Qwen3 stack: reported for the default workload only")
            cpu = cpu_leg(args, spec, mean_tau)
        except Exception as e:  # the baseline is reported beside the result, never instead of it
            cpu = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        T = res.get("temperature", 0.0)
        line = {
            "metric": "accepted_tokens_per_sec", "value": value, "value_per_gpu": value / world,
            "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{spec['label']} (BASELINE {spec['baseline_config'] if R == 1 else 'configs[2] per-GPU leg'}; "
                                   + ("HF/PyTorch-ROCm verify" if args.hf_verify else
                                      ("HF prefill, " if args.hf_prefill else "NativeTarget prefill and ") + "verify on the kernels")
                                   + f") + DFlash-b16 {cfg.num_hidden_layers}-layer draft, "
                                   + (f"dynamic block size over {res['schedule']} (EWMAPerformanceScheduler, reference default flags), "
                                      if res.get("schedule") else "block=16, ")
                                   + f"temp={T:g}, batch={R} per GPU, prefix={args.prefix}, random-init weights, "
                                   "scripted acceptance (seeded truncated-geometric, mean tau 7.3 over the timed cycles"
                                   + (", capped by the cycle's block size" if res.get("schedule") else "") + ")",
                       "target_layers": meta["layers"], "requests": world * R,
                       "target_verify": "hf" if args.hf_verify else "native", "parallelism": f"dp{world}"},
            "mean_acceptance_length": mean_tau, "raw_tau1_value": cyc_sum / dt_max,
            "lossless_fraction": res["lossless_fraction"], "roofline": res["roofline"], "hot_path": res["hot_path"],
            "ttft_side": res.get("ttft_side"), "host_side": res.get("host_side"), "batch4": b4_obj, "cpu_baseline": cpu,
        }
        if res.get("used_block_sizes"):
            line["used_block_sizes"] = res["used_block_sizes"]
        if R > 1:
            line["mode"] = res["mode"]
        if rccl_note:
            line["collective"] = rccl_note
        if rehearsal:
            line["rehearsal"] = "DFL_BENCH_SHARE_GPU=1: all ranks shared cuda:0 over gloo — control flow only, not a measurement"
        print(json.dumps(line), flush=True)
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
