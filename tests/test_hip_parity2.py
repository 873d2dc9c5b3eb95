"""Round-2 parity cases (VERDICT r1 "weak #1"): the native target verify at BASELINE.json's
full shapes against the HF forward, the T = 0.7 path with a NON-degenerate posterior, the
MoE-target config on the HF-verify path, the 30B-A3B draft geometry against the oracle,
and tapped-layer lists with repeated ids."""
import math

import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def dev():
    return torch.device("cuda", 0)


def _hf_on_gpu(cls, cfg, seed, dtype=BF16):
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(dev()):
            m = cls(cfg)
    finally:
        torch.set_default_dtype(prev)
    return m.eval()


def _verify_vs_hf(name, hf, V, H_, taps, P=70, bs=16, seed=2):
    """One block through NativeTarget.verify and through the wrapped HF model: logits of all bs
    rows, tapped hidden rows, appended K/V rows, posterior ids on margin-screened rows."""
    from transformers import DynamicCache
    from dflash_amd import NativeTarget
    nt = NativeTarget(hf)
    g = torch.Generator().manual_seed(seed)
    prompt = torch.randint(0, V - 100, (1, P), generator=g).to(dev())
    block = torch.randint(0, V - 100, (1, bs), generator=g).to(dev())
    cache = nt.new_cache(P + 64)
    nt.prefill(prompt, cache)
    logits = torch.zeros(16, V, dtype=BF16, device=dev())
    post, th = nt.verify(block[0], P, cache, tap_layers=taps, logits_out=logits)
    rc = DynamicCache()
    with torch.inference_mode():
        hf(prompt, past_key_values=rc, use_cache=True)
        ref = hf(block, position_ids=torch.arange(P, P + bs, device=dev())[None], past_key_values=rc, use_cache=True,
                 output_hidden_states=True)
    rl = ref.logits[0].float()
    H.assert_close(f"{name} verify logits", logits[:bs], rl)
    assert torch.equal(post[0], torch.argmax(logits[:bs], dim=-1))
    # random-init weights: a V-way argmax has tiny top-2 margins; the screened rows must agree
    H.assert_ids_match_where_safe(f"{name} verify ids", post[0], rl, margin_rel=2e-2, min_safe=1)
    for j, l in enumerate(taps):
        H.assert_close(f"{name} verify tap layer {l}", th[:bs, j * H_:(j + 1) * H_], ref.hidden_states[l + 1][0])
    L = hf.config.num_hidden_layers
    for li in (0, L - 1):
        H.assert_close(f"{name} verify K layer {li}", cache.k[li][:, :P + bs], rc.layers[li].keys[0],
                       max_rel=H.KV_MAX_REL)
        H.assert_close(f"{name} verify V layer {li}", cache.v[li][:, :P + bs], rc.layers[li].values[0],
                       max_rel=H.KV_MAX_REL)
    return nt


def test_native_verify_full_size_qwen3_8b():
    """BASELINE configs[1] target shapes (H 4096, FFN 12288, V 151936, 32/8 heads) at 3 layers:
    the full-size twin of test_native_verify_matches_hf_forward — every GEMM shape of the bench's
    verify (K-chunked down_proj, 151936-row lm_head) against the HF forward."""
    from dflash_amd.config import QWEN3_8B_TARGET
    from dflash_amd.synthetic import make_hf_qwen3
    torch.manual_seed(21)
    hf = make_hf_qwen3({**QWEN3_8B_TARGET, "num_layers": 3}, dev())
    _verify_vs_hf("Qwen3-8B shapes", hf, 151936, 4096, taps=[0, 1])


def test_native_verify_full_size_llama31_8b():
    """BASELINE configs[3] target shapes (Llama-3.1-8B: FFN 14336 -> K = 14336 chunked down_proj,
    V 128256, no q/k norm, "llama3" RoPE scaling with the model card's parameters) at 2 layers."""
    tf = pytest.importorskip("transformers")
    cfg = tf.LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=2,
                         num_attention_heads=32, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-5,
                         max_position_embeddings=131072, tie_word_embeddings=False, attention_bias=False,
                         mlp_bias=False,
                         rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0,
                                          "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                          "original_max_position_embeddings": 8192})
    hf = _hf_on_gpu(tf.LlamaForCausalLM, cfg, seed=23)
    nt = _verify_vs_hf("Llama-3.1-8B shapes", hf, 128256, 4096, taps=[0], bs=13)
    assert nt.layers[0]["q_norm"] is None and nt.rope_type == "llama3"


# ------------------------------------------------------------------ T = 0.7, non-degenerate
def _peaky_tiny_hf(layers=3, gain=3.0):
    """tiny HF target whose posterior has a handful of comparable modes per row: the lm_head of a
    random-init model scaled up so that logits / 0.7 have a spread of a few units."""
    from dflash_amd.synthetic import make_hf_qwen3
    torch.manual_seed(31)
    hf = make_hf_qwen3({**H.TINY_TARGET, "num_layers": layers}, dev())
    with torch.no_grad():
        hf.lm_head.weight.mul_(gain / (0.02 * math.sqrt(512)))
    return hf


def _chi2(counts: torch.Tensor, p: torch.Tensor, top: int = 8):
    """Pearson statistic of observed draws against probabilities p over `top` individual tokens
    plus one bin for the rest; returns (chi2, dof)."""
    n = int(counts.sum())
    idx = p.topk(top).indices
    obs = torch.cat([counts[idx].double(), (n - counts[idx].sum()).double()[None]])
    exp = torch.cat([p[idx].double(), (1.0 - p[idx].sum()).double().clamp_min(1e-12)[None]]) * n
    keep = exp > 5.0
    return float(((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()), int(keep.sum()) - 1


def test_temperature_sampling_is_distributed_as_softmax():
    """model/utils.py:30-34 on the product's device path with a posterior that is NOT an argmax in
    disguise: NativeTarget.verify(temperature = 0.7) materialises the bf16 logits with the lm_head
    GEMM and draws with softmax + multinomial.  (a) the materialised logits equal an fp32 matmul of
    the final hidden rows; (b) >= 20k draws per test follow softmax(logits / T) (chi-square over the
    top-8 tokens + rest of every row, 5-sigma bound); (c) on those very draws the accept kernel
    implements "block[i+1] == sampled[i]" (model/dflash.py:257-258).  Bit parity with the
    reference's CPU RNG stream is not achievable: parity unpinned, distribution checked."""
    from dflash_amd import NativeTarget, ops
    hf = _peaky_tiny_hf()
    nt = NativeTarget(hf)
    V, T, bs, P = 2048, 0.7, 16, 40
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, 2000, (1, P), generator=g).to(dev())
    block = torch.randint(0, 2000, (1, bs), generator=g).to(dev())
    cache = nt.new_cache(P + 64)
    nt.prefill(prompt, cache)
    logits = torch.zeros(16, V, dtype=BF16, device=dev())
    nt.verify(block[0], P, cache, logits_out=logits)
    # (a) all bs rows of the materialised logits against the HF forward's (fp32 matmul of its own rows)
    from transformers import DynamicCache
    rc = DynamicCache()
    with torch.inference_mode():
        hf(prompt, past_key_values=rc, use_cache=True)
        ref = hf(block, position_ids=torch.arange(P, P + bs, device=dev())[None], past_key_values=rc, use_cache=True,
                 output_hidden_states=True)
        hid = ref.hidden_states[-1][0]          # final-normed rows
        rl = hid.float() @ hf.lm_head.weight.float().T
    H.assert_close("T>0 materialised logits vs fp32 matmul", logits, rl)
    p_ref = torch.softmax(logits.float() / T, dim=-1).cpu()     # expectation from the product's own bf16 logits
    # the posterior must be non-degenerate: the top token of a typical row holds well under all the mass
    top1 = p_ref.max(dim=-1).values
    assert float(top1.median()) < 0.9 and float(top1.min()) < 0.6, top1
    # (b) draws through verify(): 16 rows per call
    torch.manual_seed(0)
    n_calls = 1400
    draws = torch.empty(n_calls, bs, dtype=torch.long)
    for c in range(n_calls):
        cache.crop(P)
        post, _ = nt.verify(block[0], P, cache, temperature=T)
        draws[c] = post[0].cpu()
    assert draws.numel() >= 20000
    chi, dof = 0.0, 0
    for r in range(bs):
        c2, d = _chi2(torch.bincount(draws[:, r], minlength=V), p_ref[r])
        chi, dof = chi + c2, dof + d
    bound = dof + 5.0 * math.sqrt(2.0 * dof)
    print(f"[parity] T=0.7 verify draws: chi2 {chi:.1f} over {dof} dof (5-sigma bound {bound:.1f}), "
          f"median top-1 mass {float(top1.median()):.3f}")
    H._log_parity({"what": "T=0.7 verify chi2", "chi2": chi, "dof": dof, "bound": bound})
    assert chi <= bound
    assert len({tuple(x.tolist()) for x in draws[:50]}) > 25        # really stochastic
    # (c) accept rule on real draws: k agreeing draft tokens, then a mismatch
    out = torch.full((P + 64,), 2047, dtype=torch.long, device=dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    res = torch.zeros(4, dtype=torch.int32, device=dev())
    for c in range(0, 48):
        post = draws[c].to(dev())
        k = c % bs                                   # 0..15 agreeing tokens (15 = all accepted)
        blk = torch.empty(bs, dtype=torch.long, device=dev())
        blk[0] = block[0, 0]
        blk[1:k + 1] = post[:k]
        if k + 1 < bs:
            blk[k + 1:] = (post[k:bs - 1] + 1) % 2000   # every later slot disagrees with the sample
        ops.set_dyn(dyn, 0, 0, bs, P)
        ops.accept_commit(blk, post, bs, out, dyn, None, res)
        acc, new_start = res.tolist()[:2]
        assert acc == k and new_start == P + k + 1
        assert out[P:P + k + 1].tolist() == blk[:k + 1].tolist() and int(out[P + k + 1]) == int(post[k])


def test_batched_temperature_sampling_is_distributed_as_softmax():
    """The batched verify's T > 0 branch (batch.py: materialised [MT,16,V] logits -> softmax +
    multinomial over all requests' rows): two requests, frequencies against softmax(logits / T)."""
    from dflash_amd import NativeTarget
    from dflash_amd.batch import BatchedDecoder
    cfg = H.tiny_cfg()
    from dflash_amd import DFlashDraftModel
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, dtype=BF16))
    hf = _peaky_tiny_hf(layers=6)
    nt = NativeTarget(hf)
    T, V = 0.7, 2048
    dec = BatchedDecoder(m, nt, 2, max_rows=160, out_len=140, mask_token_id=cfg.mask_token_id, temperature=T)
    g = torch.Generator().manual_seed(9)
    for r, P in enumerate((33, 21)):
        dec.admit(r, torch.randint(0, 2000, (1, P), generator=g).to(dev()), temperature=0.0)
    dec.draft()
    dec.temperature = 0.0
    dec.verify()                                     # T = 0 pass leaves nothing materialised ...
    dec.temperature = T
    torch.manual_seed(1)
    n_calls = 700
    draws = torch.empty(n_calls, 2, 16, dtype=torch.long)
    for c in range(n_calls):
        dec.verify()                                 # state does not advance without accept()
        draws[c] = dec.post[:2].cpu()
    p_ref = torch.softmax(dec._logits[:2].float() / T, dim=-1).cpu()
    assert draws.numel() >= 20000
    chi, dof = 0.0, 0
    for r in range(2):
        for j in range(16):
            c2, d = _chi2(torch.bincount(draws[:, r, j], minlength=V), p_ref[r, j])
            chi, dof = chi + c2, dof + d
    bound = dof + 5.0 * math.sqrt(2.0 * dof)
    print(f"[parity] T=0.7 batched verify draws: chi2 {chi:.1f} over {dof} dof (bound {bound:.1f})")
    H._log_parity({"what": "T=0.7 batched verify chi2", "chi2": chi, "dof": dof, "bound": bound})
    assert chi <= bound
    # the two requests' rows are different distributions: request 0's draws must NOT fit request 1's
    c_wrong, d_wrong = 0.0, 0
    for j in range(16):
        c2, d = _chi2(torch.bincount(draws[:, 0, j], minlength=V), p_ref[1, j])
        c_wrong, d_wrong = c_wrong + c2, d_wrong + d
    assert c_wrong > 10 * (d_wrong + 5.0 * math.sqrt(2.0 * d_wrong))


def test_policy_loop_with_stochastic_target_keeps_the_accept_invariant():
    """dflash_generate_policy at T = 0.7 on the non-degenerate target (BASELINE configs[3] path,
    benchmark_dynamic_schedule.py:342 samples the draft with T too): the run must be a valid
    sampled continuation — every cycle commits acc + 1 tokens, acc in [0, bs-1], output length and
    acceptance bookkeeping consistent — and two seeds give different text."""
    from dflash_amd import DFlashDraftModel, EWMAPerformanceScheduler, NativeTarget, dflash_generate_policy
    cfg = H.tiny_cfg()
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, dtype=BF16))
    nt = NativeTarget(_peaky_tiny_hf(layers=6))
    prompt = torch.randint(0, 2000, (1, 30), generator=torch.Generator().manual_seed(3)).to(dev())
    outs = []
    for seed in (0, 1):
        sched = EWMAPerformanceScheduler(candidates=[8, 12, 16], scheduler_mode="ewma", warmup_cycles=3,
                                         ewma_alpha=0.25, switch_margin=0.03, required_streak=2, cooldown_cycles=2,
                                         probe_interval=5, low_accept_threshold=0.2, low_accept_streak=3, adl_rho=0.3,
                                         adl_delta=1.0, adl_k_min=8, adl_k_max=16, adl_neighborhood=4)
        torch.manual_seed(seed)
        r = dflash_generate_policy(model=m, target=nt, input_ids=prompt, mask_token_id=cfg.mask_token_id,
                                   max_new_tokens=48, stop_token_ids=None, temperature=0.7, scheduler=sched)
        # (a sampled id can be the mask id 2047 itself, which the reference's trim then drops: :270)
        assert 44 <= r.num_output_tokens <= 48 and r.output_ids.shape[1] == 30 + r.num_output_tokens
        assert all(1 <= t <= b for t, b in zip(r.acceptance_lengths, r.used_block_sizes))
        assert sum(r.acceptance_lengths) >= 48 and r.output_ids[0, :30].tolist() == prompt[0].tolist()
        outs.append(r.output_ids[0].tolist())
    assert outs[0] != outs[1]


# ------------------------------------------------------------------ configs[4]: MoE target, 30B-A3B draft geometry
def test_moe_target_through_policy_loop_is_lossless():
    """BASELINE configs[4] (Qwen3-Coder-30B-A3B + DFlash, dynamic schedule): a tiny HF
    Qwen3MoeForCausalLM target (fp32, so the greedy argmax cannot flip between a 1-token and a
    16-token forward) through dflash_generate_policy on the HF-verify path (the native MoE verify has its own
    tests, test_hip_moe.py).  The committed ids are the MoE target's own greedy continuation."""
    tf = pytest.importorskip("transformers")
    from transformers import DynamicCache
    from dflash_amd import DFlashDraftModel, EWMAPerformanceScheduler, NativeTarget, dflash_generate_policy
    cfg_t = tf.Qwen3MoeConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, moe_intermediate_size=256,
                              num_hidden_layers=6, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                              num_experts=8, num_experts_per_tok=2, decoder_sparse_step=1, norm_topk_prob=True,
                              max_position_embeddings=4096, rms_norm_eps=1e-6, tie_word_embeddings=False,
                              rope_parameters={"rope_type": "default", "rope_theta": 1e6}, mlp_only_layers=[])
    moe = _hf_on_gpu(tf.Qwen3MoeForCausalLM, cfg_t, seed=41, dtype=torch.float32)
    cfg = H.tiny_cfg()
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, dtype=BF16))
    prompt = torch.randint(0, 2000, (1, 26), generator=torch.Generator().manual_seed(6)).to(dev())
    n_new = 40
    sched = EWMAPerformanceScheduler(candidates=[8, 12, 16], scheduler_mode="ewma", warmup_cycles=3, ewma_alpha=0.25,
                                     switch_margin=0.03, required_streak=2, cooldown_cycles=2, probe_interval=5,
                                     low_accept_threshold=0.2, low_accept_streak=3, adl_rho=0.3, adl_delta=1.0,
                                     adl_k_min=8, adl_k_max=16, adl_neighborhood=4)
    r = dflash_generate_policy(model=m, target=moe, input_ids=prompt, mask_token_id=cfg.mask_token_id,
                               max_new_tokens=n_new, stop_token_ids=None, temperature=0.0, scheduler=sched)
    # the MoE target's own greedy continuation, token by token
    with torch.inference_mode():
        c = DynamicCache()
        ar = prompt.clone()
        out = moe(ar, past_key_values=c, use_cache=True, logits_to_keep=1)
        for _ in range(n_new):
            nxt = out.logits[:, -1:].argmax(-1)
            ar = torch.cat([ar, nxt], dim=1)
            out = moe(nxt, past_key_values=c, use_cache=True)
    assert r.output_ids[0].tolist() == ar[0].tolist()
    assert set(r.used_block_sizes) <= set(range(1, 17)) and len(r.cycle_trace) == len(r.acceptance_lengths)
    # with scripted agreement (tau > 1) the loop is still lossless on the MoE target
    G = ar[0]

    def hook(blk, start, call):
        k = min(call % 7, blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]

    sched2 = EWMAPerformanceScheduler(candidates=[8, 12, 16], scheduler_mode="ewma", warmup_cycles=3, ewma_alpha=0.25,
                                      switch_margin=0.03, required_streak=2, cooldown_cycles=2, probe_interval=5,
                                      low_accept_threshold=0.2, low_accept_streak=3, adl_rho=0.3, adl_delta=1.0,
                                      adl_k_min=8, adl_k_max=16, adl_neighborhood=4)
    r2 = dflash_generate_policy(model=m, target=moe, input_ids=prompt, mask_token_id=cfg.mask_token_id,
                                max_new_tokens=n_new - 8, stop_token_ids=None, temperature=0.0, scheduler=sched2,
                                draft_token_hook=hook)
    assert r2.output_ids[0].tolist() == ar[0, :26 + n_new - 8].tolist()
    assert max(r2.acceptance_lengths) > 1


def test_draft_at_qwen3_30b_a3b_geometry_matches_oracle():
    """The draft geometry BASELINE configs[4] implies (target Qwen3-Coder-30B-A3B: H 2048, 32 query /
    4 kv heads -> GQA group 8, q_dim 4096 = 2 H; the draft shares the widths): three cycles of the
    draft forward with cache against the CPU oracle on the same seeded weights."""
    from oracle import dflash_oracle as O
    from dflash_amd import DFlashDraftModel
    from dflash_amd.config import DFlashConfig
    cfg = DFlashConfig(hidden_size=2048, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=4,
                       head_dim=128, intermediate_size=6144, vocab_size=4096, num_target_layers=48, block_size=16,
                       rope_theta=1e7, mask_token_id=4095)
    assert cfg.q_dim == 2 * cfg.hidden_size and cfg.num_attention_heads // cfg.num_key_value_heads == 8
    w = H.draft_weights(cfg, seed=27, dtype=BF16)
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(w)
    oc = H.oracle_cfg(cfg, "sdpa")
    g = torch.Generator().manual_seed(19)
    ocache = O.ListKVCache()
    cache = m.new_cache(256)
    start = 45
    for c, (ctx, bs, tau_next) in enumerate(((45, 16, 7), (7, 16, 16), (16, 8, 2))):
        th = (torch.randn(1, ctx, cfg.fc_in, generator=g) * 1.5).to(BF16)
        ne = (torch.randn(1, bs, cfg.hidden_size, generator=g) * 0.05).to(BF16)
        pos = torch.arange(ocache.get_seq_length(), start + bs)[None]
        ref = O.draft_forward(w, oc, position_ids=pos, noise_embedding=ne, target_hidden=th, cache=ocache)
        ocache.crop(start)
        got = m(target_hidden=th.to(dev()), noise_embedding=ne.to(dev()), position_ids=pos.to(dev()),
                past_key_values=cache, use_cache=True, is_causal=False)
        cache.crop(start)
        H.assert_close(f"30B-A3B-geometry draft hidden cycle {c}", got, ref)
        start += tau_next
    n = cache.get_seq_length()
    for li in range(cfg.num_hidden_layers):
        H.assert_close(f"30B-A3B-geometry draft K layer {li}", cache.k[li][:, :n], ocache.k[li][0], max_rel=H.KV_MAX_REL)
        H.assert_close(f"30B-A3B-geometry draft V layer {li}", cache.v[li][:, :n], ocache.v[li][0], max_rel=H.KV_MAX_REL)


# ------------------------------------------------------------------ repeated tap ids
def test_repeated_target_layer_ids_fill_every_tap_slot():
    """build_target_layer_ids(6, 5) = [1, 2, 2, 2, 3]: the reference concatenates layer 2's state
    three times (model/utils.py:16-25).  Every slot of the native verify's tap rows must hold it —
    single-request and batched."""
    from dflash_amd import NativeTarget, build_target_layer_ids, extract_context_feature
    from dflash_amd.synthetic import make_hf_qwen3
    from transformers import DynamicCache
    ids = build_target_layer_ids(6, 5)
    assert ids == [1, 2, 2, 2, 3]
    torch.manual_seed(11)
    hf = make_hf_qwen3({**H.TINY_TARGET, "num_layers": 6}, dev())
    nt = NativeTarget(hf)
    g = torch.Generator().manual_seed(2)
    prompt = torch.randint(0, 2000, (1, 37), generator=g).to(dev())
    block = torch.randint(0, 2000, (1, 16), generator=g).to(dev())
    cache = nt.new_cache(128)
    nt.prefill(prompt, cache)
    _, th = nt.verify(block[0], 37, cache, tap_layers=ids)
    rc = DynamicCache()
    with torch.inference_mode():
        hf(prompt, past_key_values=rc, use_cache=True)
        ref = hf(block, position_ids=torch.arange(37, 53, device=dev())[None], past_key_values=rc, use_cache=True,
                 output_hidden_states=True)
    want = extract_context_feature(ref.hidden_states, ids)[0]
    H.assert_close("repeated taps, single request", th[:16], want)
    assert torch.equal(th[:, 512:1024], th[:, 1024:1536]) and torch.equal(th[:, 512:1024], th[:, 1536:2048])
    # batched verify: same rows in every slot
    from dflash_amd import DFlashDraftModel
    from dflash_amd.batch import BatchedDecoder
    cfg = H.tiny_cfg(num_hidden_layers=5, target_layer_ids=ids)
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, dtype=BF16))
    dec = BatchedDecoder(m, nt, 2, max_rows=160, out_len=120, mask_token_id=cfg.mask_token_id)
    dec.admit(0, prompt)
    dec.admit(1, prompt[:, :20])
    dec.block[0] = block[0]
    dec.verify()
    H.assert_close("repeated taps, batched request 0", dec.d["taps"][0], want)
