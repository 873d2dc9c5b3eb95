"""Multi-candidate verify on the GPU: top-k + log-sum-exp kernel, candidate attention launch,
selection + commit kernel, the native one-pass verify of up to 4 (two passes: 8) candidates, and the loop."""
import json
import os

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
G9 = json.load(open(os.path.join(H.GOLDEN, "candidates.json")))


def dev():
    return torch.device("cuda", 0)


def _tiny_hf(dtype=BF16, layers=6):
    from dflash_amd.synthetic import make_hf_qwen3
    torch.manual_seed(11)
    return make_hf_qwen3({**H.TINY_TARGET, "num_layers": layers}, dev(), dtype=dtype)


def make_model(cfg, seed=3):
    from dflash_amd import DFlashDraftModel
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, seed=seed, dtype=BF16))
    return m


@pytest.mark.parametrize("V,rows,k", [(2048, 15, 8), (151936, 15, 4), (4096 + 16 * 7, 7, 2), (2048, 1, 1)])
def test_topk_rows(V, rows, k):
    """dfl_topk_rows vs torch.topk / logsumexp on tie-free rows (golden G9's logits at V = 2048, distinct random
    bf16 values otherwise), and its documented order on engineered ties."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(V + rows)
    if V == 2048 and rows == 15:
        x = torch.tensor(np.array(G9["cases"][0]["logits_bits"], dtype=np.int16)).view(BF16)
    else:
        # distinct values per row: a random permutation of V equally spaced fp32 numbers, those that survive bf16
        # rounding without colliding among the top few are what torch.topk and the kernel must agree on
        x = (torch.rand(rows, V, generator=g) * 8 - 4).to(BF16)
    xd = x.to(dev())
    val, idx, lse = ops.topk_rows(xd, k)
    rv, ri = torch.topk(x.float(), k=k, dim=-1)
    assert torch.equal(val[:, :k].cpu(), rv)                                  # values: exact, ties or not
    for r in range(rows):                                                     # indices: exact where the value is unique
        for j in range(k):
            if (x[r].float() == rv[r, j]).sum() == 1:
                assert int(idx[r, j]) == int(ri[r, j]), (r, j)
            else:                                                             # tie: lowest indices first
                same = (x[r].float() == rv[r, j]).nonzero()[:, 0].tolist()
                assert int(idx[r, j]) in same
    assert torch.allclose(lse.cpu(), torch.logsumexp(x.float(), dim=-1), rtol=1e-5, atol=1e-5)
    # engineered ties: equal maxima -> ascending index
    t = torch.zeros(2, 64, dtype=BF16)
    t[0, [5, 9, 40]] = 3.0
    t[1, [63, 0]] = 1.0
    _, ti, _ = ops.topk_rows(t.to(dev()), 3)
    assert ti[0, :3].tolist() == [5, 9, 40] and ti[1, :2].tolist() == [0, 63]


def test_candidate_select_matches_the_rule():
    """dfl_candidate_select vs the oracle's restatement of :586-613 on random and engineered candidate sets."""
    from dflash_amd import ops
    from oracle import candidates_oracle as CO
    g = torch.Generator().manual_seed(3)
    for trial in range(40):
        C, bs = int(torch.randint(1, 9, (1,), generator=g)), int(torch.randint(1, 17, (1,), generator=g))
        post = torch.randint(0, 50, (C, bs), generator=g)
        blocks = torch.randint(0, 50, (C, bs), generator=g)
        for c in range(C):   # agree on a random prefix
            k = int(torch.randint(0, bs, (1,), generator=g))
            blocks[c, 1:k + 1] = post[c, :k]
        scores = (torch.randn(C, generator=g) * (0 if trial % 3 == 0 else 5)).float()
        meta = [{"draft_score": float(s)} for s in scores]
        win, acc, taus = CO.choose_candidate(blocks, post, meta)
        start = 7
        out = torch.full((64,), 99, dtype=torch.long, device=dev())
        dyn = torch.zeros(8, dtype=torch.int32, device=dev())
        res = torch.zeros(12, dtype=torch.int32, device=dev())
        ops.set_dyn(dyn, 0, 0, bs, start)
        ops.candidate_select(blocks.to(dev()), post.to(dev()), scores.to(dev()), bs, out, dyn, None, res)
        r = res.tolist()
        assert (r[0], r[1], r[3]) == (acc, start + acc + 1, win), trial
        assert [a + 1 for a in r[4:4 + C]] == taus and all(a == -1 for a in r[4 + C:])
        want = blocks[win, :acc + 1].tolist() + [int(post[win, acc])]
        assert out[start:start + acc + 2].tolist() == want and int(out[start + acc + 2]) == 99
    # stop flag: a stop id among the committed tokens
    blocks = torch.tensor([[4, 5, 6, 7]], device=dev())
    post = torch.tensor([[5, 6, 1, 1]], device=dev())
    out = torch.zeros(32, dtype=torch.long, device=dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    res = torch.zeros(12, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 0, 4, 3)
    ops.candidate_select(blocks, post, torch.zeros(1, device=dev()), 4, out, dyn, torch.tensor([1], device=dev()), res)
    assert res.tolist()[:4] == [2, 6, 1, 0]


def test_native_candidate_pass_equals_single_verifies():
    """One pass of the native verifier over 3 (and two passes over 6) candidate blocks vs NativeTarget.verify of each
    block alone on the same prefix: posterior ids on margin-screened rows, tapped rows and staged K/V within the bf16
    tolerance; after keep(), the cache holds the winner's rows and decoding continues from it."""
    from dflash_amd import NativeTarget
    from dflash_amd.candidates import NativeCandidateVerifier
    hf = _tiny_hf()
    nt = NativeTarget(hf)
    g = torch.Generator().manual_seed(6)
    P, bs = 45, 16
    prompt = torch.randint(0, 2000, (1, P), generator=g).to(dev())
    cands = torch.randint(0, 2000, (3, bs), generator=g).to(dev())
    cands[:, 0] = cands[0, 0]
    cache = nt.new_cache(160)
    nt.prefill(prompt, cache)
    taps = [1, 3]
    ver = NativeCandidateVerifier(nt, len(taps))
    post = ver.verify(cands, P, cache, taps).clone()
    assert cache.get_seq_length() == P                     # the pass leaves the cache alone
    for c in range(3):
        c2 = nt.new_cache(160)
        nt.prefill(prompt, c2)
        logits = torch.zeros(32, 2048, dtype=BF16, device=dev())
        p1, th = nt.verify(cands[c], P, c2, tap_layers=taps, logits_out=logits)
        H.assert_ids_match_where_safe(f"candidate {c} posterior", post[c], logits[:bs].float())
        H.assert_close(f"candidate {c} taps", ver.taps[c], th[:16])
        for li in (0, 5):
            H.assert_close(f"candidate {c} staged K l{li}", ver.stage_k[li, c, :, :bs], c2.k[li][:, P:P + bs], max_rel=H.KV_MAX_REL)
            H.assert_close(f"candidate {c} staged V l{li}", ver.stage_v[li, c, :, :bs], c2.v[li][:, P:P + bs], max_rel=H.KV_MAX_REL)
    ver.keep(2, P, bs, cache)
    assert cache.get_seq_length() == P + bs
    assert torch.equal(cache.k[3][:, P:P + bs], ver.stage_k[3, 2, :, :bs]) and torch.equal(cache.v[0][:, P:P + bs], ver.stage_v[0, 2, :, :bs])


@pytest.mark.parametrize("mode,kw", [("branch_beam", dict(branch_top_k=2, max_candidates=4)),
                                     ("fixed_prefix_rank", dict(branch_top_k=4, max_candidates=4)),
                                     ("uncertainty_sparse_rank", dict(branch_top_k=3, max_candidates=6)),
                                     ("branch_beam", dict(branch_top_k=3, max_candidates=8, adaptive_candidates=True))])
def test_candidate_loop_is_lossless_on_native_target(mode, kw):
    """dflash_generate_candidate_solutions on the native target (large-margin greedy walk): the committed ids are the
    target's greedy continuation whatever candidates are tried; the chosen candidate has the maximal tau of its cycle;
    cycle_trace / candidate_summary carry the reference's fields."""
    from dflash_amd import NativeTarget, dflash_generate_candidate_solutions
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk
    cfg = H.tiny_cfg()
    m = make_model(cfg)
    hf = _tiny_hf()
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf)
    prompt = torch.randint(0, 2000, (1, 33), generator=torch.Generator().manual_seed(4)).to(dev())
    n_new = 70
    G = greedy_walk(perm, prompt, n_new + 40).to(dev())
    plan = H.make_plan(64, 16, 17)

    def hook(blk, start, call):       # scripted base block: k agreeing tokens, then a wrong one
        k = min(plan[call], blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            w = G[start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])

    r = dflash_generate_candidate_solutions(m, nt, prompt, cfg.mask_token_id, n_new, 16, None, candidate_mode=mode,
                                            collect_profile=True, draft_token_hook=hook, **kw)
    assert r.output_ids[0].tolist() == G[:33 + n_new].tolist()
    assert sum(r.acceptance_lengths) >= n_new and len(r.cycle_trace) == len(r.acceptance_lengths)
    for row in r.cycle_trace:
        assert row["tau"] == max(row["candidate_taus"]) and row["num_candidates"] == len(row["candidate_taus"])
        # (branch_beam appends one beam before it tests the cap, :166-176: a budget of 1 can yield 2 candidates)
        assert row["cycle_max_candidates"] <= kw["max_candidates"]
        assert row["num_candidates"] <= max(row["cycle_max_candidates"], 2 if mode == "branch_beam" else 1)
        assert {"draft_s", "target_s", "cycle_s", "selected_positions", "chosen_candidate_idx",
                "candidate_draft_scores", "candidate_rank_variants"} <= set(row)
        # candidate 0 is the (scripted) base block: a winner with a longer prefix must be another candidate
        w = row["chosen_candidate_idx"]
        assert (w == 0 or row["tau"] > row["candidate_taus"][0]
                or row["candidate_draft_scores"][w] >= row["candidate_draft_scores"][0])
    assert max(row["num_candidates"] for row in r.cycle_trace) > 1
    cs = r.candidate_summary
    assert cs["candidate_mode"] == mode and cs["candidate_count_sum"] == sum(x["num_candidates"] for x in r.cycle_trace)
    assert cs["candidate_verify_calls"] == len(r.cycle_trace)
    assert r.profile_summary["profiled_cycles"] == len(r.cycle_trace)


def test_candidate_that_is_right_beats_a_wrong_base_block():
    """Engineered win: the base block breaks at slot 3, candidate 2 carries the target's true continuation.  The pass
    must choose it (tau 16), commit its tokens and leave ITS K/V rows in the cache, so that the next verify of the
    true continuation from there agrees with a cache built by plain verifies."""
    from dflash_amd import NativeTarget, ops
    from dflash_amd.candidates import NativeCandidateVerifier
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk
    hf = _tiny_hf()
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf)
    P, bs = 40, 16
    prompt = torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(9)).to(dev())
    G = greedy_walk(perm, prompt, 80).to(dev())
    cache = nt.new_cache(160)
    out0 = nt.prefill(prompt, cache)
    first = int(out0.logits[0, -1].argmax())
    assert first == int(G[P])
    cands = torch.stack([G[P:P + bs].clone() for _ in range(3)])
    cands[0, 3:] = (cands[0, 3:] + 1) % 2000            # base: 2 agreeing tokens, then wrong
    cands[1, 9:] = (cands[1, 9:] + 7) % 2000            # better, still wrong at slot 9
    ver = NativeCandidateVerifier(nt, 2)
    post = ver.verify(cands, P, cache, [1, 3])
    out = torch.full((P + 64,), 2047, dtype=torch.long, device=dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    res = torch.zeros(12, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 0, bs, P)
    ops.candidate_select(cands, post, torch.tensor([0.0, -1.0, -2.0], device=dev()), bs, out, dyn, None, res)
    r = res.tolist()
    assert r[3] == 2 and r[0] == 15 and r[4:7] == [2, 8, 15]
    assert out[P:P + 17].tolist() == G[P:P + 17].tolist()
    ver.keep(2, P, bs, cache)
    cache.crop(P + 16)
    post2, _ = nt.verify(G[P + 16:P + 32], P + 16, cache)
    assert post2[0].tolist() == G[P + 17:P + 33].tolist()


def test_candidate_loop_on_hf_target_path():
    """A target the native verify does not cover (HF Qwen3 in fp32 here; an MoE model in BASELINE configs[4]) goes
    through its own forward on a batch-expanded copy of its DynamicCache, as the reference does (:570-585, :604-608):
    lossless against the target's greedy continuation."""
    from transformers import DynamicCache
    from dflash_amd import dflash_generate_candidate_solutions
    cfg = H.tiny_cfg()
    m = make_model(cfg)
    hf = _tiny_hf(dtype=torch.float32)
    prompt = torch.randint(0, 2000, (1, 21), generator=torch.Generator().manual_seed(2)).to(dev())
    n_new = 24
    r = dflash_generate_candidate_solutions(m, hf, prompt, cfg.mask_token_id, n_new, 16, None,
                                            candidate_mode="fixed_prefix_rank", branch_top_k=3, max_candidates=3,
                                            fixed_prefix_len=2)
    with torch.inference_mode():
        c = DynamicCache()
        ar = prompt.clone()
        o = hf(ar, past_key_values=c, use_cache=True, logits_to_keep=1)
        for _ in range(n_new):
            nxt = o.logits[:, -1:].argmax(-1)
            ar = torch.cat([ar, nxt], dim=1)
            o = hf(nxt, past_key_values=c, use_cache=True)
    assert r.output_ids[0].tolist() == ar[0].tolist()
    assert all(x["num_candidates"] == 3 for x in r.cycle_trace if x["effective_block_size"] > 2)


def test_topk_rows_with_masked_vocabulary_entries():
    """-inf logits (a masked vocabulary range, also as the FIRST entries a thread sees) add nothing to the log-sum-exp —
    torch.log_softmax handles them; exp(-inf - -inf) must not poison the row."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(5, 4096, generator=g) * 3).to(BF16)
    x[0, :1024] = float("-inf")           # every thread's first chunk
    x[1, ::2] = float("-inf")
    x[2, 100:4000] = float("-inf")
    x[3, -1] = float("-inf")
    val, idx, lse = ops.topk_rows(x.to(dev()), 4)
    ref = torch.logsumexp(x.float(), dim=-1)
    assert torch.isfinite(lse).all()
    assert torch.allclose(lse.cpu(), ref, rtol=1e-5, atol=1e-4)
    tv, ti = torch.topk(x.float(), 4, dim=-1)
    assert torch.equal(val[:, :4].cpu(), tv)


def _moe_target():
    import test_hip_moe as TM
    return TM._moe_hf(mlp_only=(1,))


@pytest.mark.parametrize("C", [1, 2, 3])
@pytest.mark.parametrize("shared_pass", [False, True])
def test_native_candidate_pass_on_moe_target_equals_single_verifies(shared_pass, C):
    """The one-pass candidate verify on a sparse-MoE target (round 3: VERDICT r2 "missing" #3 — it fell back to the HF
    forward on a batch-expanded cache): C candidate blocks in one pass vs NativeTarget.verify of each block alone on the
    same prefix.  C = 1 and C = 2: the verifier always holds four tile slots while the norm launch
    behind the experts strides the expert shares by batch_tiles(C) = 2 tiles — the per-tile branch of moe_mlp_tiles
    must lay the shares out with THAT stride (C = 1 always takes it; C = 2 with the shared pass off).  shared_pass False: every candidate routes its own rows through the per-tile kernels — the SAME router
    kernel on the same fragments as the single verify, so no near-tie can flip.  True (the default from three tiles on):
    the 48 rows go through ONE pass over the experts (the prefill's grouped kernels); the router logits then come from the
    batch GEMM (another fp32 summation order), a near-tie at the k-th place may flip for a row, and the max bar is wider."""
    from dflash_amd import NativeTarget
    from dflash_amd.candidates import NativeCandidateVerifier
    hf = _moe_target()
    nt = NativeTarget(hf)
    nt.moe_shared_pass = shared_pass
    g = torch.Generator().manual_seed(16)
    P, bs = 45, 16
    prompt = torch.randint(0, 2000, (1, P), generator=g).to(dev())
    cands = torch.randint(0, 2000, (3, bs), generator=g).to(dev())[:C]
    cands[:, 0] = cands[0, 0].clone()
    cache = nt.new_cache(160)
    nt.prefill(prompt, cache)
    taps = [0, 2]
    ver = NativeCandidateVerifier(nt, len(taps))
    ver.part_h.fill_(float("nan"))      # stale shares must never be read
    post = ver.verify(cands, P, cache, taps).clone()
    assert cache.get_seq_length() == P
    for c in range(C):
        c2 = nt.new_cache(160)
        nt.prefill(prompt, c2)
        logits = torch.zeros(32, 2048, dtype=BF16, device=dev())
        nt.verify(cands[c], P, c2, tap_layers=taps, logits_out=logits)
        _, th = nt.verify(cands[c], P, c2, tap_layers=taps, logits_out=logits)
        H.assert_ids_match_where_safe(f"MoE candidate {c} posterior", post[c], logits[:bs].float())
        H.assert_close(f"MoE candidate {c} taps", ver.taps[c], th[:16], max_rel=6e-2 if shared_pass else H.MAX_REL)
        H.assert_close(f"MoE candidate {c} staged K l3", ver.stage_k[3, c, :, :bs], c2.k[3][:, P:P + bs],
                       max_rel=6e-2 if shared_pass else H.KV_MAX_REL)


def test_candidate_loop_is_lossless_on_native_moe_target():
    """dflash_generate_candidate_solutions with a NativeTarget over an MoE model (BASELINE configs[4] class) no longer
    leaves the kernels: committed ids = the target's greedy walk, winners have the maximal tau of their cycle."""
    from dflash_amd import DFlashDraftModel, NativeTarget, dflash_generate_candidate_solutions
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk
    cfg = H.tiny_cfg(num_target_layers=4, target_layer_ids=[0, 2])
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, dtype=BF16))
    hf = _moe_target()
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf)
    prompt = torch.randint(0, 2000, (1, 33), generator=torch.Generator().manual_seed(4)).to(dev())
    n_new = 60
    G = greedy_walk(perm, prompt, n_new + 40).to(dev())
    plan = H.make_plan(64, 16, 17)

    def hook(blk, start, call):
        k = min(plan[call], blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            w = G[start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])

    r = dflash_generate_candidate_solutions(m, nt, prompt, cfg.mask_token_id, n_new, 16, None,
                                            candidate_mode="fixed_prefix_rank", branch_top_k=4, max_candidates=4,
                                            draft_token_hook=hook)
    assert r.output_ids[0].tolist() == G[:33 + n_new].tolist()
    for row in r.cycle_trace:
        assert row["tau"] == max(row["candidate_taus"])
    assert max(row["num_candidates"] for row in r.cycle_trace) > 1


def test_native_candidate_pass_with_wide_blocks_equals_single_verifies():
    """Candidate blocks of 24 rows (two tiles each, two candidates per pass; VERDICT r2 missing #4) vs NativeTarget.verify of
    each block alone on the same prefix: posterior ids on margin-screened rows, tapped rows, staged K/V; keep()."""
    from dflash_amd import NativeTarget
    from dflash_amd.candidates import NativeCandidateVerifier
    hf = _tiny_hf()
    nt = NativeTarget(hf)
    g = torch.Generator().manual_seed(8)
    P, bs = 45, 24
    prompt = torch.randint(0, 2000, (1, P), generator=g).to(dev())
    cands = torch.randint(0, 2000, (2, bs), generator=g).to(dev())
    cands[:, 0] = cands[0, 0]
    cache = nt.new_cache(160)
    nt.prefill(prompt, cache)
    taps = [1, 3]
    ver = NativeCandidateVerifier(nt, len(taps))
    with pytest.raises(ValueError):
        ver.verify(torch.cat([cands, cands[:1]]), P, cache, taps)     # three wide candidates do not fit a pass
    post = ver.verify(cands, P, cache, taps).clone()
    assert cache.get_seq_length() == P and post.shape == (2, bs)
    for c in range(2):
        c2 = nt.new_cache(160)
        nt.prefill(prompt, c2)
        logits = torch.zeros(32, 2048, dtype=BF16, device=dev())
        p1, th = nt.verify(cands[c], P, c2, tap_layers=taps, logits_out=logits)
        H.assert_ids_match_where_safe(f"wide candidate {c} posterior", post[c], logits[:bs].float())
        H.assert_close(f"wide candidate {c} taps", ver.cand_taps(c, bs)[:bs], th[:bs])
        for li in (0, 5):
            H.assert_close(f"wide candidate {c} staged K l{li}", ver.stage_k[li, c, :, :bs], c2.k[li][:, P:P + bs], max_rel=H.KV_MAX_REL)
            H.assert_close(f"wide candidate {c} staged V l{li}", ver.stage_v[li, c, :, :bs], c2.v[li][:, P:P + bs], max_rel=H.KV_MAX_REL)
    ver.keep(1, P, bs, cache)
    assert cache.get_seq_length() == P + bs and torch.equal(cache.k[3][:, P:P + bs], ver.stage_k[3, 1, :, :bs])


@pytest.mark.parametrize("mode,kw", [("branch_beam", dict(branch_top_k=2, max_candidates=4)),
                                     ("fixed_prefix_rank", dict(branch_top_k=4, max_candidates=3))])
def test_candidate_loop_with_wide_blocks_is_lossless(mode, kw):
    """dflash_generate_candidate_solutions with 24-row blocks on the native target: two candidates per pass over the
    weights, several passes per cycle; the committed ids are the target's greedy continuation."""
    from dflash_amd import NativeTarget, dflash_generate_candidate_solutions
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk
    cfg = H.tiny_cfg()
    m = make_model(cfg)
    hf = _tiny_hf()
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf)
    prompt = torch.randint(0, 2000, (1, 33), generator=torch.Generator().manual_seed(4)).to(dev())
    n_new = 90
    G = greedy_walk(perm, prompt, n_new + 60).to(dev())
    plan = H.make_plan(64, 24, 19)

    def hook(blk, start, call):
        k = min(plan[call], blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            w = G[start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])

    r = dflash_generate_candidate_solutions(m, nt, prompt, cfg.mask_token_id, n_new, 24, None, candidate_mode=mode,
                                            draft_token_hook=hook, **kw)
    assert r.output_ids[0].tolist() == G[:33 + n_new].tolist()
    assert max(row["num_candidates"] for row in r.cycle_trace) > 2          # more than one pass of two in some cycle
    if mode == "branch_beam":                                               # (fixed_prefix_rank keeps a 5-token prefix)
        assert max(r.acceptance_lengths) > 16
    for row in r.cycle_trace:
        assert row["tau"] == max(row["candidate_taus"])
