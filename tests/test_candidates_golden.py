"""Multi-candidate verify: the oracle's candidate builders and budget rule against the vectors the
REFERENCE's own functions produced (golden G9, tests/golden/candidates.json), bit for bit on the CPU; the product's
host-side builders (which start from top-k + log-sum-exp instead of the full 15 x V logits) against the same vectors."""
import json
import os

import numpy as np
import pytest
import torch

import helpers as H
from oracle import candidates_oracle as CO

G = json.load(open(os.path.join(H.GOLDEN, "candidates.json")))


def _case(c):
    bits = torch.tensor(np.array(c["logits_bits"], dtype=np.int16))
    return torch.tensor([c["block"]]), bits.view(torch.bfloat16)[None]


def _args(run):
    return dict(branch_depth=run.get("branch_depth", 6), branch_top_k=run["branch_top_k"],
                max_candidates=run["max_candidates"], margin_threshold=run.get("margin_threshold", -1.0),
                fixed_prefix_len=run.get("fixed_prefix_len", 5), sparse_max_positions=run.get("sparse_max_positions", 4))


def _close_meta(got, want):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert set(a) == set(b)
        for k in b:
            if isinstance(b[k], float):
                assert a[k] == pytest.approx(b[k], rel=1e-5, abs=1e-5), k
            else:
                assert a[k] == b[k], k


def test_oracle_builders_match_reference():
    n = 0
    for c in G["cases"]:
        block, logits = _case(c)
        for run in c["runs"]:
            cands, meta, pos = CO.build_candidates(run["mode"], block, logits, **_args(run))
            assert [x[0].tolist() for x in cands] == run["candidates"], run
            assert [int(p) for p in pos] == run["selected_positions"]
            assert meta == run["meta"], run          # same torch ops on the same CPU: bit-identical floats
            n += 1
    assert n == 102


def test_budget_rule_matches_reference():
    from dflash_amd.candidates import resolve_cycle_max_candidates as product_rule
    for b in G["budget"]:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in b.items() if k != "out"}
        assert CO.resolve_cycle_max_candidates(**kw) == b["out"]
        assert product_rule(**kw) == b["out"]


def test_product_builders_from_topk_match_reference():
    """dflash_amd.candidates builds from (top-8 values, indices, log-sum-exp) per row — what dfl_topk_rows returns — not
    from the full logits.  Fed with the exact top-k of the golden logits (computed here on the CPU; the fixtures have no
    ties inside a row) it must reproduce the reference's candidates, positions and, to fp32 rounding, scores."""
    from dflash_amd import candidates as PC
    for c in G["cases"]:
        block, logits = _case(c)
        rows = logits[0].float()
        k = min(8, rows.shape[1])
        vals, idx = torch.topk(rows, k=k, dim=-1)
        top = PC.TopK(vals=vals, idx=idx.to(torch.int64), lse=torch.logsumexp(rows, dim=-1))
        for run in c["runs"]:
            cands, meta, pos = PC.build_candidates(run["mode"], block, top, **_args(run))
            assert [x[0].tolist() for x in cands] == run["candidates"], run
            assert [int(p) for p in pos] == run["selected_positions"]
            _close_meta(meta, run["meta"])


def test_choose_candidate_rule():
    """:592-601 on engineered cases: longest accepted prefix wins; ties go to the higher draft score, then to the lower
    candidate index."""
    blk = torch.tensor([[5, 1, 2, 3, 4], [5, 1, 2, 9, 9], [5, 1, 7, 7, 7]])
    post = torch.tensor([[1, 2, 8, 0, 0], [1, 2, 9, 9, 1], [1, 0, 0, 0, 0]])
    meta = [{"draft_score": 0.0}, {"draft_score": -3.0}, {"draft_score": 9.0}]
    assert CO.choose_candidate(blk, post, meta) == (1, 4, [3, 5, 2])
    post2 = torch.tensor([[1, 2, 8, 0, 0], [1, 2, 0, 0, 0], [1, 0, 0, 0, 0]])
    assert CO.choose_candidate(blk, post2, meta)[:2] == (0, 2)                 # tau 3, 3, 2: score 0.0 beats -3.0
    assert CO.choose_candidate(blk, post2, [{"draft_score": 1.0}] * 3)[:2] == (0, 2)   # full tie: lower index


def test_oracle_candidate_loop_is_lossless_and_picks_the_longest():
    """The restated loop (parity unpinned, see oracle/candidates_oracle.py): on a natural run (random weights, fp32) the
    committed ids are the target's greedy continuation in every mode; with scripted agreement the chosen candidate's tau
    is the maximum over the cycle's candidates."""
    cfg = H.tiny_cfg()
    w = H.draft_weights(cfg, dtype=torch.float32)
    oc = H.oracle_cfg(cfg, "eager")
    e2e = json.load(open(os.path.join(H.GOLDEN, "e2e.json")))["f32_eager/natural"]
    prompt = torch.tensor([e2e["prompt"]])
    for mode in ("branch_beam", "fixed_prefix_rank", "uncertainty_sparse_rank"):
        base = H.tiny_target(dtype=torch.float32)
        r = CO.dflash_generate_candidate_solutions(w, oc, base, prompt, cfg.mask_token_id, e2e["max_new_tokens"], 16,
                                                   None, candidate_mode=mode, branch_top_k=3, max_candidates=4)
        assert r.output_ids[0].tolist() == e2e["ids"], mode
        assert all(t["tau"] == max(t["candidate_taus"]) for t in r.cycle_trace)
        assert max(t["num_candidates"] for t in r.cycle_trace) > 1
