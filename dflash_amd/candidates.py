"""Multi-candidate verify: `dflash_generate_candidate_solutions` with the reference's
signature and return fields (benchmark_candidate_solutions.py:416-741) on the gfx950 kernels.

Per cycle the reference (a) builds up to `max_candidates` variants of the drafted block from the draft
logits (:84-414), (b) verifies them in ONE target call by cloning and batch-repeating the whole target
KV cache (:570-585), (c) keeps the variant with the longest accepted prefix (:586-613).  Here:

* (a) the builders never see the 15 x V logits: `dfl_topk_rows` reduces each drafted row to its 8 largest
  logits (+ indices, log-sum-exp) on the device, the builders run on those few numbers on the host with
  the reference's op sequence and dtypes (several steps are bf16 arithmetic) — pinned by golden G9;
* (b) with a `NativeTarget` the candidates are the request tiles of the ragged-batch kernels: one pass over
  the target's weights for up to 4 candidates (`NativeCandidateVerifier`), all attending the SAME cached
  prefix (no cache clone), each candidate's new K/V rows staged beside the cache (`dfl_attn_head_cand`);
  any other target (e.g. an MoE HF model) is verified through its own forward on a batch-expanded cache,
  as the reference does;
* (c) `dfl_candidate_select` computes every acceptance length, the reference's lexicographic choice and
  the commit on the device; the winner's staged K/V rows are copied into the cache.
"""
from __future__ import annotations

import copy
from collections import Counter, namedtuple
from types import SimpleNamespace
from typing import Sequence

import torch

from . import ops
from .generate import DecodeSession, _taps, cuda_time
from .utils import sample

BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64

# per drafted row (block positions 1 .. bs-1): the k largest logits (fp32 copies of bf16 values), their token ids,
# the row's log-sum-exp — host tensors
TopK = namedtuple("TopK", ["vals", "idx", "lse"])


# ----------------------------------------------------------------------------- builders (host, from top-k)
def _greedy_only(block, positions):
    return ([block.clone()], [{"candidate_idx": 0, "draft_score": 0.0, "replaced_positions": [], "rank_variant": 1}],
            positions)


def _prob_margin(top: TopK, rows) -> torch.Tensor:
    """softmax(x)[top1] - softmax(x)[top2] in fp32 for the given rows (:98-101, :305-307)."""
    p = torch.exp(top.vals[rows, :2] - top.lse[rows, None])
    return p[:, 0] - p[:, 1]


def _branch_beam(block, top: TopK, branch_depth, branch_top_k, max_candidates, margin_threshold):
    """:84-178 — positions 1..min(bs-1, depth) (optionally gated by the top-2 probability margin), beam over
    their top-k tokens scored by log-softmax = logit - lse."""
    if max_candidates < 1:
        raise ValueError("max_candidates must be >= 1")
    bs = int(block.shape[1])
    last = min(bs - 1, branch_depth)
    pos = list(range(1, last + 1)) if last > 0 else []
    if pos and margin_threshold >= 0:
        m = _prob_margin(top, torch.tensor([j - 1 for j in pos])).tolist()
        pos = [j for j, d in zip(pos, m) if d <= margin_threshold]
    if not pos:
        return [block.clone()], [{"candidate_idx": 0, "draft_score": 0.0, "replaced_positions": []}], pos
    k = min(branch_top_k, top.vals.shape[1])
    lp = top.vals - top.lse[:, None]
    base_score = 0.0
    beams = [({}, 0.0)]
    for j in pos:
        base_score += float(lp[j - 1, 0])          # the greedy token is the row's top-1
    for j in pos:
        toks, scores = top.idx[j - 1, :k].tolist(), lp[j - 1, :k].tolist()
        grown = [({**a, j: int(t)}, sc + float(ts)) for a, sc in beams for t, ts in zip(toks, scores)]
        grown.sort(key=lambda e: e[1], reverse=True)
        beams = grown[:max_candidates]
    cands = [block.clone()]
    meta = [{"candidate_idx": 0, "draft_score": base_score, "replaced_positions": []}]
    seen = {tuple(block[0, 1:].tolist())}
    for assign, sc in beams:
        blk = block.clone()
        changed = [int(j) for j, t in assign.items() if int(blk[0, j]) != t]
        for j, t in assign.items():
            blk[0, j] = t
        key = tuple(blk[0, 1:].tolist())
        if key in seen:
            continue
        seen.add(key)
        cands.append(blk)
        meta.append({"candidate_idx": len(cands) - 1, "draft_score": float(sc), "replaced_positions": changed})
        if len(cands) >= max_candidates:
            break
    return cands, meta, pos


def _fixed_prefix_rank(block, top: TopK, vocab, fixed_prefix_len, rank_top_k, max_candidates):
    """:181-249 — candidate r = greedy prefix + the rank-(r+1) token at every suffix position."""
    if max_candidates < 1:
        raise ValueError("max_candidates must be >= 1")
    if rank_top_k < 1:
        raise ValueError("rank_top_k must be >= 1")
    bs = int(block.shape[1])
    s0 = max(1, min(fixed_prefix_len, bs))
    suffix = list(range(s0, bs))
    if not suffix:
        return _greedy_only(block, [])
    total = min(max_candidates, rank_top_k, vocab)
    if total <= 1:
        return _greedy_only(block, suffix)
    vals = top.vals[s0 - 1:, :total].to(BF16)               # the logits are bf16; so is the reference's sum
    scores = vals.transpose(0, 1).sum(dim=1)
    stacked = block.expand(total, -1).clone()
    stacked[:, s0:] = top.idx[s0 - 1:, :total].transpose(0, 1)
    meta = [{"candidate_idx": r, "draft_score": float(scores[r].item()),
             "replaced_positions": [] if r == 0 else suffix, "rank_variant": r + 1} for r in range(total)]
    return [stacked[r:r + 1] for r in range(total)], meta, suffix


def _uncertainty_sparse_rank(block, top: TopK, vocab, fixed_prefix_len, rank_top_k, max_candidates, sparse_max_positions,
                             margin_threshold):
    """:252-379 — one changed position per non-greedy candidate, positions by smallest top-1/top-2 logit margin."""
    if max_candidates < 1:
        raise ValueError("max_candidates must be >= 1")
    if rank_top_k < 1:
        raise ValueError("rank_top_k must be >= 1")
    if sparse_max_positions < 1:
        raise ValueError("sparse_max_positions must be >= 1")
    bs = int(block.shape[1])
    s0 = max(1, min(fixed_prefix_len, bs))
    if bs - s0 <= 0:
        return _greedy_only(block, [])
    rk = min(rank_top_k, vocab)
    if rk <= 1 or max_candidates <= 1:
        return _greedy_only(block, list(range(s0, bs)))
    vals = top.vals[s0 - 1:, :rk].to(BF16)                  # bf16 arithmetic from here on, as in the reference
    idx = top.idx[s0 - 1:, :rk]
    unc = -(vals[:, 0] - vals[:, 1])
    order = torch.argsort(unc, descending=True)
    if margin_threshold >= 0:
        ok = _prob_margin(top, torch.arange(s0 - 1, bs - 1)) <= margin_threshold
        order = order[ok[order]]
    if order.numel() == 0:
        return _greedy_only(block, [])
    sel = order[:min(int(sparse_max_positions), int(order.numel()))]
    sel_pos = sel + s0
    alts = rk - 1
    total = min(max_candidates, 1 + int(sel.numel()) * alts)
    if total <= 1:
        return _greedy_only(block, [int(p) for p in sel_pos.tolist()])
    sv, si, su = vals[sel], idx[sel], unc[sel]
    alt_v, alt_t = sv[:, 1:], si[:, 1:]
    comp = su[:, None] * 1e6 + alt_v
    n = total - 1
    top_c, flat = torch.topk(comp.reshape(-1), k=n, dim=0)
    pc = torch.div(flat, alts, rounding_mode="floor")
    ac = flat % alts
    ch_pos, ch_tok = sel_pos[pc], alt_t[pc, ac]
    base_score = sv[:, 0].sum()
    cand_scores = base_score - sv[pc, 0] + alt_v[pc, ac]
    stacked = block.expand(total, -1).clone()
    stacked[torch.arange(1, total), ch_pos] = ch_tok
    meta = [{"candidate_idx": 0, "draft_score": float(base_score.item()), "replaced_positions": [], "rank_variant": 1}]
    for i in range(n):
        meta.append({"candidate_idx": i + 1, "draft_score": float(cand_scores[i].item()),
                     "replaced_positions": [int(ch_pos[i].item())], "rank_variant": int(ac[i].item()) + 2,
                     "composite_score": float(top_c[i].item())})
    return [stacked[i:i + 1] for i in range(total)], meta, [int(p) for p in sel_pos.tolist()]


def build_candidates(mode: str, block: torch.Tensor, top: TopK, *, vocab: int = 1 << 30, branch_depth=6, branch_top_k=2,
                     max_candidates=4, margin_threshold=-1.0, fixed_prefix_len=5, sparse_max_positions=4):
    """The mode switch of :530-565 on host tensors.  block int64 [1, bs] with the greedy tokens filled in."""
    have = int(top.vals.shape[1])
    need = min(max_candidates, branch_top_k) if mode == "fixed_prefix_rank" else branch_top_k
    if min(need, vocab) > have:
        raise NotImplementedError(f"top-{need} wanted but {have} logits per row are kept (dfl_topk_rows keeps at most 8)")
    if mode == "fixed_prefix_rank":
        return _fixed_prefix_rank(block, top, vocab, fixed_prefix_len, branch_top_k, max_candidates)
    if mode == "uncertainty_sparse_rank":
        return _uncertainty_sparse_rank(block, top, vocab, fixed_prefix_len, branch_top_k, max_candidates,
                                        sparse_max_positions, margin_threshold)
    return _branch_beam(block, top, branch_depth, branch_top_k, max_candidates, margin_threshold)


def resolve_cycle_max_candidates(*, enabled, max_candidates, cycle_idx, last_accept_ratio, budgets, accept_thresholds,
                                 warmup_cycles, probe_interval) -> int:
    """:382-413."""
    if not enabled:
        return int(max_candidates)
    low, mid, high = budgets
    hi_acc, mid_acc = accept_thresholds
    if cycle_idx < warmup_cycles or (probe_interval > 0 and cycle_idx > 0 and cycle_idx % probe_interval == 0):
        pick = high
    elif last_accept_ratio is None:
        pick = high
    elif last_accept_ratio >= hi_acc:
        pick = low
    elif last_accept_ratio >= mid_acc:
        pick = mid
    else:
        pick = high
    return int(max(1, min(max_candidates, pick)))


# ----------------------------------------------------------------------------- verify: native target
class NativeCandidateVerifier:
    """Up to 4 candidate blocks of <= 16 rows (2 of 17..32 rows: two tiles each) per pass through the target's weights
    (`gemm_batch.hip`, request tile = candidate tile), all on ONE cached prefix; new K/V rows and tapped rows of every
    candidate staged, the winner's copied in."""

    MT = 4

    def __init__(self, target, n_taps: int, max_splits: int = 32):
        from .target import NativeTarget
        if not isinstance(target, NativeTarget):
            raise TypeError("NativeCandidateVerifier needs a dflash_amd.NativeTarget")
        t, MT = target, self.MT
        self.t, self.max_splits = t, max_splits
        dev = t.device
        z = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        H, ks = t.H, ops.batch_ksplit
        self.ids = z(MT, 16, dt=I64)
        self.dyn = z(MT, 8, dt=I32)
        self.h, self.ss_emb = z(MT, 16, H), z(MT, 16, dt=F32)
        self.xn, self.attn, self.act = z(MT, 16 * H), z(MT, 16 * t.q_dim), z(MT, 16 * t.I)
        self.xq = z(MT, 16, t.nqkv)
        self.part_h = z(max(ks(t.q_dim), ks(t.I), t.moe_nsplit if getattr(t, "is_moe", False) else 0) * MT * 16 * H, dt=F32)
        nmax, kmax = max(t.V, 2 * t.I, t.nqkv), max(H, t.I, t.q_dim)
        self.gws = torch.zeros(max(ops.lib().dfl_gemm_batch_ws_bytes(n, k) for n, k in ((nmax, H), (H, kmax))),
                               dtype=torch.uint8, device=dev)
        self.head_ws = torch.zeros(MT * ops.lib().dfl_attn_head_ws_bytes(t.n_q, max_splits, 2), dtype=torch.uint8,
                                   device=dev)
        self.stage_k, self.stage_v = z(t.L, MT, t.n_kv, 32, 128), z(t.L, MT, t.n_kv, 32, 128)
        self.taps = z(MT, 16, max(1, n_taps) * H)
        self.post = z(MT, 16, dt=I64)
        self.src = dict(xn=ops.brows_frag(self.xn), attn=ops.brows_frag(self.attn), act=ops.brows_frag(self.act))

    @torch.inference_mode()
    def verify(self, cands: torch.Tensor, start: int, cache, tap_layers: Sequence[int]) -> torch.Tensor:
        """cands int64 [C, bs] (device) at positions start.., C <= 4 for bs <= 16, C <= 2 for bs 17..32 -> posterior ids
        [C, bs] (a view of the verifier's buffer).  Afterwards stage_k / stage_v [L, c, n_kv, :bs] and cand_taps(c, bs)
        hold candidate c's rows."""
        t, MT, H = self.t, self.MT, self.t.H
        C, bs = cands.shape
        TPR = 1 if bs <= 16 else 2              # 16-row tiles per candidate
        if not 1 <= bs <= 32 or not 1 <= C <= MT // TPR:
            raise ValueError("a pass verifies 1..4 candidates of 1..16 rows or 1..2 candidates of 17..32 rows")
        if start + bs > cache.max_rows:
            raise ValueError("target KV cache too small")
        if t.lm_wp is None:
            t.lm_wp = ops.pack_weight(t.lm_head.weight.detach().to(BF16).contiguous())
        tl = list(tap_layers)
        if tl and max(tl) >= t.L - 1:
            raise NotImplementedError("tapping the last layer (post-norm state) is not supported")
        cos, sin = t._rope_tab(start + bs + 64)
        rec = [[start, 0, max(0, min(16, bs - 16 * (j % TPR))) if j < C * TPR else 0, start, start, 0, 0, 0] for j in range(MT)]
        self.dyn.copy_(torch.tensor(rec, dtype=I32))             # one record per TILE (valid rows of the tile)
        self.ids.view(-1, 16 * TPR)[:C, :bs].copy_(cands)
        dyn, s, R = self.dyn, self.src, C * TPR
        ops.embed_rows_batch(t.embed, self.ids, R, self.h, H, self.ss_emb, dyn, ops.DYN_BS)
        slots = {}
        for j, l in enumerate(tl):
            slots.setdefault(l, []).append(j)
        pend, ptap, pdup = 0, None, ()

        def spread(dups):
            for a, b in dups:
                self.taps[:, :, b * H:(b + 1) * H].copy_(self.taps[:, :, a * H:(a + 1) * H])

        pns = None   # share count of the pending sums when they are MoE expert shares, not K parts
        for i, lw in enumerate(t.layers):
            ops.norm_frag_batch(self.h, R, lw["ln1"], t.eps, self.xn, dyn, ops.DYN_BS,
                                part=self.part_h if pend else None, N=H, K=pend, tap=ptap, nsplit=pns)
            spread(pdup)
            ops.gemm_resid_batch(lw["qkv"], s["xn"], R, t.nqkv, H, self.xq, add_residual=False, ws=self.gws, dyn=dyn)
            ops.attn_head_cand(xq=self.xq[:R], q_col=0, k_col=t.q_dim, v_col=t.q_dim + t.kv_dim, n_q=t.n_q, n_kv=t.n_kv,
                               q_norm_w=lw["q_norm"], k_norm_w=lw["k_norm"], eps=t.eps, cos_tab=cos, sin_tab=sin,
                               kcache=cache.k[i], vcache=cache.v[i], scale=128 ** -0.5, S=start, bs=bs,
                               ws=self.head_ws, max_splits=self.max_splits, out_frag=self.attn,
                               k_out=self.stage_k[i], v_out=self.stage_v[i], q_tiles=TPR)
            ops.gemm_f32_batch(lw["o"], s["attn"], R, H, t.q_dim, self.part_h, dyn)
            ops.norm_frag_batch(self.h, R, lw["ln2"], t.eps, self.xn, dyn, ops.DYN_BS, part=self.part_h, N=H, K=t.q_dim)
            if "gu_e" in lw:   # sparse-MoE layer: every candidate routes its own rows (round 3)
                pns, pend = t.moe_mlp_tiles(lw, R, MT, dyn, self.xn, self.part_h), 1
            else:
                ops.gemm_silu_mul_batch(lw["gu"], s["xn"], R, t.I, H, self.act, self.gws, dyn)
                ops.gemm_f32_batch(lw["down"], s["act"], R, H, t.I, self.part_h, dyn)
                pns, pend = None, t.I
            sl = slots.get(i, ())
            ptap = self.taps[:, :, sl[0] * H:(sl[0] + 1) * H] if sl else None
            pdup = [(sl[0], b) for b in sl[1:]]
        ops.norm_frag_batch(self.h, R, t.norm, t.eps, self.xn, dyn, ops.DYN_BS, part=self.part_h, N=H, K=pend, tap=ptap,
                            nsplit=pns)
        spread(pdup)
        ops.gemm_argmax_batch(t.lm_wp, s["xn"], R, t.V, H, 0, 16, self.gws, self.post, 0, dyn, nrows_dyn_word=ops.DYN_BS)
        return self.post.view(-1, 16 * TPR)[:C, :bs]

    def cand_taps(self, c: int, bs: int) -> torch.Tensor:
        """Tapped rows [bs padded to its tiles, n_taps * H] of candidate c of the last pass."""
        tpr = 1 if bs <= 16 else 2
        return self.taps.view(-1, 16 * tpr, self.taps.shape[2])[c]

    def keep(self, win: int, start: int, bs: int, cache) -> None:
        """The winner's new K/V rows become cache rows start..start+bs-1 (the reference keeps the winner's copy of the
        whole batch-expanded cache, :604-608)."""
        cache.k[:, :, start:start + bs].copy_(self.stage_k[:, win, :, :bs])
        cache.v[:, :, start:start + bs].copy_(self.stage_v[:, win, :, :bs])
        cache.length = start + bs


# ----------------------------------------------------------------------------- the loop
@torch.inference_mode()
def dflash_generate_candidate_solutions(model, target, input_ids: torch.Tensor, mask_token_id: int, max_new_tokens: int,
                                        block_size: int, stop_token_ids, branch_depth: int = 6, branch_top_k: int = 2,
                                        max_candidates: int = 4, margin_threshold: float = -1.0,
                                        candidate_mode: str = "branch_beam", fixed_prefix_len: int = 5,
                                        sparse_max_positions: int = 4, adaptive_candidates: bool = False,
                                        adaptive_budgets=(1, 4, 8), adaptive_accept_thresholds=(0.85, 0.65),
                                        adaptive_warmup_cycles: int = 2, adaptive_probe_interval: int = 5,
                                        temperature: float = 0.0, collect_profile: bool = False,
                                        draft_token_hook=None) -> SimpleNamespace:
    """benchmark_candidate_solutions.py:416-741: signature (plus the test hook), cycle_trace rows, candidate_summary
    and timing fields of the reference."""
    if temperature >= 1e-5:
        raise ValueError("benchmark_candidate_solutions.py currently supports only temperature=0.0")   # :438-439
    if block_size > 32:
        raise ValueError("candidate blocks take 1..32 rows")
    if max_candidates > 8:
        raise ValueError("at most 8 candidates per cycle (dfl_candidate_select)")
    s = DecodeSession(model, target, input_ids, mask_token_id=mask_token_id, max_new_tokens=max_new_tokens,
                      max_block_size=block_size, stop_token_ids=stop_token_ids, temperature=0.0,
                      draft_token_hook=draft_token_hook)
    dev, V = s.dev, model.config.vocab_size
    t0 = cuda_time()
    s.prefill()
    ttft = cuda_time() - t0
    native = s.native
    ver = NativeCandidateVerifier(target, len(model.target_layer_ids)) if (native and s.use_draft) else None
    BW = 16 if block_size <= 16 else 32
    s.draft_logits = torch.zeros(BW, V, dtype=BF16, device=dev) if s.use_draft else None
    result = torch.zeros(12, dtype=I32, device=dev)
    cand_buf = torch.zeros(8, BW, dtype=I64, device=dev)
    post_buf = torch.zeros(8, BW, dtype=I64, device=dev)
    score_buf = torch.zeros(8, dtype=F32, device=dev)
    decode_start = cuda_time()
    taus, trace, last_ratio, first_done = [], [], None, False
    cand_sum = verify_calls = 0
    budget_counts: Counter = Counter()
    while s.start < s.max_length:
        ev = {}

        def mark(key, which):
            if collect_profile:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.setdefault(key, [None, None])[which] = e

        mark("cycle", 0)
        start = s.start
        bs = min(block_size, s.max_length - start)
        budget = resolve_cycle_max_candidates(enabled=adaptive_candidates, max_candidates=max_candidates,
                                              cycle_idx=len(taus), last_accept_ratio=last_ratio,
                                              budgets=adaptive_budgets, accept_thresholds=adaptive_accept_thresholds,
                                              warmup_cycles=adaptive_warmup_cycles, probe_interval=adaptive_probe_interval)
        budget_counts[budget] += 1
        blk = s.block[:, :bs]
        blk.copy_(s.output_ids[:, start:start + bs])
        cands = [blk]
        meta, sel = [{"candidate_idx": 0, "draft_score": 0.0, "replaced_positions": []}], []
        if bs > 1:
            mark("draft", 0)
            s._draft(blk, bs, 1)                                   # draft forward + lm_head: greedy ids AND bf16 logits
            kk = min(8, V, max(2, branch_top_k, budget if candidate_mode == "fixed_prefix_rank" else 2))
            tv, ti, lse = ops.topk_rows(s.draft_logits[1:bs], kk)
            top = TopK(tv[:, :kk].cpu(), ti[:, :kk].to(I64).cpu(), lse.cpu())
            host_blk = blk.cpu()
            cands, meta, sel = build_candidates(candidate_mode, host_blk, top, vocab=V, branch_depth=branch_depth,
                                                branch_top_k=branch_top_k, max_candidates=budget,
                                                margin_threshold=margin_threshold, fixed_prefix_len=fixed_prefix_len,
                                                sparse_max_positions=sparse_max_positions)
            mark("draft", 1)
        C = len(cands)
        cand_sum += C
        stacked = torch.cat(cands, dim=0).to(dev)
        cand_buf[:C, :bs].copy_(stacked)
        score_buf[:C].copy_(torch.tensor([float(m["draft_score"]) for m in meta], dtype=F32))
        # ---- verify every candidate against the same prefix (:567-585)
        mark("target", 0)
        hs_all = None
        parked = None
        if ver is not None:
            per = ver.MT if bs <= 16 else ver.MT // 2     # candidates per pass (blocks of 17..32 rows take two tiles each)
            if C <= per:
                post_buf[:C, :bs].copy_(ver.verify(cand_buf[:C, :bs], start, s.tcache, model.target_layer_ids))
            else:       # several passes: each pass's staged rows are set aside before the next overwrites them
                parked = []
                for c0 in range(0, C, per):
                    c1 = min(C, c0 + per)
                    post_buf[c0:c1, :bs].copy_(ver.verify(cand_buf[c0:c1, :bs], start, s.tcache, model.target_layer_ids))
                    for c in range(c1 - c0):
                        parked.append((ver.stage_k[:, c, :, :bs].clone(), ver.stage_v[:, c, :, :bs].clone(),
                                       ver.cand_taps(c, bs).clone()))
        elif native:   # bs == 1 tail without a draft: the plain verify
            post, _ = target.verify(cand_buf[0, :bs], start, s.tcache)
            post_buf[:1, :bs].copy_(post)
        else:
            vc = copy.deepcopy(s.tcache)
            if C > 1:
                vc.batch_repeat_interleave(C)
            out = target(stacked, position_ids=s.position_ids[:, start:start + bs].repeat(C, 1), past_key_values=vc,
                         use_cache=True, output_hidden_states=s.use_draft)
            post_buf[:C, :bs].copy_(sample(out.logits, 0.0))
            hs_all = out.hidden_states
        mark("target", 1)
        verify_calls += 1
        # ---- acceptance lengths, choice, commit (:586-613)
        ops.set_dyn(s.dyn, 0, 0, bs, start)
        ops.candidate_select(cand_buf[:C], post_buf[:C], score_buf, bs, s.output_ids[0], s.dyn, s.stop_t, result)
        res = result.tolist()
        acc, win = res[0], res[3]
        tau = acc + 1
        s.start = start + tau
        if ver is not None:
            if parked is not None:
                k_, v_, tp = parked[win]
                s.tcache.k[:, :, start:start + bs].copy_(k_)
                s.tcache.v[:, :, start:start + bs].copy_(v_)
                s.tcache.length = start + bs
                winner_taps = tp
            else:
                ver.keep(win, start, bs, s.tcache)
                winner_taps = ver.cand_taps(win, bs)
            s.tcache.crop(s.start)
            if s.use_draft:
                s.taps_buf[:winner_taps.shape[0]].copy_(winner_taps)
                s.target_hidden = s.taps_buf[None, :tau]
        elif native:
            s.tcache.crop(s.start)
        else:
            if C > 1:
                vc.batch_select_indices(torch.tensor([win], dtype=torch.long, device=dev))
            s.tcache = vc
            s.tcache.crop(s.start)
            if s.use_draft:
                s.target_hidden = _taps([h[win:win + 1] for h in hs_all], model.target_layer_ids)[:, :tau, :]
        taus.append(tau)
        row = {"cycle_idx": len(taus) - 1, "generated_tokens_before": int(start - s.n_in),
               "effective_block_size": int(bs), "tau": int(tau), "acceptance_ratio": float(tau / max(1, bs)),
               "num_candidates": int(C), "cycle_max_candidates": int(budget),
               "selected_positions": [int(x) for x in sel], "chosen_candidate_idx": int(win),
               "candidate_taus": [int(a) + 1 for a in res[4:4 + C]],
               "candidate_draft_scores": [float(m["draft_score"]) for m in meta],
               "candidate_rank_variants": [int(m.get("rank_variant", 1)) for m in meta]}
        mark("cycle", 1)
        if collect_profile:
            row["_events"] = ev
        trace.append(row)
        last_ratio = float(tau / max(1, bs))
        s.stopped = bool(s.stop_always or res[2])
        if s.stopped:
            break
        if not first_done:
            decode_start = cuda_time()          # :660-662: after the first cycle
            first_done = True
    output_ids = s.finish()
    n_out = output_ids.shape[1] - s.n_in
    decode_s = cuda_time() - decode_start
    summary = {"candidate_mode": str(candidate_mode), "fixed_prefix_len": int(fixed_prefix_len),
               "sparse_max_positions": int(sparse_max_positions), "adaptive_candidates": bool(adaptive_candidates),
               "adaptive_budgets": [int(x) for x in adaptive_budgets],
               "adaptive_accept_thresholds": [float(x) for x in adaptive_accept_thresholds],
               "adaptive_warmup_cycles": int(adaptive_warmup_cycles),
               "adaptive_probe_interval": int(adaptive_probe_interval),
               "adaptive_budget_counts": {str(k): int(v) for k, v in sorted(budget_counts.items())},
               "avg_candidates_per_cycle": float(cand_sum / max(1, len(taus))),
               "candidate_verify_calls": int(verify_calls), "candidate_count_sum": int(cand_sum)}
    profile = None
    if collect_profile:
        torch.cuda.synchronize()
        tot = {"draft": 0.0, "target": 0.0, "cycle": 0.0}
        for row in trace:
            e = row.pop("_events")
            for k in tot:
                p = e.get(k)
                sec = p[0].elapsed_time(p[1]) / 1000.0 if p and p[0] is not None and p[1] is not None else 0.0
                row[f"{k}_s"] = float(sec)
                tot[k] += sec
        den = max(1e-12, tot["draft"] + tot["target"])
        profile = {"target_prefill_s": float(ttft), "target_decode_s": float(tot["target"]),
                   "draft_decode_s": float(tot["draft"]), "cycle_decode_s_sum": float(tot["cycle"]),
                   "decode_wall_s": float(decode_s), "profiled_cycles": int(len(trace)),
                   "draft_share_decode": float(tot["draft"] / den), "target_share_decode": float(tot["target"] / den)}
    return SimpleNamespace(output_ids=output_ids, num_input_tokens=s.n_in, num_output_tokens=n_out,
                           time_to_first_token=ttft, time_per_output_token=decode_s / max(1, n_out),
                           acceptance_lengths=taus, cycle_trace=trace, candidate_summary=summary,
                           profile_summary=profile)
