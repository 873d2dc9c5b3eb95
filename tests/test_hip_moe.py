"""Sparse-MoE target on the native verify (BASELINE configs[4], Qwen3-Coder-30B-A3B class): router, grouped expert
GEMMs, and the whole verify against the HF Qwen3MoeForCausalLM forward."""
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def dev():
    return torch.device("cuda", 0)


def _moe_hf(layers=4, E=16, top_k=4, Ie=256, mlp_only=(), dtype=BF16, seed=41, norm_topk=True, hidden=512, heads=4, kv=2):
    tf = pytest.importorskip("transformers")
    cfg = tf.Qwen3MoeConfig(vocab_size=2048, hidden_size=hidden, intermediate_size=2 * hidden, moe_intermediate_size=Ie,
                            num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=kv, head_dim=128,
                            num_experts=E, num_experts_per_tok=top_k, decoder_sparse_step=1, norm_topk_prob=norm_topk,
                            max_position_embeddings=4096, rms_norm_eps=1e-6, tie_word_embeddings=False,
                            rope_parameters={"rope_type": "default", "rope_theta": 1e6}, mlp_only_layers=list(mlp_only))
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(dev()):
            m = tf.Qwen3MoeForCausalLM(cfg)
    finally:
        torch.set_default_dtype(prev)
    with torch.no_grad():   # the default init leaves the router nearly flat: give the routing something to decide
        for layer in m.model.layers:
            if hasattr(layer.mlp, "gate"):
                layer.mlp.gate.weight.mul_(8.0)
    return m.eval()


def _tie_free_logits(rows, E, g):
    """bf16 router logits in about +-9 with no two equal values in a row (random bf16 draws tie: 8 mantissa bits), so the
    top-k selection of every row is unambiguous in any correct implementation."""
    pool = torch.unique((torch.randn(20000, generator=g) * 3).to(BF16))
    assert pool.numel() >= E
    return torch.stack([pool[torch.randperm(pool.numel(), generator=g)[:E]] for _ in range(rows)])


@pytest.mark.parametrize("E,top_k,norm", [(16, 4, True), (128, 8, True), (24, 2, False), (200, 8, True)])
def test_moe_route_matches_torch(E, top_k, norm):
    """dfl_moe_route vs Qwen3MoeTopKRouter.forward's arithmetic (fp32 softmax of the bf16 logits, top-k, renormalise,
    cast): the dense weight matrix, the active flags and the ascending list; rows beyond the block route nowhere."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(E + top_k)
    ep = (E + 15) // 16 * 16
    logits = torch.zeros(16, ep, dtype=BF16)
    logits[:, :E] = _tie_free_logits(16, E, g)
    rows = 11
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 0, rows, 0)
    wt = torch.full((16, E), 7.0, dtype=BF16, device=dev())
    active = torch.full((E,), 9, dtype=torch.int32, device=dev())
    lst = torch.zeros(E, dtype=torch.int32, device=dev())
    n = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.moe_route(logits.to(dev()), E, top_k, norm, wt, active, lst, n, dyn=dyn, dyn_word=ops.DYN_BS)
    p = torch.softmax(logits[:rows, :E].float(), dim=-1)
    v, i = torch.topk(p, top_k, dim=-1)
    if norm:
        v = v / v.sum(dim=-1, keepdim=True)
    ref = torch.zeros(16, E, dtype=torch.float32)
    ok_rows = []
    for m in range(rows):      # rows whose k-th and (k+1)-th probabilities differ: the selection is unambiguous
        srt = p[m].sort(descending=True).values
        if len(set(srt[:top_k + 1].tolist())) == top_k + 1:
            ok_rows.append(m)
        ref[m, i[m]] = v[m].to(BF16).float()
    got = wt.float().cpu()
    assert len(ok_rows) == rows               # the logits are tie-free by construction: every row is checked
    assert torch.equal(got[rows:], torch.zeros(16 - rows, E))
    for m in ok_rows:
        assert (got[m] != 0).sum() == top_k
        assert torch.equal(got[m] != 0, ref[m] != 0), m
        assert torch.allclose(got[m], ref[m], rtol=2 ** -7, atol=1e-4)
    act_ref = (got != 0).any(dim=0)
    assert torch.equal(active.cpu() != 0, act_ref)
    k = int(n)
    assert k == int(act_ref.sum()) and lst[:k].cpu().tolist() == act_ref.nonzero()[:, 0].tolist()


@pytest.mark.parametrize("K,E,top_k,rows,with_norm", [(2048, 128, 8, 16, True), (2048, 128, 8, 5, True), (512, 16, 4, 11, True),
                                                       (4096, 200, 8, 16, True), (2048, 128, 8, 9, False), (1024, 24, 2, 16, False)])
def test_moe_router_one_launch_equals_the_three(K, E, top_k, rows, with_norm):
    """dfl_moe_router (RMSNorm + gate Linear + routing in one launch, the logits handed to the last workgroup through
    write-through stores) against dfl_norm_pack + dfl_gemm_resid + dfl_moe_route: same fragments, same routing weights,
    flags, list and count — over repeated launches (the arrival ticket re-arms itself) on changing inputs; the logits
    agree within the two GEMMs' summation orders and the normalised rows with Qwen3MoeRMSNorm's rounding points."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(K + E + rows)
    ep = (E + 15) // 16 * 16
    rw = torch.zeros(ep, K, dtype=BF16)
    rw[:E] = (torch.randn(E, K, generator=g) * 0.3).to(BF16)
    wp = ops.pack_weight(rw.to(dev()))
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF16).to(dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 0, rows, 0)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev())

    def outs():
        return dict(wt=torch.full((16, E), 7.0, dtype=BF16, device=dev()), active=torch.full((E,), 9, dtype=torch.int32, device=dev()),
                    lst=torch.full((E,), -1, dtype=torch.int32, device=dev()), n=torch.zeros(1, dtype=torch.int32, device=dev()),
                    rlog=torch.zeros(16, ep, dtype=BF16, device=dev()), xn=torch.full((16 * K,), 3.0, dtype=BF16, device=dev()))

    for step in range(3):
        h = torch.randn(16, K + 64, generator=g).to(BF16).to(dev())[:, :K]      # a row stride wider than K
        a, b = outs(), outs()
        ops.norm_pack(norm_w=nw, frag=a["xn"], H=K, eps=1e-6, resid_in=h.contiguous(), dyn=dyn, dyn_word=ops.DYN_BS)
        if not with_norm:
            b["xn"].copy_(a["xn"])
        ops.gemm_resid(wp, ops.rows_frag(a["xn"]), ep, K, a["rlog"], add_residual=False, dyn=dyn)
        ops.moe_route(a["rlog"], E, top_k, True, a["wt"], a["active"], a["lst"], a["n"], dyn=dyn, dyn_word=ops.DYN_BS)
        ops.moe_router(h=h if with_norm else None, norm_w=nw if with_norm else None, eps=1e-6, xn=b["xn"], wp_router=wp, K=K, E=E,
                       top_k=top_k, norm_topk=True, rlog=b["rlog"], wt=b["wt"], active=b["active"], lst=b["lst"], n_active=b["n"],
                       ticket=ticket, dyn=dyn, dyn_word=ops.DYN_BS)
        torch.cuda.synchronize()
        assert int(ticket) == 0
        if with_norm:    # Qwen3MoeRMSNorm: fp32 normalise, cast to bf16, times weight; absent rows are zero fragments
            hf = h.float()
            ref = (nw.float() * (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF16).float()).to(BF16)
            ref[rows:] = 0
            got = b["xn"].view(K // 8, 16, 8).permute(1, 0, 2).reshape(16, K)
            assert torch.equal(got, ref), step
        assert torch.equal(a["xn"], b["xn"]), step
        # the two GEMMs sum K in different orders: bf16 logits may differ by a rounding step
        d = (a["rlog"][:, :E].float() - b["rlog"][:, :E].float()).abs().max()
        assert d <= 2 ** -6 * a["rlog"].float().abs().max(), (step, d)
        # routing of the fused launch == dfl_moe_route applied to ITS logits (exact), and == the split path wherever
        # the logits are bit-equal (nearly always)
        c = outs()
        ops.moe_route(b["rlog"], E, top_k, True, c["wt"], c["active"], c["lst"], c["n"], dyn=dyn, dyn_word=ops.DYN_BS)
        assert torch.equal(b["wt"], c["wt"]) and torch.equal(b["active"], c["active"]) and int(b["n"]) == int(c["n"]), step
        k = int(b["n"])
        assert torch.equal(b["lst"][:k], c["lst"][:k]), step
        if torch.equal(a["rlog"], b["rlog"]):
            assert torch.equal(a["wt"], b["wt"]) and torch.equal(a["lst"][:k], b["lst"][:k])


@pytest.mark.parametrize("E,top_k", [(128, 8), (16, 4), (200, 8), (2, 1)])
def test_moe_route_nan_row_and_edges(E, top_k):
    """A row of NaN logits (a NaN anywhere upstream makes the whole softmax NaN) must not fault or hang: it routes to
    experts 0 .. k-1 with NaN weights — the NaN goes on through the expert sums as it does through torch.topk +
    index_add_ — while the other rows route as always; E = 2 / top-1 and -inf logits are handled too."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(E)
    ep = (E + 15) // 16 * 16
    logits = torch.zeros(16, ep, dtype=BF16)
    logits[:, :E] = _tie_free_logits(16, E, g) if E > 2 else torch.tensor([[0.5, -0.25]] * 16, dtype=BF16)
    logits[3, :E] = float("nan")
    if E > 2:
        logits[5, 1] = float("-inf")     # probability exactly 0: never selected while k others are positive
    wt = torch.full((16, E), 7.0, dtype=BF16, device=dev())
    active = torch.zeros(E, dtype=torch.int32, device=dev())
    lst = torch.zeros(E, dtype=torch.int32, device=dev())
    n = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.moe_route(logits.to(dev()), E, top_k, True, wt, active, lst, n)
    torch.cuda.synchronize()
    got = wt.float().cpu()
    assert torch.isnan(got[3, :top_k]).all() and torch.count_nonzero(got[3, top_k:]) == 0
    for m in (0, 5, 15):
        p = torch.softmax(logits[m, :E].float(), dim=-1)
        v, i = torch.topk(p, top_k)
        ref = torch.zeros(E)
        ref[i] = (v / v.sum()).to(BF16).float()
        assert torch.equal(got[m] != 0, ref != 0) and torch.allclose(got[m], ref, rtol=2 ** -7, atol=1e-4), m
    if E > 2:
        assert got[5, 1] == 0
    k = int(n)
    assert lst[:k].cpu().tolist() == (torch.nan_to_num(got, nan=1.0) != 0).any(dim=0).nonzero()[:, 0].tolist()


def test_moe_router_hand_off_under_load():
    """The logits of dfl_moe_router cross workgroups INSIDE the launch (write-through stores, a returning ticket, sc1
    loads by the last arriver).  A stale read there shows only under load and with changing data: 300 launches on fresh
    rows each, while a side stream keeps the memory system busy with large copies; every launch's routing must equal
    dfl_moe_route applied to the logits the launch left in memory, and the logits must move with the rows."""
    from dflash_amd import ops
    K, E, top_k = 2048, 128, 8
    g = torch.Generator().manual_seed(9)
    wp = ops.pack_weight((torch.randn(E, K, generator=g) * 0.3).to(BF16).to(dev()))
    nw = torch.ones(K, dtype=BF16, device=dev())
    ticket = torch.zeros(1, dtype=torch.int32, device=dev())
    n_it = 300
    hs = torch.randn(n_it, 16, K, generator=g).to(BF16).to(dev())
    xn = torch.zeros(16 * K, dtype=BF16, device=dev())
    rlog = torch.zeros(n_it, 16, E, dtype=BF16, device=dev())
    wt = torch.zeros(n_it, 16, E, dtype=BF16, device=dev())
    active = torch.zeros(E, dtype=torch.int32, device=dev())
    lst = torch.zeros(E, dtype=torch.int32, device=dev())
    ns = torch.zeros(n_it, 1, dtype=torch.int32, device=dev())
    big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev())
    big_b = torch.empty_like(big_a)
    side = torch.cuda.Stream()
    stop = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(40):
            big_b.copy_(big_a)
        stop.record()
    for i in range(n_it):
        ops.moe_router(h=hs[i], norm_w=nw, eps=1e-6, xn=xn, wp_router=wp, K=K, E=E, top_k=top_k, norm_topk=True, rlog=rlog[i],
                       wt=wt[i], active=active, lst=lst, n_active=ns[i], ticket=ticket)
    busy_at_end = not stop.query()     # (informational: the copies outlasted the launches)
    torch.cuda.synchronize()
    assert int(ticket) == 0
    ref_wt = torch.zeros(16, E, dtype=BF16, device=dev())
    ref_n = torch.zeros(1, dtype=torch.int32, device=dev())
    bad = 0
    for i in range(n_it):
        ops.moe_route(rlog[i], E, top_k, True, ref_wt, active, lst, ref_n)
        bad += int(not torch.equal(ref_wt, wt[i])) + int(int(ref_n) != int(ns[i]))
    assert bad == 0, (bad, busy_at_end)
    assert len({rlog[i].float().sum().item() for i in range(0, n_it, 37)}) > 5      # the logits follow the rows


@pytest.mark.parametrize("pair_kernel", [False, True])
def test_grouped_expert_gemms_match_torch(pair_kernel):
    """dfl_gemm_silu_mul_experts + dfl_moe_down against the HF experts loop in fp32 (Qwen3MoeExperts.forward): only the
    active experts' outputs are computed, the routing-weighted sum over experts comes out as K-part sums."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(5)
    E, I, Hd, rows = 16, 256, 512, 13
    gu = (torch.randn(E, 2 * I, Hd, generator=g) * 0.05).to(BF16).to(dev())
    dn = (torch.randn(E, Hd, I, generator=g) * 0.05).to(BF16).to(dev())
    x = torch.randn(16, Hd, generator=g).to(BF16).to(dev())
    x[rows:] = 0
    wt = torch.zeros(16, E, dtype=BF16, device=dev())
    for m in range(rows):
        sel = torch.randperm(E, generator=g)[:3]
        wt[m, sel] = torch.rand(3, generator=g).to(BF16).to(dev())
    active = (wt != 0).any(dim=0).to(torch.int32)
    lst = torch.zeros(E, dtype=torch.int32, device=dev())
    k = int(active.sum())
    lst[:k] = active.nonzero()[:, 0].to(torch.int32)
    n = torch.tensor([k], dtype=torch.int32, device=dev())
    assert 0 < k < E
    gu_p = torch.stack([ops.pack_weight_gateup(gu[e, :I].contiguous(), gu[e, I:].contiguous()) for e in range(E)])
    dn_p = torch.stack([ops.pack_weight(dn[e].contiguous()) for e in range(E)])
    xf = torch.empty(16 * Hd, dtype=BF16, device=dev())
    ops.pack_rows(x, 16, xf)
    act = torch.full((E, 16 * I), float("nan"), dtype=BF16, device=dev())
    if pair_kernel:   # K <= 2048: one LDS meeting per (gate, up) tile pair
        ops.moe_gate_up(gu_p, xf, E, I, Hd, act, lst, n)
    else:
        ops.gemm_silu_mul_experts(gu_p, ops.rows_frag(xf), E, I, Hd, act, lst, n)
    out = torch.zeros(2, 16, Hd, dtype=torch.float32, device=dev())
    ops.moe_down(dn_p, act, wt, lst, n, E, Hd, I, 2, out)
    got = out.sum(0)
    ref = torch.zeros(16, Hd, device=dev())
    for e in range(E):
        if not active[e]:
            assert torch.isnan(act[e].float()).all()        # inactive experts were not touched
            continue
        gate, up = (x.float() @ gu[e].float().T).to(BF16).chunk(2, dim=-1)
        a = (torch.nn.functional.silu(gate.float()).to(BF16).float() * up.float()).to(BF16)
        ref += wt[:, e].float()[:, None] * (a.float() @ dn[e].float().T)
    assert (got - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert (got - ref).abs().mean() <= 3e-3 * ref.abs().max()


@pytest.mark.parametrize("pair_kernel", [False, True])
def test_expert_kernels_at_30b_a3b_shapes(pair_kernel):
    """BASELINE configs[4] at its REAL expert geometry (Qwen3-Coder-30B-A3B: H 2048, moe_intermediate 768, 128 experts,
    top-8): dfl_moe_route -> dfl_moe_gate_up / dfl_gemm_silu_mul_experts -> dfl_moe_down on 16 rows with a balanced
    routing spread (~80 active experts), against torch's router arithmetic and the fp32 experts loop
    (tf:models/qwen3_moe/modeling_qwen3_moe.py, Qwen3MoeExperts.forward)."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(30)
    E, I, Hd, rows, top_k = 128, 768, 2048, 16, 8
    gu = (torch.randn(E, 2 * I, Hd, generator=g) * 0.03).to(BF16).to(dev())
    dn = (torch.randn(E, Hd, I, generator=g) * 0.03).to(BF16).to(dev())
    x = torch.randn(16, Hd, generator=g).to(BF16).to(dev())
    logits = _tie_free_logits(16, E, g)
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 0, rows, 0)
    wt = torch.zeros(16, E, dtype=BF16, device=dev())
    active = torch.zeros(E, dtype=torch.int32, device=dev())
    lst = torch.zeros(E, dtype=torch.int32, device=dev())
    n = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.moe_route(logits.to(dev()), E, top_k, True, wt, active, lst, n, dyn=dyn, dyn_word=ops.DYN_BS)
    p = torch.softmax(logits.float(), dim=-1)
    v, i = torch.topk(p, top_k, dim=-1)
    v = (v / v.sum(dim=-1, keepdim=True)).to(BF16)
    wref = torch.zeros(16, E, dtype=BF16)
    wref[torch.arange(16)[:, None], i] = v
    assert torch.equal((wt != 0).cpu(), wref != 0)
    assert torch.allclose(wt.float().cpu(), wref.float(), rtol=2 ** -7, atol=1e-4)
    k = int(n)
    assert 60 <= k <= 100, k                       # a balanced spread, as a real router gives on 16 rows x top-8
    gu_p = torch.stack([ops.pack_weight_gateup(gu[e, :I].contiguous(), gu[e, I:].contiguous()) for e in range(E)])
    dn_p = torch.stack([ops.pack_weight(dn[e].contiguous()) for e in range(E)])
    xf = torch.empty(16 * Hd, dtype=BF16, device=dev())
    ops.pack_rows(x, 16, xf)
    act = torch.full((E, 16 * I), float("nan"), dtype=BF16, device=dev())
    if pair_kernel:
        ops.moe_gate_up(gu_p, xf, E, I, Hd, act, lst, n)
    else:
        ops.gemm_silu_mul_experts(gu_p, ops.rows_frag(xf), E, I, Hd, act, lst, n)
    out = torch.zeros(2, 16, Hd, dtype=torch.float32, device=dev())
    ops.moe_down(dn_p, act, wt, lst, n, E, Hd, I, 2, out)
    got = out.sum(0)
    ref = torch.zeros(16, Hd, device=dev())
    worst_act = 0.0
    for e in range(E):
        if not active[e]:
            assert torch.isnan(act[e].float()).all()
            continue
        gate, up = (x.float() @ gu[e].float().T).to(BF16).chunk(2, dim=-1)
        a = (torch.nn.functional.silu(gate.float()).to(BF16).float() * up.float()).to(BF16)
        ref += wt[:, e].float()[:, None] * (a.float() @ dn[e].float().T)
        # the expert's own activation tile, frag16 [I/8][16][8] -> rows
        got_a = act[e].view(I // 8, 16, 8).permute(1, 0, 2).reshape(16, I).float()
        worst_act = max(worst_act, float((got_a - a.float()).abs().max() / a.float().abs().max()))
    assert worst_act <= 2e-2, worst_act            # SiLU(gate) * up of every active expert (a bf16 ulp at the rounding points)
    H.assert_close(f"30B-A3B experts (pair={pair_kernel})", got, ref)


def test_native_moe_verify_at_30b_a3b_widths():
    """A 2-layer Qwen3MoeForCausalLM of Qwen3-Coder-30B-A3B's widths (H 2048, 32 q / 4 kv heads, 128 experts of 768,
    top-8) through NativeTarget.verify vs its own forward (call site model/dflash.py:249-255): routing agreement,
    logits, taps, K/V on the rows whose routing agrees, ids on margin-screened rows."""
    _moe_verify_vs_hf(_moe_hf(layers=2, E=128, top_k=8, Ie=768, hidden=2048, heads=32, kv=4, seed=43), (), tap_layers=(0,),
                      kv_layers=(0, 1), min_same=11, fp32_arbiter=True)


@pytest.mark.parametrize("mlp_only", [(), (1,)])
def test_native_moe_verify_matches_hf_forward(mlp_only):
    """A Qwen3MoeForCausalLM (16 experts, top-4; optionally a dense layer in between) through NativeTarget.verify vs
    its own forward: router decisions equal on (almost) every row, logits / taps / K/V of the rows whose routing
    agrees within the bf16 tolerance, posterior ids on margin-screened rows."""
    _moe_verify_vs_hf(_moe_hf(mlp_only=mlp_only), mlp_only, tap_layers=(0, 2), kv_layers=(0, 3), min_same=13)


def _moe_verify_vs_hf(hf, mlp_only, tap_layers, kv_layers, min_same, fp32_arbiter=False):
    """fp32_arbiter: with 8 experts per row the HF forward adds the expert outputs one by one INTO A bf16 TENSOR
    (Qwen3MoeExperts.forward: index_add_), this path sums them in fp32 and rounds once — the two bf16 results then differ
    by more than two roundings of the same arithmetic.  An fp32 forward of the same weights arbitrates: the native
    logits must be at least as close to it as HF's own bf16 logits are (and within 6e-2 of those)."""
    from transformers import DynamicCache
    from dflash_amd import NativeTarget
    cfg = hf.config
    L, E, top_k, Hd, V = cfg.num_hidden_layers, cfg.num_experts, cfg.num_experts_per_tok, cfg.hidden_size, cfg.vocab_size
    nt = NativeTarget(hf, prefill="hf")    # (the verify is the subject here; the native MoE prefill has its own tests below)
    assert nt.is_moe and all(("gu" in nt.layers[i]) == (i in mlp_only) for i in range(L))
    g = torch.Generator().manual_seed(7)
    P, bs = 50, 16
    prompt = torch.randint(0, 2000, (1, P), generator=g).to(dev())
    block = torch.randint(0, 2000, (1, bs), generator=g).to(dev())
    cache = nt.new_cache(128)
    nt.prefill(prompt, cache)
    logits = torch.zeros(32, V, dtype=BF16, device=dev())
    nt.debug_routing = []
    post, th = nt.verify(block[0], P, cache, tap_layers=list(tap_layers), logits_out=logits)
    routing = nt.debug_routing
    nt.debug_routing = None
    rc = DynamicCache()
    with torch.inference_mode():
        hf(prompt, past_key_values=rc, use_cache=True)
        ref = hf(block, position_ids=torch.arange(P, P + bs, device=dev())[None], past_key_values=rc, use_cache=True,
                 output_hidden_states=True, output_router_logits=True)
    moe_layers = [i for i in range(L) if i not in mlp_only]
    assert len(routing) == len(moe_layers) == len(ref.router_logits)
    same = torch.ones(bs, dtype=torch.bool)
    for (li, wt), rl in zip(routing, ref.router_logits):
        p = torch.softmax(rl.float(), dim=-1)
        idx = torch.topk(p, top_k, dim=-1).indices.cpu()
        want = torch.zeros(bs, E, dtype=torch.bool)
        want[torch.arange(bs)[:, None], idx] = True
        same &= ((wt[:bs].float().cpu() != 0) == want).all(dim=-1)
    assert int(same.sum()) >= min_same, same          # a near-tie at the k-th place may fall either way in bf16
    if fp32_arbiter:
        import copy
        hf32 = copy.deepcopy(hf).float()
        with torch.inference_mode():
            rc32 = DynamicCache()
            hf32(prompt, past_key_values=rc32, use_cache=True)
            ref32 = hf32(block, position_ids=torch.arange(P, P + bs, device=dev())[None], past_key_values=rc32,
                         use_cache=True, output_router_logits=True)
        for (li, wt), rl in zip(routing, ref32.router_logits):      # rows routed alike by all three
            idx = torch.topk(torch.softmax(rl.float(), dim=-1), top_k, dim=-1).indices.cpu()
            want = torch.zeros(bs, E, dtype=torch.bool)
            want[torch.arange(bs)[:, None], idx] = True
            same &= ((wt[:bs].float().cpu() != 0) == want).all(dim=-1)
        assert int(same.sum()) >= min_same - 3, same
        rows = same.nonzero()[:, 0].to(dev())
        r32 = ref32.logits[0][rows].float()
        e_nat = (logits[:bs][rows].float() - r32).abs()
        e_hf = (ref.logits[0][rows].float() - r32).abs()
        scale = float(r32.abs().max())
        print(f"[parity] vs fp32 forward: native max {float(e_nat.max()) / scale:.3e} mean {float(e_nat.mean()) / scale:.3e}; "
              f"HF bf16 max {float(e_hf.max()) / scale:.3e} mean {float(e_hf.mean()) / scale:.3e}")
        assert float(e_nat.mean()) <= 1.05 * float(e_hf.mean()) and float(e_nat.max()) <= 1.25 * float(e_hf.max())
        del hf32, rc32, ref32
    rows = same.nonzero()[:, 0].to(dev())
    H.assert_close("MoE verify logits", logits[:bs][rows], ref.logits[0][rows], max_rel=6e-2 if fp32_arbiter else H.MAX_REL)
    assert torch.equal(post[0], torch.argmax(logits[:bs], dim=-1))
    H.assert_ids_match_where_safe("MoE verify ids", post[0][rows], ref.logits[0][rows])
    for j, l in enumerate(tap_layers):
        H.assert_close(f"MoE verify tap {l}", th[:bs, j * Hd:(j + 1) * Hd][rows], ref.hidden_states[l + 1][0][rows])
    for li in kv_layers:
        H.assert_close(f"MoE verify K layer {li}", cache.k[li][:, :P + bs][:, torch.cat([torch.arange(P, device=dev()), P + rows])],
                       rc.layers[li].keys[0][:, torch.cat([torch.arange(P, device=dev()), P + rows])],
                       max_rel=4e-2 if fp32_arbiter else H.KV_MAX_REL)   # (behind a top-8 MoE layer: see the docstring)


def test_native_moe_target_end_to_end_lossless_walk():
    """dflash_generate / dflash_generate_policy on the NATIVE MoE verify with a large-margin greedy rule: committed ids
    are the target's closed-form greedy walk at block 16 and over a {8, 12, 16} schedule, scripted acceptance."""
    from dflash_amd import (DFlashDraftModel, EWMAPerformanceScheduler, NativeTarget, dflash_generate,
                            dflash_generate_policy)
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk
    cfg = H.tiny_cfg(num_target_layers=4, target_layer_ids=[0, 2])
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, dtype=BF16))
    hf = _moe_hf()
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf)
    prompt = torch.randint(0, 2000, (1, 30), generator=torch.Generator().manual_seed(4)).to(dev())
    n_new = 60
    G = greedy_walk(perm, prompt, n_new + 40).to(dev())
    plan = H.make_plan(64, 16, 17)

    def hook(blk, start, call):
        k = min(plan[call], blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            w = G[start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])

    r = dflash_generate(m, nt, prompt, cfg.mask_token_id, n_new, 16, None, 0.0, draft_token_hook=hook)
    assert r.output_ids[0].tolist() == G[:30 + n_new].tolist()
    assert max(r.acceptance_lengths) == 16
    sched = EWMAPerformanceScheduler(candidates=[8, 12, 16], scheduler_mode="ewma", warmup_cycles=3, ewma_alpha=0.25,
                                     switch_margin=0.03, required_streak=2, cooldown_cycles=2, probe_interval=5,
                                     low_accept_threshold=0.2, low_accept_streak=3, adl_rho=0.3, adl_delta=1.0,
                                     adl_k_min=8, adl_k_max=16, adl_neighborhood=4)
    rp = dflash_generate_policy(model=m, target=nt, input_ids=prompt, mask_token_id=cfg.mask_token_id,
                                max_new_tokens=n_new, stop_token_ids=None, temperature=0.0, scheduler=sched,
                                draft_token_hook=hook)
    assert rp.output_ids[0].tolist() == G[:30 + n_new].tolist()
    # blocks of 17..32 rows on the MoE target (two 16-row tiles through the per-tile MoE path), fixed and scheduled
    plan24 = H.make_plan(64, 24, 19)

    def hook24(blk, start, call):
        k = min(plan24[call], blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            blk[0, k + 1] = (G[start + k + 1] + 1) % 2000

    rw = dflash_generate(m, nt, prompt, cfg.mask_token_id, n_new, 24, None, 0.0, draft_token_hook=hook24)
    assert rw.output_ids[0].tolist() == G[:30 + n_new].tolist() and max(rw.acceptance_lengths) > 16
    sched2 = EWMAPerformanceScheduler(candidates=[12, 20, 24], scheduler_mode="ewma", warmup_cycles=3, ewma_alpha=0.25,
                                      switch_margin=0.03, required_streak=2, cooldown_cycles=2, probe_interval=5,
                                      low_accept_threshold=0.2, low_accept_streak=3, adl_rho=0.3, adl_delta=1.0,
                                      adl_k_min=12, adl_k_max=24, adl_neighborhood=4)
    rq = dflash_generate_policy(model=m, target=nt, input_ids=prompt, mask_token_id=cfg.mask_token_id,
                                max_new_tokens=n_new, stop_token_ids=None, temperature=0.0, scheduler=sched2,
                                draft_token_hook=hook24)
    assert rq.output_ids[0].tolist() == G[:30 + n_new].tolist() and max(rq.used_block_sizes) > 16


@pytest.mark.parametrize("block_size", [16, 24])
def test_moe_target_in_the_ragged_batch(block_size):
    """An MoE target under dflash_generate_batch (round 3: attention and dense projections of the group in one pass over
    the weights, the expert MLP per request tile): every request's committed ids are the target's greedy walk and equal the
    single-request loop's, acceptance lengths included.  block_size 24: two tiles per request (blocks of 17..32 rows)."""
    from dflash_amd import DFlashDraftModel, NativeTarget, dflash_generate
    from dflash_amd.batch import dflash_generate_batch
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk
    cfg = H.tiny_cfg(num_target_layers=4, target_layer_ids=[0, 2])
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, dtype=BF16))
    hf = _moe_hf(mlp_only=(1,))          # three MoE layers and a dense one in between
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf)
    lens, n_new = (30, 47, 12), 50
    prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(60 + i)).to(dev())
               for i, P in enumerate(lens)]
    Gs = [greedy_walk(perm, p, n_new + 40).to(dev()) for p in prompts]
    plans = [H.make_plan(64, block_size, 41 + i) for i in range(len(lens))]

    def hook_for(i):
        def hook(blk, start, call):
            k = min(plans[i][call], blk.shape[1] - 1)
            blk[0, 1:k + 1] = Gs[i][start + 1:start + k + 1]
            if k + 1 < blk.shape[1]:
                w = Gs[i][start + k + 1]
                blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])
        return hook

    hooks = [hook_for(i) for i in range(len(lens))]
    singles = [dflash_generate(m, nt, prompts[i], cfg.mask_token_id, n_new, block_size, None, 0.0, draft_token_hook=hooks[i])
               for i in range(len(lens))]

    def bhook(i, blk, start, call):
        hooks[i](blk[:, :min(block_size, lens[i] + n_new - start)], start, call)

    outs = dflash_generate_batch(m, nt, prompts, cfg.mask_token_id, n_new, block_size, None, 0.0, draft_token_hook=bhook)
    for i, (a, b) in enumerate(zip(singles, outs)):
        assert b.output_ids[0].tolist() == a.output_ids[0].tolist() == Gs[i][:lens[i] + n_new].tolist(), f"request {i}"
        assert b.acceptance_lengths == a.acceptance_lengths, f"request {i}"


# ----------------------------------------------------------------------------- prefill of an MoE target on the kernels
def _routed_alike(routing, router_logits, top_k, P):
    """Rows whose top-k expert SETS agree with the HF forward's in every MoE layer (a near-tie at the k-th place may fall
    either way between two bf16 evaluations of the same router)."""
    same = torch.ones(P, dtype=torch.bool)
    for (li, pe), rl in zip(routing, router_logits):
        ref = torch.topk(torch.softmax(rl.float().view(P, -1), dim=-1), top_k, dim=-1).indices.sort(dim=-1).values.cpu()
        same &= (pe.cpu().long().sort(dim=-1).values == ref).all(dim=-1)
    return same


def _moe_prefill_vs_hf(name, hf, P, taps, min_same, fp32_arbiter=False, seed=5):
    from transformers import DynamicCache
    from dflash_amd import NativeTarget
    cfg = hf.config
    L, top_k, V = cfg.num_hidden_layers, cfg.num_experts_per_tok, cfg.vocab_size
    nt = NativeTarget(hf)
    assert nt.is_moe and nt.native_prefill
    g = torch.Generator().manual_seed(seed)
    prompt = torch.randint(0, 2000, (1, P), generator=g).to(dev())
    cache = nt.new_cache(P + 64)
    nt.debug_routing = []
    out = nt.prefill(prompt, cache, output_hidden_states=True, tap_layers=taps)
    routing, nt.debug_routing = nt.debug_routing, None
    rc = DynamicCache()
    with torch.inference_mode():
        ref = hf(prompt, position_ids=torch.arange(P, device=dev())[None], past_key_values=rc, use_cache=True,
                 output_hidden_states=True, output_router_logits=True)
    assert len(routing) == len(ref.router_logits) == sum("gu_e" in lw for lw in nt.layers)
    same = _routed_alike(routing, ref.router_logits, top_k, P)
    ref_last = ref.logits[0, -1:]
    arb = None
    if fp32_arbiter:
        import copy
        hf32 = copy.deepcopy(hf).float()
        rc32 = DynamicCache()
        with torch.inference_mode():
            ref32 = hf32(prompt, position_ids=torch.arange(P, device=dev())[None], past_key_values=rc32, use_cache=True,
                         output_router_logits=True)
        same &= _routed_alike(routing, ref32.router_logits, top_k, P)
        arb = (ref32.logits[0, -1:].float(), rc32)
        del hf32
    assert int(same.sum()) >= min_same, (int(same.sum()), P)
    rows = same.nonzero()[:, 0].to(dev())
    assert out.logits.shape == (1, 1, V) and cache.get_seq_length() == P

    def closer(what, got, ref_bf16, ref_fp32):
        """The HF forward adds the expert outputs one by one into a bf16 tensor, this path sums them in fp32 and rounds
        once: two bf16 evaluations that differ by more than two roundings of the same arithmetic (most after the next
        layer's per-head k-norm).  The fp32 forward arbitrates: the native result is at least as close to it as HF's."""
        e_nat, e_hf = (got.float() - ref_fp32.float()).abs(), (ref_bf16.float() - ref_fp32.float()).abs()
        scale = float(ref_fp32.abs().max())
        print(f"[parity] {name} {what} vs fp32 forward: native max {float(e_nat.max()) / scale:.3e} mean "
              f"{float(e_nat.mean()) / scale:.3e}; HF bf16 max {float(e_hf.max()) / scale:.3e} mean {float(e_hf.mean()) / scale:.3e}")
        assert float(e_nat.mean()) <= 1.05 * float(e_hf.mean()) + 1e-6 * scale, what
        assert float(e_nat.max()) <= 1.25 * float(e_hf.max()) + 1e-6 * scale, what

    if arb is None:
        if bool(same[P - 1]):
            H.assert_close(f"{name} prefill logits (last row)", out.logits[0], ref_last)
        for l in taps:
            H.assert_close(f"{name} prefill tap layer {l}", out.hidden_states[l + 1][0][rows], ref.hidden_states[l + 1][0][rows])
        for li in (0, L - 1):
            H.assert_close(f"{name} prefill K layer {li}", cache.k[li][:, :P][:, rows], rc.layers[li].keys[0][:, rows],
                           max_rel=H.KV_MAX_REL)
            H.assert_close(f"{name} prefill V layer {li}", cache.v[li][:, :P][:, rows], rc.layers[li].values[0][:, rows],
                           max_rel=H.KV_MAX_REL)
    else:
        r32, rc32 = arb
        assert bool(same[P - 1]), "pick a seed whose last row routes alike"
        closer("logits (last row)", out.logits[0], ref_last, r32)
        H.assert_close(f"{name} prefill logits (last row)", out.logits[0], ref_last, max_rel=6e-2)
        for li in (0, L - 1):
            closer(f"K layer {li}", cache.k[li][:, :P][:, rows], rc.layers[li].keys[0][:, rows], rc32.layers[li].keys[0][:, rows])
            closer(f"V layer {li}", cache.v[li][:, :P][:, rows], rc.layers[li].values[0][:, rows], rc32.layers[li].values[0][:, rows])
            H.assert_close(f"{name} prefill V layer {li}", cache.v[li][:, :P][:, rows], rc.layers[li].values[0][:, rows],
                           max_rel=8e-2)   # (vs HF's own bf16 result; the bar that matters is the arbiter's, above)
        for l in taps:
            H.assert_close(f"{name} prefill tap layer {l}", out.hidden_states[l + 1][0][rows], ref.hidden_states[l + 1][0][rows],
                           max_rel=6e-2)
    return nt, prompt, cache, same


@pytest.mark.parametrize("P,mlp_only", [(45, ()), (300, ()), (130, (1,))])
def test_native_moe_prefill_matches_hf_forward(P, mlp_only):
    """Prompt rows of a Qwen3MoeForCausalLM (16 experts, top-4; optionally a dense layer in between) through the native
    prefill (rows sorted by expert, grouped MFMA GEMMs) vs the HF forward: last-row logits, taps, K/V on the rows that
    both route alike (model/dflash.py:218-225 on an MoE target)."""
    _moe_prefill_vs_hf(f"MoE P={P}", _moe_hf(mlp_only=mlp_only), P, taps=[0, 2], min_same=int(0.85 * P))


def test_native_moe_prefill_at_30b_a3b_widths():
    """Two layers of Qwen3-Coder-30B-A3B's widths (H 2048, 32 q / 4 kv heads, 128 experts of 768, top-8), 256 prompt rows."""
    _moe_prefill_vs_hf("30B-A3B widths", _moe_hf(layers=2, E=128, top_k=8, Ie=768, hidden=2048, heads=32, kv=4, seed=43), 256,
                       taps=[0], min_same=160, fp32_arbiter=True)


@pytest.mark.parametrize("rows_per_item,P,skew", [(64, 70, False), (128, 70, False), (64, 300, True), (128, 300, True)])
def test_moe_prefill_pieces_are_exact_on_integers(rows_per_item, P, skew):
    """dfl_prefill_moe_*: with small-integer weights and activations every product and sum is exact, so the grouped
    GEMMs, the gather and the combine must reproduce a torch evaluation of the same routed sum bit for bit."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(3)
    Hd, Ie, E, k = 128, 64, 5, 2     # skew: expert 0 serves every row (19 tiles: several work items of one expert)
    x = torch.randint(-2, 3, (P, Hd), generator=g).float()
    gate = torch.randint(-1, 2, (E, Ie, Hd), generator=g).float()
    up = torch.randint(-1, 2, (E, Ie, Hd), generator=g).float()
    down = torch.randint(-1, 2, (E, Hd, Ie), generator=g).float()
    logits = _tie_free_logits(P, E, g)
    if skew:
        logits[:, 0] = 20.0
    h0 = torch.randint(-3, 4, (P, Hd), generator=g).float()
    d = dev()
    Pp = ops.prefill_rows_padded(P)
    xr = torch.zeros(Pp, Hd, dtype=BF16, device=d)
    xr[:P] = x.to(BF16)
    xf = torch.zeros(Pp * Hd, dtype=BF16, device=d)
    ops.prefill_norm_pack(xr, P, Hd, None, 1e-6, xf)
    gu_e = torch.stack([ops.pack_weight_gateup(gate[e].to(BF16).to(d), up[e].to(BF16).to(d)) for e in range(E)])
    down_e = torch.stack([ops.pack_weight(down[e].to(BF16).to(d)) for e in range(E)])
    sc = ops.prefill_moe_scratch(P, Hd, Ie, E, k, 128, d)
    sc["rows_per_item"] = rows_per_item
    tpi = rows_per_item // 16
    # the router GEMM is skipped: the logits are given (rows of the scratch buffer), the rest of the chain as in the product
    L, st = ops.lib(), 0
    sc["rlog"][:P, :E] = logits.to(d)
    ops.prefill_moe_route(sc["rlog"], P, sc, True)
    prob = torch.softmax(logits.float(), dim=-1)
    tv, ti = torch.topk(prob, k, dim=-1)
    wref = (tv / tv.sum(-1, keepdim=True)).to(BF16).float()
    assert torch.equal(sc["pair_e"][:, :k].cpu().long(), ti)
    assert torch.allclose(sc["pair_w"][:, :k].cpu(), wref, atol=8e-3)     # (division in fp32 here and there: a bf16 ulp)
    cnt = torch.bincount(ti.flatten(), minlength=E)
    assert torch.equal(sc["cnt"].cpu().long(), cnt)
    n_items, n_tiles = sc["n_items"].tolist()
    assert n_tiles == int(((cnt + 15) // 16).sum()) and n_items == int(((((cnt + 15) // 16) + tpi - 1) // tpi).sum())
    h = torch.zeros(Pp, Hd, dtype=BF16, device=d)
    h[:P] = h0.to(BF16)
    # the remaining calls through the product wrapper would recompute the router GEMM: call the pieces
    ops.check(L.dfl_prefill_moe_gather(xf.data_ptr(), P, Hd, k, E, sc["src_row"].data_ptr(), sc["n_items"].data_ptr(),
                                       sc["xg"].data_ptr(), st), "gather")
    # (gathered copy xg as the operand — the product gathers in the kernel's LDS-DMA addresses instead: second call below)
    ops.check(L.dfl_prefill_moe_gemm_silu(gu_e.data_ptr(), gu_e.stride(0), sc["xg"].data_ptr(), sc["items"].data_ptr(),
                                          sc["n_items"].data_ptr(), sc["max_items"], Ie, Hd, sc["act_g"].data_ptr(), rows_per_item,
                                          None, None, st), "silu")
    act_from_copy = sc["act_g"][:n_tiles * 16 * Ie].clone()
    sc["act_g"].zero_()
    ops.check(L.dfl_prefill_moe_gemm_silu(gu_e.data_ptr(), gu_e.stride(0), xf.data_ptr(), sc["items"].data_ptr(),
                                          sc["n_items"].data_ptr(), sc["max_items"], Ie, Hd, sc["act_g"].data_ptr(), rows_per_item,
                                          sc["src_row"].data_ptr(), sc["zeros"].data_ptr(), st), "silu (gather in the addresses)")
    assert torch.equal(sc["act_g"][:n_tiles * 16 * Ie], act_from_copy)
    ops.check(L.dfl_prefill_moe_gemm_down(down_e.data_ptr(), down_e.stride(0), sc["act_g"].data_ptr(), sc["items"].data_ptr(),
                                          sc["n_items"].data_ptr(), sc["max_items"], Hd, Ie, sc["row_w"].data_ptr(),
                                          sc["out32"].data_ptr(), rows_per_item, st), "down")
    pw = sc["pair_w"][:, :k].cpu()
    # every pair's gathered slot holds w * down_e(act_e(x)) of ITS row and expert, in fp32 (act in bf16 roundings as the kernel's)
    pos = sc["posmap"][:, :k].cpu().long()
    assert pos.flatten().unique().numel() == P * k
    srow = sc["src_row"][:n_tiles * 16].cpu().long()
    assert torch.equal(srow[pos.flatten()], torch.arange(P).repeat_interleave(k)) and int((srow >= 0).sum()) == P * k
    out32 = sc["out32"].cpu()
    total = torch.zeros(P, Hd)
    for m in range(P):
        for r in range(k):
            e = int(ti[m, r])
            gb, ub = (gate[e] @ x[m]).to(BF16).float(), (up[e] @ x[m]).to(BF16).float()
            act = ((gb * torch.sigmoid(gb)).to(BF16).float() * ub).to(BF16).float()
            want = pw[m, r] * (down[e] @ act)
            got = out32[pos[m, r]]
            assert torch.allclose(got, want, rtol=2e-2, atol=2e-2 * float(want.abs().max() + 1)), (m, r)   # (silu: __expf vs torch)
            total[m] += got
    ops.check(L.dfl_prefill_moe_combine(sc["out32"].data_ptr(), sc["posmap"].data_ptr(), P, Hd, k, h.data_ptr(), h.stride(0),
                                        None, 0, None, st), "combine")
    want_h = (h0.to(BF16).float() + total.to(BF16).float()).to(BF16)
    assert torch.equal(h[:P].cpu(), want_h)
    assert int(h[P:].abs().sum()) == 0


def test_moe_target_one_copy_keep_hf_false():
    """NativeTarget(keep_hf=False) on an MoE target: prefill + verify with the wrapped model gone give what the two-copy
    target gives (same kernels, same weights)."""
    from dflash_amd import NativeTarget
    hf = _moe_hf()
    a, b = NativeTarget(hf), NativeTarget(hf, keep_hf=False)
    assert b.hf is None and b.native_prefill
    g = torch.Generator().manual_seed(9)
    P = 77
    prompt = torch.randint(0, 2000, (1, P), generator=g).to(dev())
    block = torch.randint(0, 2000, (16,), generator=g).to(dev())
    ca, cb = a.new_cache(P + 64), b.new_cache(P + 64)
    oa, ob = a.prefill(prompt, ca), b.prefill(prompt, cb)
    assert torch.equal(oa.logits, ob.logits) and torch.equal(ca.k[:, :, :P], cb.k[:, :, :P])
    pa, _ = a.verify(block, P, ca)
    pb, _ = b.verify(block, P, cb)
    assert torch.equal(pa, pb)
