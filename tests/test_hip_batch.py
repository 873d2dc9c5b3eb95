"""GPU parity of the ragged multi-request batch (dflash_amd.batch, dfl_*_batch kernels):
every request of a batch must come out exactly as the single-request loop produces it —
the reference's contract for several prompts is a loop over them (benchmark.py:445-470,
benchmark_batched.py:212-243)."""
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def dev():
    return torch.device("cuda", 0)


def _setup(layers=6, seed=5):
    from dflash_amd import DFlashDraftModel, NativeTarget
    from dflash_amd.synthetic import impose_greedy_walk, make_hf_qwen3
    cfg = H.tiny_cfg()
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, seed=3, dtype=BF16))
    torch.manual_seed(11)
    hf = make_hf_qwen3({**H.TINY_TARGET, "num_layers": layers}, dev(), dtype=BF16)
    perm = impose_greedy_walk(hf, seed=seed)
    return cfg, m, hf, NativeTarget(hf), perm


def _hook_for(G, plan, vocab=2000):
    def hook(blk, start, call):
        k = min(plan[call], blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            w = G[start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % vocab, blk[0, k + 1])
    return hook


@pytest.mark.parametrize("lens", [(33, 7, 50, 21), (40, 18), (25,), (20, 31, 64)])
def test_batch_matches_single_request_loop(lens):
    """Ragged prompts, different acceptance plans per request, tail clamp, requests that
    finish at different cycles: ids and acceptance lengths equal the single-request run."""
    from dflash_amd import dflash_generate
    from dflash_amd.batch import dflash_generate_batch
    from dflash_amd.synthetic import greedy_walk
    cfg, m, hf, nt, perm = _setup()
    n_new = 70
    prompts, Gs, plans = [], [], []
    for i, P in enumerate(lens):
        p = torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(40 + i)).to(dev())
        prompts.append(p)
        Gs.append(greedy_walk(perm, p, n_new + 40).to(dev()))
        plans.append(H.make_plan(64, 16, 17 + i))
    hooks = [_hook_for(Gs[i], plans[i]) for i in range(len(lens))]
    singles = [dflash_generate(m, nt, prompts[i], cfg.mask_token_id, n_new, 16, None, 0.0, draft_token_hook=hooks[i])
               for i in range(len(lens))]

    def bhook(i, blk, start, call):
        # the batched block always has 16 slots; the single loop's tail block is shorter
        hooks[i](blk[:, :min(16, lens[i] + n_new - start)], start, call)

    outs = dflash_generate_batch(m, nt, prompts, cfg.mask_token_id, n_new, 16, None, 0.0, draft_token_hook=bhook)
    for i, (a, b) in enumerate(zip(singles, outs)):
        assert b.output_ids[0].tolist() == a.output_ids[0].tolist() == Gs[i][:lens[i] + n_new].tolist(), f"request {i}"
        assert b.acceptance_lengths == a.acceptance_lengths, f"request {i}"
        assert b.num_output_tokens == a.num_output_tokens == n_new


def test_batch_stop_token_parks_one_request():
    """A stop token committed by one request ends it (model/dflash.py:265-275) while the
    others keep decoding; its output is cut after the stop token like the single loop's."""
    from dflash_amd import dflash_generate
    from dflash_amd.batch import dflash_generate_batch
    from dflash_amd.synthetic import greedy_walk
    cfg, m, hf, nt, perm = _setup()
    lens, n_new = (30, 44, 19), 60
    prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(70 + i)).to(dev())
               for i, P in enumerate(lens)]
    Gs = [greedy_walk(perm, p, n_new + 40).to(dev()) for p in prompts]
    stop = [int(Gs[1][44 + 23])]                      # request 1 meets it after 24 new tokens
    plans = [H.make_plan(64, 16, 90 + i) for i in range(3)]
    hooks = [_hook_for(Gs[i], plans[i]) for i in range(3)]
    singles = [dflash_generate(m, nt, prompts[i], cfg.mask_token_id, n_new, 16, stop, 0.0, draft_token_hook=hooks[i])
               for i in range(3)]
    outs = dflash_generate_batch(m, nt, prompts, cfg.mask_token_id, n_new, 16, stop, 0.0,
                                 draft_token_hook=lambda i, blk, s, c: hooks[i](blk[:, :min(16, lens[i] + n_new - s)], s, c))
    for i in range(3):
        assert outs[i].output_ids[0].tolist() == singles[i].output_ids[0].tolist(), f"request {i}"
        assert outs[i].acceptance_lengths == singles[i].acceptance_lengths
    assert outs[1].output_ids[0, -1].item() == stop[0] or stop[0] in Gs[1][:44].tolist()


def test_batch_temperature_path_with_sharp_logits():
    """T = 0.7 in the batched loop (BASELINE config 4's sampling rule: a draft token is accepted
    iff it equals the token SAMPLED from the target, model/utils.py:30-34): the walk target's
    logit margin (~70) makes the draw deterministic, so ids and acceptance lengths equal T = 0."""
    from dflash_amd.batch import dflash_generate_batch
    from dflash_amd.synthetic import greedy_walk
    cfg, m, hf, nt, perm = _setup()
    lens, n_new = (30, 17), 40
    prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(60 + i)).to(dev())
               for i, P in enumerate(lens)]
    Gs = [greedy_walk(perm, p, n_new + 40).to(dev()) for p in prompts]
    plans = [H.make_plan(64, 16, 33 + i) for i in range(2)]
    hooks = [_hook_for(Gs[i], plans[i]) for i in range(2)]
    bh = lambda i, blk, s, c: hooks[i](blk[:, :min(16, lens[i] + n_new - s)], s, c)  # noqa: E731
    cold = dflash_generate_batch(m, nt, prompts, cfg.mask_token_id, n_new, 16, None, 0.0, draft_token_hook=bh)
    torch.manual_seed(0)
    warm = dflash_generate_batch(m, nt, prompts, cfg.mask_token_id, n_new, 16, None, 0.7, draft_token_hook=bh)
    for i in range(2):
        assert warm[i].output_ids[0].tolist() == cold[i].output_ids[0].tolist() == Gs[i][:lens[i] + n_new].tolist()
        assert warm[i].acceptance_lengths == cold[i].acceptance_lengths


def test_batched_draft_and_verify_match_single_kernels():
    """Three cycles, 3 requests with different prefix lengths, interleaved with three
    single-request sessions on the same target: committed ids and tau exact; draft ids,
    taps and the K/V both paths appended within the bf16 tolerance of DESIGN.md §2."""
    from dflash_amd.batch import BatchedDecoder
    from dflash_amd.generate import DecodeSession
    cfg, m, hf, nt, perm = _setup()
    lens = (45, 23, 70)
    prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(7 + i)).to(dev())
               for i, P in enumerate(lens)]
    dec = BatchedDecoder(m, nt, 3, max_rows=200, out_len=200, mask_token_id=cfg.mask_token_id)
    sess = []
    for r, p in enumerate(prompts):
        dec.admit(r, p)
        s = DecodeSession(m, nt, p, mask_token_id=cfg.mask_token_id, max_new_tokens=100, max_block_size=16,
                          stop_token_ids=None, temperature=0.0)
        s.prefill()
        sess.append(s)
    for cyc in range(3):
        dec.draft()
        blocks = dec.block.clone()
        dec.verify()
        taps = dec.d["taps"].clone()
        res = dec.accept()
        for r, s in enumerate(sess):
            start = s.start
            got = {}   # the drafted block (after the cycle s.block already holds the NEXT block, re-armed by the accept)
            out = s.cycle(16, after_draft=lambda b: got.update(blk=b[0].clone()))
            assert out.tau == res[r][0], (cyc, r)
            # draft ids: random draft weights leave near-ties that the two summation orders may
            # break differently; the block's first token (committed) must agree, most others do
            assert got["blk"][0] == blocks[r, 0]
            assert int((got["blk"] == blocks[r]).sum()) >= 12, (cyc, r, got["blk"].tolist(), blocks[r].tolist())
            # the re-armed next block: the bonus token, then mask ids (model/dflash.py:235)
            assert s.block[0, 0] == s.output_ids[0, s.start] and (s.block[0, 1:] == cfg.mask_token_id).all()
            assert torch.equal(s.output_ids[0, :s.start + 1], dec.output_ids[r, :s.start + 1])
            H.assert_close(f"batch vs single taps c{cyc} r{r}", taps[r, :out.tau], s.target_hidden[0])
            for li in (0, cfg.num_hidden_layers - 1):
                H.assert_close(f"batch vs single draft K l{li} c{cyc} r{r}", dec.dk[r, li][:, :start],
                               s.dcache.k[li][:, :start], max_rel=H.KV_MAX_REL)
                H.assert_close(f"batch vs single draft V l{li} c{cyc} r{r}", dec.dv[r, li][:, :start],
                               s.dcache.v[li][:, :start], max_rel=H.KV_MAX_REL)
            for li in (0, nt.L - 1):
                H.assert_close(f"batch vs single target K l{li} c{cyc} r{r}", dec.tk[r, li][:, :s.start],
                               s.tcache.k[li][:, :s.start], max_rel=H.KV_MAX_REL)


# ------------------------------------------------------------------ kernel level
def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _unfrag(frag, K):
    return frag.view(K // 8, 16, 8).permute(1, 0, 2).reshape(16, K)


def _frag_of(x):  # x [MT, 16, K] bf16 cuda -> [MT, 16*K] frag16
    from dflash_amd import ops
    MT, _, K = x.shape
    out = torch.empty(MT, 16 * K, dtype=BF16, device=x.device)
    for r in range(MT):
        ops.pack_rows(x[r], 16, out[r])
    return out


def _dyn(vals, MT):
    d = torch.zeros(MT, 8, dtype=torch.int32)
    for r, (tau, bs) in enumerate(vals):
        d[r, 1], d[r, 2] = tau, bs
    return d.to(dev())


@pytest.mark.parametrize("R,N,K", [(1, 64, 512), (2, 48, 4096), (3, 256, 4096), (4, 4096, 4096), (4, 512, 12288),
                                   (4, 256, 20480), (3, 128, 9728), (4, 64, 2048 + 32)])
def test_gemm_resid_batch_exact_small_ints(R, N, K):
    """Small-integer operands: products and sums are exact in fp32 whatever the order, so
    the batched GEMM (K parts through partial tiles, last-arriver finish, residual add,
    taps, sums of squares, per-request valid rows) must match torch bit for bit."""
    from dflash_amd import ops
    MT = ops.batch_tiles(R)
    g = _gen(R * 1000 + N + K)
    w = torch.randint(-2, 3, (N, K), generator=g).to(BF16).to(dev())
    x = torch.randint(-2, 3, (MT, 16, K), generator=g).to(BF16).to(dev())
    h0 = torch.randint(-8, 9, (MT, 16, N), generator=g).to(BF16).to(dev())
    rows = [(0, [16, 5, 1, 9][r] if r < R else 0) for r in range(MT)]
    dyn = _dyn(rows, MT)
    wp = ops.pack_weight(w)
    ws = ops.gemm_batch_ws(N, K, dev())
    for mode in ("frag", "rows"):
        h = h0.clone()
        tap = torch.zeros(MT, 16, N + 32, dtype=BF16, device=dev())
        ss = torch.zeros(MT, N, dtype=torch.float32, device=dev())
        src = ops.brows_frag(_frag_of(x)) if mode == "frag" else ops.brows_plain(x, ops.DYN_BS)
        for _ in range(2):   # twice: the tickets must be back at zero after a launch
            h.copy_(h0)
            ops.gemm_resid_batch(wp, src, R, N, K, h, add_residual=True, ws=ws, dyn=dyn, ss_out=ss,
                                 tap=tap[:, :, 16:16 + N])
        for r in range(R):
            nv = 16 if mode == "frag" else rows[r][1]
            xe = x[r].float().clone()
            xe[nv:] = 0
            want = (h0[r].float() + (xe @ w.float().T).to(BF16).float()).to(BF16)
            assert torch.equal(h[r], want), (mode, r)
            assert torch.equal(tap[r, :, 16:16 + N], want)
            sse = want.float().pow(2).view(16, N // 16, 16).sum(-1).T          # [tile][row]
            assert torch.allclose(ss[r].view(N // 16, 16), sse, rtol=1e-6, atol=0)


@pytest.mark.parametrize("R,I,K", [(2, 64, 512), (4, 1024, 4096), (3, 12288 // 8, 4096)])
def test_gemm_silu_mul_batch(R, I, K):
    from dflash_amd import ops
    MT = ops.batch_tiles(R)
    g = _gen(I + K + R)
    gate = (torch.randn(I, K, generator=g) * 0.05).to(BF16).to(dev())
    up = (torch.randn(I, K, generator=g) * 0.05).to(BF16).to(dev())
    x = torch.randn(MT, 16, K, generator=g).to(BF16).to(dev())
    dyn = _dyn([(0, 16)] * MT, MT)
    act = torch.zeros(MT, 16 * I, dtype=BF16, device=dev())
    act1 = torch.zeros(16 * I, dtype=BF16, device=dev())
    wp = ops.pack_weight_gateup(gate, up)
    ws = ops.gemm_batch_ws(2 * I, K, dev())
    ops.gemm_silu_mul_batch(wp, ops.brows_plain(x, ops.DYN_BS), R, I, K, act, ws, dyn)
    for r in range(R):
        gb = (x[r].float() @ gate.float().T).to(BF16).float()
        ub = (x[r].float() @ up.float().T).to(BF16).float()
        want = (torch.nn.functional.silu(gb).to(BF16).float() * ub).to(BF16).float()
        got = _unfrag(act[r], I).float()
        d = (got - want).abs()
        assert d.max() <= 2e-2 * want.abs().max() and d.mean() <= 2e-3 * want.abs().max()
        # and against the single-request kernel (same rounding points, other summation order)
        ops.gemm_silu_mul(wp, ops.rows_plain(x[r]), I, K, act1)
        d1 = (got - _unfrag(act1, I).float()).abs()
        assert d1.max() <= 2e-2 * want.abs().max()


@pytest.mark.parametrize("R,V,K,row0", [(2, 2048, 512, 1), (4, 4096 + 16 * 7, 4096, 0), (3, 151936, 4096, 1)])
def test_gemm_argmax_batch(R, V, K, row0):
    """Fused lm_head + argmax for R requests with different row counts: ids equal
    torch.argmax of the bf16 logits the same kernel materialises, and the logits equal
    the fp32 reference within bf16 rounding."""
    from dflash_amd import ops
    MT = ops.batch_tiles(R)
    g = _gen(V + K)
    w = (torch.randn(V, K, generator=g) * 0.05).to(BF16).to(dev())
    x = torch.randn(MT, 16, K, generator=g).to(BF16).to(dev())
    bss = [16, 12, 3, 7][:R] + [0] * (MT - R)
    dyn = _dyn([(0, b) for b in bss], MT)
    wp = ops.pack_weight(w)
    ws = ops.gemm_batch_ws(V, K, dev())
    ids = torch.full((MT, 16), -1, dtype=torch.int64, device=dev())
    logits = torch.zeros(MT, 16, V, dtype=BF16, device=dev())
    ops.gemm_argmax_batch(wp, ops.brows_plain(x, ops.DYN_BS), R, V, K, row0, 16 - row0, ws, ids, row0, dyn,
                          nrows_dyn_word=ops.DYN_BS, logits=logits)
    for r in range(R):
        n = bss[r] - row0
        ref = (x[r].float() @ w.float().T)[row0:bss[r]]
        lg = logits[r, row0:bss[r]]
        assert (lg.float() - ref).abs().max() <= 2e-2 * ref.abs().max()
        assert torch.equal(ids[r, row0:row0 + n], torch.argmax(lg, dim=-1))
        assert torch.all(ids[r, row0 + n:] == -1) and torch.all(ids[r, :row0] == -1)


def test_norm_frag_then_gemm_f32_batch():
    """dfl_norm_frag_batch (RMSNorm -> frag16, per-request valid rows) feeding the fp32 partial
    GEMM over the K parts: sum of the parts equals the reference GEMM of the normalised rows;
    the batched GEMMs reject the in-GEMM norm source (mode 2)."""
    from dflash_amd import ops
    from dflash_amd._lib import DFlashHipError
    R, N, K = 3, 6144, 4096
    MT = ops.batch_tiles(R)
    g = _gen(5)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16).to(dev())
    h = torch.randn(MT, 16, K, generator=g).to(BF16).to(dev())
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF16).to(dev())
    dyn = _dyn([(0, 16), (0, 9), (0, 2), (0, 0)], MT)
    xn = torch.full((MT, 16 * K), 7.0, dtype=BF16, device=dev())
    ops.norm_frag_batch(h, R, nw, 1e-6, xn, dyn, ops.DYN_BS)
    ks = ops.batch_ksplit(K)
    out = torch.zeros(ks * MT * 16 * N, dtype=torch.float32, device=dev())
    wp = ops.pack_weight(w)
    ops.gemm_f32_batch(wp, ops.brows_frag(xn), R, N, K, out, dyn)
    got = out.view(ks, MT, 16, N).sum(0)
    for r, nv in enumerate([16, 9, 2]):
        hf = h[r].float()
        xr = (nw.float() * (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF16).float()).to(BF16)
        xr[nv:] = 0
        assert torch.equal(_unfrag(xn[r], K), xr), r          # same rounding points as Qwen3RMSNorm
        want = xr.float() @ w.float().T
        d = (got[r] - want).abs()
        assert d.max() <= 1e-2 * want.abs().max(), r
        assert torch.count_nonzero(got[r, nv:]) == 0
    ss = torch.ones(MT, K, device=dev())
    with pytest.raises(DFlashHipError):
        ops.gemm_f32_batch(wp, ops.brows_normed(h, ss, K // 16, nw, 1e-6, ops.DYN_BS), R, N, K, out, dyn)


def test_graph_replay_matches_eager_cycles():
    """The captured hipGraphs (draft body / lm_head / verify + accept) advance the requests
    exactly like the kernel-by-kernel launches: same committed ids, same tau per cycle."""
    from dflash_amd.batch import BatchedDecoder
    from dflash_amd.synthetic import greedy_walk
    cfg, m, hf, nt, perm = _setup()
    lens = (33, 50, 21)
    prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(40 + i)).to(dev())
               for i, P in enumerate(lens)]
    Gs = [greedy_walk(perm, p, 300).to(dev()) for p in prompts]
    plans = [H.make_plan(64, 16, 17 + i) for i in range(3)]
    hooks = [_hook_for(Gs[i], plans[i]) for i in range(3)]
    runs = []
    for use_graph in (False, True):
        dec = BatchedDecoder(m, nt, 3, max_rows=400, out_len=400, mask_token_id=cfg.mask_token_id)
        for r, p in enumerate(prompts):
            dec.admit(r, p)
        taus = [dec.cycle(lambda r, b, s, c: hooks[r](b, s, c))]
        if use_graph:
            dec.capture()
        for _ in range(12):
            step = dec.cycle_graph if use_graph else dec.cycle
            taus.append(step(lambda r, b, s, c: hooks[r](b, s, c)))
        runs.append((taus, dec.output_ids.clone(), list(dec.start)))
    assert runs[0][0] == runs[1][0]
    assert runs[0][2] == runs[1][2]
    assert torch.equal(runs[0][1], runs[1][1])
    for r in range(3):
        n = runs[1][2][r]
        assert torch.equal(runs[1][1][r, :n + 1], Gs[r][:n + 1])


def test_full_size_batch_matches_single_request_path():
    """BASELINE.json's full shapes (Qwen3-8B-DFlash-b16 draft: H 4096, 5 layers, 32/8 heads,
    FFN 12288, 5 taps of 4096, vocabulary 151936; Qwen3-8B-shaped target cut to 6 layers): four
    ragged requests through the batched launches — K parts 2 / 6 / 10, slabs, tickets, the
    launch-boundary residual reduce — against four single-request sessions over two cycles:
    same committed ids and tau, draft ids agreeing where the margins allow, taps and the K/V
    rows both paths appended within the bf16 tolerance of DESIGN.md §2."""
    from dflash_amd import DFlashConfig, DFlashDraftModel, NativeTarget
    from dflash_amd.batch import BatchedDecoder
    from dflash_amd.config import QWEN3_8B_DRAFT, QWEN3_8B_TARGET
    from dflash_amd.generate import DecodeSession
    from dflash_amd.synthetic import impose_greedy_walk, make_hf_qwen3
    L = 6
    cfg = DFlashConfig(**{**QWEN3_8B_DRAFT, "num_target_layers": L, "target_layer_ids": [0, 1, 2, 3, 4]})
    m = DFlashDraftModel(cfg, device=dev())
    g = torch.Generator(device=dev()).manual_seed(0)
    m.load_state_dict({k: (torch.randn(s, generator=g, device=dev(), dtype=torch.float32) * 0.02).to(BF16)
                       if len(s) == 2 else (1 + 0.1 * torch.randn(s, generator=g, device=dev())).to(BF16)
                       for k, s in cfg.state_dict_shapes().items()})
    torch.manual_seed(0)
    hf = make_hf_qwen3({**QWEN3_8B_TARGET, "num_layers": L}, dev())
    impose_greedy_walk(hf, seed=9)
    nt = NativeTarget(hf)
    lens = (70, 33, 129, 48)
    prompts = [torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(21 + i)).to(dev())
               for i, P in enumerate(lens)]
    dec = BatchedDecoder(m, nt, 4, max_rows=256, out_len=256, mask_token_id=cfg.mask_token_id)
    sess = []
    for r, p in enumerate(prompts):
        dec.admit(r, p)
        s = DecodeSession(m, nt, p, mask_token_id=cfg.mask_token_id, max_new_tokens=64, max_block_size=16,
                          stop_token_ids=None, temperature=0.0)
        s.prefill()
        sess.append(s)
    agree = []
    for cyc in range(2):
        dec.draft()
        blocks = dec.block.clone()
        dec.verify()
        taps = dec.d["taps"].clone()
        res = dec.accept()
        for r, s in enumerate(sess):
            start = s.start
            got = {}   # the drafted block (s.block holds the re-armed NEXT block once the cycle is over)
            out = s.cycle(16, after_draft=lambda b: got.update(blk=b[0].clone()))
            assert out.tau == res[r][0], (cyc, r)
            assert torch.equal(s.output_ids[0, :s.start + 1], dec.output_ids[r, :s.start + 1])
            agree.append(float((got["blk"] == blocks[r]).float().mean()))
            H.assert_close(f"8B batch vs single taps c{cyc} r{r}", taps[r, :out.tau], s.target_hidden[0])
            for nm, a, b in (("draft K l4", dec.dk[r, 4][:, :start], s.dcache.k[4][:, :start]),
                             ("draft V l0", dec.dv[r, 0][:, :start], s.dcache.v[0][:, :start]),
                             ("target K last", dec.tk[r, L - 1][:, :s.start], s.tcache.k[L - 1][:, :s.start])):
                H.assert_close(f"8B batch vs single {nm} c{cyc} r{r}", a, b, max_rel=H.KV_MAX_REL)
    # random-weight draft logits are near-tied (margins ~ bf16 ulp): most, not all, tokens agree
    assert sum(agree) / len(agree) >= 0.7, agree


def test_batch_with_llama_style_target():
    """BASELINE config 4's target family in the batched loop: a Llama-layout target (no per-head
    q/k norm, "llama3" RoPE frequency scaling active at these positions) behind NativeTarget;
    every request equals its single-request run."""
    tf = pytest.importorskip("transformers")
    from dflash_amd import DFlashDraftModel, NativeTarget, dflash_generate
    from dflash_amd.batch import dflash_generate_batch
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk
    cfg = H.tiny_cfg(num_target_layers=4)
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, seed=3, dtype=BF16))
    lc = tf.LlamaConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                        num_attention_heads=4, num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5,
                        max_position_embeddings=4096, tie_word_embeddings=False, attention_bias=False, mlp_bias=False,
                        rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0,
                                         "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                         "original_max_position_embeddings": 32})
    lc._attn_implementation = "sdpa"
    torch.manual_seed(3)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(BF16)
    try:
        with torch.device(dev()):
            hf = tf.LlamaForCausalLM(lc).eval()
    finally:
        torch.set_default_dtype(prev)
    perm = impose_greedy_walk(hf, seed=6)
    nt = NativeTarget(hf)
    lens, n_new = (26, 41, 9), 48
    prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(80 + i)).to(dev())
               for i, P in enumerate(lens)]
    Gs = [greedy_walk(perm, p, n_new + 40).to(dev()) for p in prompts]
    plans = [H.make_plan(64, 16, 50 + i) for i in range(3)]
    hooks = [_hook_for(Gs[i], plans[i]) for i in range(3)]
    singles = [dflash_generate(m, nt, prompts[i], cfg.mask_token_id, n_new, 16, None, 0.0, draft_token_hook=hooks[i])
               for i in range(3)]
    outs = dflash_generate_batch(m, nt, prompts, cfg.mask_token_id, n_new, 16, None, 0.0,
                                 draft_token_hook=lambda i, blk, s, c: hooks[i](blk[:, :min(16, lens[i] + n_new - s)], s, c))
    for i in range(3):
        assert outs[i].output_ids[0].tolist() == singles[i].output_ids[0].tolist() == Gs[i][:lens[i] + n_new].tolist(), i
        assert outs[i].acceptance_lengths == singles[i].acceptance_lengths


def _ewma(cands):
    from dflash_amd import EWMAPerformanceScheduler
    return EWMAPerformanceScheduler(candidates=list(cands), scheduler_mode="ewma", warmup_cycles=3, ewma_alpha=0.25,
                                    switch_margin=0.03, required_streak=2, cooldown_cycles=2, probe_interval=5,
                                    low_accept_threshold=0.2, low_accept_streak=3, adl_rho=0.3, adl_delta=1.0,
                                    adl_k_min=min(cands), adl_k_max=max(cands), adl_neighborhood=4)


def test_batched_policy_loop_one_scheduler_per_request():
    """dflash_generate_policy_batch (benchmark_dynamic_schedule.py:260-434, one EWMA scheduler per request): a group mixes
    8-, 12- and 16-row blocks in ONE pass over the weights.  Scheduler decisions depend on measured cycle times, so the
    check is what every schedule must satisfy: committed ids are the target's greedy walk, each cycle's block size is a
    candidate of that request's scheduler (or the clamped tail), acceptance lengths fit the blocks and add up, trace
    rows carry the reference's fields; a fixed schedule reproduces the single-request loop exactly."""
    from dflash_amd import dflash_generate_policy
    from dflash_amd.batch import dflash_generate_policy_batch
    from dflash_amd.synthetic import greedy_walk
    cfg, m, hf, nt, perm = _setup()
    lens, n_new = (33, 18, 50), 80
    prompts = [torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(90 + i)).to(dev())
               for i, P in enumerate(lens)]
    Gs = [greedy_walk(perm, p, n_new + 40).to(dev()) for p in prompts]
    plans = [H.make_plan(96, 16, 31 + i) for i in range(len(lens))]

    def bhook(i, blk, start, call):
        bs_now = min(16, lens[i] + n_new - start)
        k = min(plans[i][call], bs_now - 1)
        blk[0, 1:k + 1] = Gs[i][start + 1:start + k + 1]
        if k + 1 < 16:
            w = Gs[i][start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])

    cands = [(8, 12, 16), (8, 16), (12, 16)]
    outs = dflash_generate_policy_batch(model=m, target=nt, input_ids=prompts, mask_token_id=cfg.mask_token_id,
                                        max_new_tokens=n_new, stop_token_ids=None, temperature=0.0,
                                        schedulers=[_ewma(c) for c in cands], draft_token_hook=bhook)
    mixed = set()
    for i, o in enumerate(outs):
        assert o.output_ids[0].tolist() == Gs[i][:lens[i] + n_new].tolist(), f"request {i}"
        assert o.num_output_tokens == n_new and sum(o.acceptance_lengths) == n_new
        assert len(o.used_block_sizes) == len(o.acceptance_lengths) == len(o.cycle_trace) > 4
        for row, bs, tau in zip(o.cycle_trace, o.used_block_sizes, o.acceptance_lengths):
            assert row["chosen_block_size"] in cands[i] and bs == min(row["chosen_block_size"], lens[i] + n_new - row["start_idx"])
            assert 1 <= tau <= bs and row["tau"] == tau and row["cycle_s"] > 0
        mixed.update(o.used_block_sizes)
    assert len(mixed) >= 2            # the group really ran blocks of different sizes together

    class Fixed:                      # a schedule that does not depend on timing: 12-row blocks throughout
        candidates = (12,)
        tau_hat = cycle_hat = score_hat = {}
        adl_lgen_hat = adl_lacc_hat = None
        current = adl_target_k = adl_target_bs = 12

        def select(self, cyc):
            return 12

        def update(self, **_):
            pass

    outs12 = dflash_generate_policy_batch(model=m, target=nt, input_ids=prompts, mask_token_id=cfg.mask_token_id,
                                          max_new_tokens=n_new, stop_token_ids=None, temperature=0.0,
                                          schedulers=[Fixed() for _ in lens], draft_token_hook=bhook)
    for i, o in enumerate(outs12):
        def shook(blk, start, call, i=i):
            k = min(plans[i][call], blk.shape[1] - 1)
            blk[0, 1:k + 1] = Gs[i][start + 1:start + k + 1]
            if k + 1 < blk.shape[1]:
                w = Gs[i][start + k + 1]
                blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])
        ref = dflash_generate_policy(model=m, target=nt, input_ids=prompts[i], mask_token_id=cfg.mask_token_id,
                                     max_new_tokens=n_new, stop_token_ids=None, temperature=0.0, fixed_block_size=12,
                                     draft_token_hook=shook)
        assert o.output_ids[0].tolist() == ref.output_ids[0].tolist()
        assert o.acceptance_lengths == ref.acceptance_lengths and o.used_block_sizes == ref.used_block_sizes
    with pytest.raises(NotImplementedError):
        dflash_generate_policy_batch(model=m, target=nt, input_ids=prompts, mask_token_id=cfg.mask_token_id,
                                     max_new_tokens=8, stop_token_ids=None, temperature=0.7,
                                     schedulers=[_ewma((8, 16)) for _ in lens])


@pytest.mark.parametrize("block_size,lens", [(32, (33, 50)), (24, (40, 18)), (32, (25,)), (20, (21, 64, 30))])
def test_wide_blocks_in_the_ragged_batch(block_size, lens):
    """Blocks of 17..32 rows in the batch (VERDICT r2 next #5a; benchmark.py's block-size sweep with several requests per
    GPU): a request takes two of the group's four 16-row tiles.  Ids and acceptance lengths equal the single-request
    loop's (which runs these blocks as two tiles of one request), with acceptance plans that accept more than 16 rows,
    tail clamps through both tiles, and (three prompts) a second group."""
    from dflash_amd import dflash_generate
    from dflash_amd.batch import dflash_generate_batch
    from dflash_amd.synthetic import greedy_walk
    cfg, m, hf, nt, perm = _setup()
    n_new = 110
    prompts, Gs, plans = [], [], []
    for i, P in enumerate(lens):
        for seed in range(60 + 10 * i, 70 + 10 * i):     # (a walk that runs into the mask id would be trimmed there)
            p = torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(seed)).to(dev())
            G = greedy_walk(perm, p, n_new + 80).to(dev())
            if int((G[:P + n_new] == cfg.mask_token_id).sum()) == 0:
                break
        prompts.append(p)
        Gs.append(G)
        plans.append(H.make_plan(64, block_size, 23 + i))
    assert max(max(pl) for pl in plans) > 16           # some cycles accept into the second tile
    hooks = [_hook_for(Gs[i], plans[i]) for i in range(len(lens))]
    singles = [dflash_generate(m, nt, prompts[i], cfg.mask_token_id, n_new, block_size, None, 0.0, draft_token_hook=hooks[i])
               for i in range(len(lens))]

    def bhook(i, blk, start, call):
        # the batched block always has 32 slots; the single loop's block is block_size wide and shorter at the tail
        hooks[i](blk[:, :min(block_size, lens[i] + n_new - start)], start, call)

    outs = dflash_generate_batch(m, nt, prompts, cfg.mask_token_id, n_new, block_size, None, 0.0, draft_token_hook=bhook)
    for i, (a, b) in enumerate(zip(singles, outs)):
        assert a.output_ids[0].tolist() == Gs[i][:lens[i] + n_new].tolist(), f"request {i} (single)"
        assert b.output_ids[0].tolist() == a.output_ids[0].tolist(), f"request {i}"
        assert b.acceptance_lengths == a.acceptance_lengths, f"request {i}"


def test_hidden_5120_runs_through_the_ragged_batch_kernels():
    """hidden_size > 4096 (a 14B / 32B-class target; VERDICT r2 missing #5): the single-request kernels do not take it, the
    request runs as a group of one through the ragged-batch kernels.  Two layers of Qwen3-32B's widths (H 5120, 64 q / 8 kv
    heads) + a 2-layer draft of H 5120: native prefill vs the HF forward, then dflash_generate / spec_generate commit the
    target's greedy walk with the scripted acceptance lengths."""
    from dflash_amd import DFlashDraftModel, NativeTarget, dflash_generate
    from dflash_amd.config import DFlashConfig
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk, make_hf_qwen3
    from transformers import DynamicCache
    dims = dict(vocab_size=2048, hidden_size=5120, num_layers=2, num_heads=64, num_kv_heads=8, head_dim=128,
                intermediate_size=10240, rope_theta=1e6)
    torch.manual_seed(13)
    hf = make_hf_qwen3(dims, dev(), dtype=BF16)
    nt = NativeTarget(hf)
    assert nt.wide_hidden and nt.native_prefill
    P = 130
    prompt = torch.randint(0, 2000, (1, P), generator=torch.Generator().manual_seed(2)).to(dev())
    cache = nt.new_cache(P + 64)
    out = nt.prefill(prompt, cache, output_hidden_states=True, tap_layers=[0])
    rc = DynamicCache()
    with torch.inference_mode():
        ref = hf(prompt, position_ids=torch.arange(P, device=dev())[None], past_key_values=rc, use_cache=True,
                 logits_to_keep=1, output_hidden_states=True)
    H.assert_close("H5120 prefill logits (last row)", out.logits[0], ref.logits[0])
    H.assert_close("H5120 prefill tap layer 0", out.hidden_states[1][0], ref.hidden_states[1][0])
    H.assert_close("H5120 prefill K layer 1", cache.k[1][:, :P], rc.layers[1].keys[0], max_rel=H.KV_MAX_REL)
    with pytest.raises(NotImplementedError):
        nt.verify(prompt[0, :16], P, cache)
    # the loop: a large-margin greedy walk imposed on the target, scripted acceptance
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf)
    cfg = DFlashConfig(hidden_size=5120, num_hidden_layers=2, num_attention_heads=40, num_key_value_heads=8, head_dim=128,
                       intermediate_size=10240, vocab_size=2048, num_target_layers=2, block_size=16, rope_theta=1e6,
                       mask_token_id=2047, target_layer_ids=[0, 0])
    m = DFlashDraftModel(cfg, device=dev())
    m.load_state_dict(H.draft_weights(cfg, seed=3, dtype=BF16))
    assert m.wide_hidden
    n_new = 60
    p2 = torch.randint(0, 2000, (1, 37), generator=torch.Generator().manual_seed(9)).to(dev())
    G = greedy_walk(perm, p2, n_new + 40).to(dev())
    plan = H.make_plan(64, 16, 31)
    hook = _hook_for(G, plan)
    r = dflash_generate(m, nt, p2, cfg.mask_token_id, n_new, 16, None, 0.0, draft_token_hook=hook)
    assert r.output_ids[0].tolist() == G[:37 + n_new].tolist()
    exp, tot, c = [], 0, 0
    while tot < n_new:
        t = min(plan[c] + 1, n_new - tot)
        exp.append(t)
        tot += t
        c += 1
    assert r.acceptance_lengths == exp
    ids = m.spec_generate(target=nt, input_ids=p2, max_new_tokens=n_new, stop_token_ids=None, temperature=0.0,
                          draft_token_hook=_hook_for(G, plan))
    assert ids[0].tolist() == G[:37 + n_new].tolist()
