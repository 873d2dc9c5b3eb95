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
            out = s.cycle(16)
            assert out.tau == res[r][0], (cyc, r)
            # draft ids: random draft weights leave near-ties that the two summation orders may
            # break differently; the block's first token (committed) must agree, most others do
            assert s.block[0, 0] == blocks[r, 0]
            assert int((s.block[0] == blocks[r]).sum()) >= 12, (cyc, r, s.block[0].tolist(), blocks[r].tolist())
            assert torch.equal(s.output_ids[0, :s.start + 1], dec.output_ids[r, :s.start + 1])
            got, want = taps[r, :out.tau].float(), s.target_hidden[0].float()
            d = (got - want).abs()
            assert d.max() <= 4e-2 * want.abs().max() and d.mean() <= 4e-3 * want.abs().max()
            for li in (0, cfg.num_hidden_layers - 1):
                for a, b in ((dec.dk[r, li][:, :start], s.dcache.k[li][:, :start]),
                             (dec.dv[r, li][:, :start], s.dcache.v[li][:, :start])):
                    dd = (a.float() - b.float()).abs()
                    assert dd.max() <= 6e-2 * b.float().abs().max() and dd.mean() <= 4e-3 * b.float().abs().max()
            for li in (0, nt.L - 1):
                a, b = dec.tk[r, li][:, :s.start], s.tcache.k[li][:, :s.start]
                dd = (a.float() - b.float()).abs()
                assert dd.max() <= 6e-2 * b.float().abs().max() and dd.mean() <= 4e-3 * b.float().abs().max()
