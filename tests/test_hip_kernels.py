"""GPU parity of each C-ABI kernel against the oracle's leaf ops / plain fp32 torch
math on the same seeded inputs.  Integer-valued inputs give exact (layout) checks;
random bf16 inputs are checked within a stated tolerance."""
import json
import os

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from dflash_amd import ops as o
    return o


def dev():
    return torch.device("cuda", 0)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def unfrag(frag, K):
    return frag.view(K // 8, 16, 8).permute(1, 0, 2).reshape(16, K)


def to_frag(ops, x):  # x [rows<=16, K] bf16 cuda
    out = torch.empty(16 * x.shape[1], dtype=BF16, device=x.device)
    ops.pack_rows(x, x.shape[0], out)
    return out


# ------------------------------------------------------------------ layout
def test_pack_rows_roundtrip(ops):
    x = torch.randn(11, 512, generator=gen(0)).to(BF16).to(dev())
    fr = to_frag(ops, x)
    back = unfrag(fr, 512)
    assert torch.equal(back[:11], x)
    assert torch.count_nonzero(back[11:]) == 0


@pytest.mark.parametrize("N,K,rows,ksplit", [(64, 512, 16, 1), (48, 1024, 5, 1), (4096, 4096, 16, 1),
                                            (256, 4096 * 3, 16, 3), (512, 2560, 16, 1), (128, 9728, 7, 3),
                                            (64, 512, 16, 2)])
def test_gemm_f32_exact_small_ints(ops, N, K, rows, ksplit):
    """Small-integer operands: every product and partial sum is exact in fp32, so the
    result must equal torch's bit for bit — a pure test of fragment layouts, K
    splitting and the LDS reduction.  Asymmetric operands (guide §3)."""
    g = gen(N + K)
    w = torch.randint(-3, 4, (N, K), generator=g).to(BF16).to(dev())
    x = torch.randint(-2, 3, (rows, K), generator=g).to(BF16).to(dev())
    wp = ops.pack_weight(w)
    xf = to_frag(ops, x)
    out = torch.full((ksplit, 16, N), float("nan"), device=dev())
    ops.gemm_f32(wp, xf, None, 1, N, K, ksplit, out)
    got = out.sum(0)[:rows]
    ref = x.float() @ w.float().t()
    assert torch.equal(got, ref)


def test_gemm_f32_two_row_tiles(ops):
    N, K = 6144, 4096
    g = gen(5)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF16).to(dev())
    x0 = torch.randn(7, K, generator=g).to(BF16).to(dev())
    x1 = torch.randn(16, K, generator=g).to(BF16).to(dev())
    wp = ops.pack_weight(w)
    ks = ops.min_ksplit(K, 2)
    out = torch.zeros(ks, 32, N, device=dev())
    ops.gemm_f32(wp, to_frag(ops, x0), to_frag(ops, x1), 2, N, K, ks, out)
    got = out.sum(0)
    ref0, ref1 = x0.float() @ w.float().t(), x1.float() @ w.float().t()
    # fp32 accumulation in a different order than torch: relative 1e-4 of the row scale
    for a, b in ((got[:7], ref0), (got[16:], ref1)):
        assert (a - b).abs().max() <= 2e-4 * b.abs().max()


def test_gemm_silu_mul(ops):
    I, K = 2560, 1024
    g = gen(6)
    wg = (torch.randn(I, K, generator=g) * 0.05).to(BF16).to(dev())
    wu = (torch.randn(I, K, generator=g) * 0.05).to(BF16).to(dev())
    x = torch.randn(16, K, generator=g).to(BF16).to(dev())
    act = torch.empty(16 * I, dtype=BF16, device=dev())
    ops.gemm_silu_mul(ops.pack_weight_gateup(wg, wu), to_frag(ops, x), I, K, act)
    got = unfrag(act, I).float()
    gl = (x.float() @ wg.float().t()).to(BF16)
    ul = (x.float() @ wu.float().t()).to(BF16)
    ref = (torch.nn.functional.silu(gl.float()).to(BF16).float() * ul.float()).to(BF16).float()
    # identical rounding points; residual differences are 1-ulp flips of the two bf16 Linear outputs
    assert (got - ref).abs().max() <= 2 ** -6 * ref.abs().max()
    assert ((got - ref).abs() > 0).float().mean() < 0.05


def test_gemm_silu_mul_exact_ints(ops):
    I, K = 64, 512
    g = gen(7)
    wg = torch.randint(-1, 2, (I, K), generator=g).to(BF16).to(dev())
    wu = torch.randint(-1, 2, (I, K), generator=g).to(BF16).to(dev())
    x = torch.zeros(16, K, dtype=BF16)
    x[:, :8] = torch.randint(-1, 2, (16, 8), generator=g).to(BF16)
    x = x.to(dev())
    act = torch.empty(16 * I, dtype=BF16, device=dev())
    ops.gemm_silu_mul(ops.pack_weight_gateup(wg, wu), to_frag(ops, x), I, K, act)
    gl, ul = x.float() @ wg.float().t(), x.float() @ wu.float().t()
    ref = (torch.nn.functional.silu(gl).to(BF16).float() * ul).to(BF16).float()
    got = unfrag(act, I).float()
    assert (got - ref).abs().max() <= 2 ** -7 * max(1.0, ref.abs().max())  # __expf vs torch exp


@pytest.mark.parametrize("V,K,bs", [(2048, 512, 16), (4096 + 16 * 7, 1024, 12), (151936, 4096, 16)])
def test_gemm_argmax(ops, V, K, bs):
    g = gen(V)
    w = (torch.randn(V, K, generator=g) * 0.02).to(BF16).to(dev())
    x = torch.randn(16, K, generator=g).to(BF16).to(dev())
    w[V // 3] = w[5]          # duplicated rows: exact ties, the lower index must win
    w[V - 1] = w[5]
    wp = ops.pack_weight(w)
    ids = torch.full((16,), -1, dtype=torch.long, device=dev())
    logits = torch.zeros(16, V, dtype=BF16, device=dev())
    ops.gemm_argmax(wp, to_frag(ops, x), V, K, 1, bs - 1, ops.argmax_ws(dev()), ids, 1, logits=logits)
    ref_logits = (x.float() @ w.float().t())
    assert (logits[1:bs].float() - ref_logits[1:bs]).abs().max() <= 2 ** -7 * ref_logits.abs().max()
    # ids are the first-max index of the kernel's own bf16 logits (the reference's argmax semantics)
    assert torch.equal(ids[1:bs], torch.argmax(logits[1:bs], dim=-1))
    assert int(ids[0]) == -1 and (ids[bs:] == -1).all()
    # and, with the fused path (no logits written), the same ids
    ids2 = torch.full((16,), -1, dtype=torch.long, device=dev())
    ops.gemm_argmax(wp, to_frag(ops, x), V, K, 1, bs - 1, ops.argmax_ws(dev()), ids2, 1)
    assert torch.equal(ids, ids2)


@pytest.mark.parametrize("V,K,bs", [(2048, 512, 16), (4096 + 16 * 7, 1024, 9), (151936, 4096, 16)])
def test_gemm_argmax_margins(ops, V, K, bs):
    """Top-1 minus top-2 logit per row (the reference's confidence statistic,
    benchmark_candidate_solutions.py:296-302) from the fused kernel == torch.topk(2) on the
    bf16 logits the same kernel materialises; duplicated rows give exact ties (margin 0)."""
    g = gen(V + 1)
    w = (torch.randn(V, K, generator=g) * 0.02).to(BF16).to(dev())
    x = torch.randn(16, K, generator=g).to(BF16).to(dev())
    wp = ops.pack_weight(w)
    ids = torch.full((16,), -1, dtype=torch.long, device=dev())
    mg = torch.full((16,), -1.0, dtype=torch.float32, device=dev())
    logits = torch.zeros(16, V, dtype=BF16, device=dev())
    ops.gemm_argmax(wp, to_frag(ops, x), V, K, 1, bs - 1, ops.argmax_ws(dev()), ids, 1, logits=logits, margins=mg)
    top2 = torch.topk(logits[1:bs].float(), 2, dim=-1).values
    assert torch.equal(mg[1:bs], top2[:, 0] - top2[:, 1])
    assert torch.equal(ids[1:bs], torch.argmax(logits[1:bs], dim=-1))
    assert (mg[bs:] == -1).all() and float(mg[0]) == -1
    # make the winner of every row a duplicated weight row: the runner-up ties, margin 0
    w2 = w.clone()
    for r in range(1, bs):
        w2[(int(ids[r]) + V // 2) % V] = w2[int(ids[r])]
    ops.gemm_argmax(ops.pack_weight(w2), to_frag(ops, x), V, K, 1, bs - 1, ops.argmax_ws(dev()), ids, 1, margins=mg)
    assert (mg[1:bs] == 0).all()


def test_gemm_argmax_forced_tie(ops):
    V, K = 2048, 512
    w = torch.zeros(V, K, dtype=BF16)
    w[[77, 900, 2047], 0] = 1.0
    x = torch.zeros(16, K, dtype=BF16)
    x[:, 0] = 1.0
    ids = torch.zeros(16, dtype=torch.long, device=dev())
    ops.gemm_argmax(ops.pack_weight(w.to(dev())), to_frag(ops, x.to(dev())), V, K, 0, 16, ops.argmax_ws(dev()), ids, 0)
    assert (ids == 77).all()
    x[:, 0] = -1.0   # now all maxima are the zero rows: index 0 wins
    ops.gemm_argmax(ops.pack_weight(w.to(dev())), to_frag(ops, x.to(dev())), V, K, 0, 16, ops.argmax_ws(dev()), ids, 0)
    assert (ids == 0).all()


def test_gemm_row_sources_and_resid_epilogue(ops):
    """The fused residual/norm pipeline: embed_rows -> GEMM reading (h, ss) with the
    RMSNorm applied in its prologue -> GEMM with the residual epilogue (h, taps, new ss),
    against the oracle's rms_norm + plain fp32 matmuls.  Also the K-chunked form."""
    from oracle.dflash_oracle import rms_norm
    Hd, I, V, bs = 1024, 12288 // 4, 512, 11
    g = gen(21)
    emb = (torch.randn(V, Hd, generator=g) * 0.7).to(BF16)
    ids = torch.randint(0, V, (16,), generator=g)
    nw = (1 + 0.1 * torch.randn(Hd, generator=g)).to(BF16)
    w1 = (torch.randn(I, Hd, generator=g) * 0.03).to(BF16)      # consumes norm(h)
    w2 = (torch.randn(Hd, I, generator=g) * 0.03).to(BF16)      # K = 3072 > 128 k-steps x ... chunked path? (3072/32=96: single pass)
    w3 = (torch.randn(Hd, 20480 // 4, generator=g) * 0.02).to(BF16)  # K = 5120 -> 160 k-steps: chunked
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 7, bs, 0)
    h = torch.zeros(16, Hd, dtype=BF16, device=dev())
    ss0 = torch.zeros(16, device=dev())
    ops.embed_rows(emb.to(dev()), ids.to(dev()), h, Hd, ss0, dyn, ops.DYN_BS)
    assert torch.equal(h[:bs].cpu(), emb[ids][:bs]) and torch.count_nonzero(h[bs:]) == 0
    assert torch.allclose(ss0[:bs].cpu(), emb[ids][:bs].float().pow(2).sum(-1), rtol=1e-5)
    # GEMM with the norm in its prologue == GEMM on oracle-normalised rows
    out = torch.zeros(1, 16, I, device=dev())
    ops.gemm_f32(ops.pack_weight(w1.to(dev())), ops.rows_normed(h, ss0, 1, nw.to(dev()), 1e-6, ops.DYN_BS), None, 1, I, Hd,
                 1, out, dyn)
    xn = rms_norm(emb[ids], nw, 1e-6)
    ref = xn.float() @ w1.float().t()
    d = (out[0, :bs].cpu() - ref[:bs]).abs()
    assert d.max() <= 2e-2 * ref.abs().max() and d.mean() <= 1e-3 * ref.abs().max()   # 1-ulp flips of normed inputs
    assert torch.count_nonzero(out[0, bs:]) == 0                                        # rows >= valid are zero
    # residual epilogue, single pass (K = 3072)
    act = (torch.randn(16, I, generator=g) * 0.5).to(BF16)
    act_frag = to_frag(ops, act.to(dev()))
    taps = torch.zeros(16, 3 * Hd, dtype=BF16, device=dev())
    ss1 = torch.zeros(Hd, device=dev())
    h0 = h.clone()
    ops.gemm_resid(ops.pack_weight(w2.to(dev())), act_frag, Hd, I, h, add_residual=True, ss_out=ss1,
                   tap=taps[:, Hd:2 * Hd], dyn=dyn)
    lin = (act.float() @ w2.float().t()).to(BF16)
    h_ref = (h0.cpu() + lin)
    dd = (h.cpu().float() - h_ref.float()).abs()
    assert dd.max() <= 2 ** -6 * h_ref.float().abs().max() and (dd > 0).float().mean() < 0.02
    assert torch.equal(taps[:, Hd:2 * Hd], h) and torch.count_nonzero(taps[:, :Hd]) == 0
    ssq = ss1.view(Hd // 16, 16).sum(0).cpu()
    assert torch.allclose(ssq, h.cpu().float().pow(2).sum(-1), rtol=1e-4)
    # chunked K (fc-like): plain rows source with 7 valid rows, no residual
    th = (torch.randn(7, 5120, generator=g)).to(BF16).to(dev())
    ctxh = torch.full((16, Hd), 7.0, dtype=BF16, device=dev())
    ssc = torch.zeros(Hd, device=dev())
    ops.gemm_resid(ops.pack_weight(w3.to(dev())), ops.rows_plain(th, ops.DYN_TAU), Hd, 5120, ctxh, add_residual=False,
                   ss_out=ssc, dyn=dyn)
    ref3 = (th.cpu().float() @ w3.float().t()).to(BF16)
    d3 = (ctxh[:7].cpu().float() - ref3.float()).abs()
    assert d3.max() <= 2 ** -6 * ref3.float().abs().max() and (d3 > 0).float().mean() < 0.02
    assert torch.count_nonzero(ctxh[7:]) == 0
    # exact-integer check of the chunked path (layout / chunk bookkeeping)
    wi = torch.randint(-2, 3, (64, 4096 * 3), generator=g).to(BF16)
    xi = torch.randint(-2, 3, (16, 4096 * 3), generator=g).to(BF16)
    hi = torch.zeros(16, 64, dtype=BF16, device=dev())
    ops.gemm_resid(ops.pack_weight(wi.to(dev())), to_frag(ops, xi.to(dev())), 64, 4096 * 3, hi, add_residual=False)
    assert torch.equal(hi.cpu().float(), (xi.float() @ wi.float().t()).to(BF16).float())


# ------------------------------------------------------------------ row stages
def test_norm_pack_variants(ops):
    from oracle.dflash_oracle import rms_norm
    Hd = 1024
    g = gen(8)
    nw = (1 + 0.1 * torch.randn(Hd, generator=g)).to(BF16)
    part = torch.randn(3, 16, Hd, generator=g)
    resid = torch.randn(16, Hd, generator=g).to(BF16)
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 5, 12, 0)
    frag = torch.empty(16 * Hd, dtype=BF16, device=dev())
    hout = torch.zeros(16, Hd, dtype=BF16, device=dev())
    # partial sums + residual (o_proj / down_proj epilogue), 12 valid rows
    ops.norm_pack(norm_w=nw.to(dev()), frag=frag, H=Hd, eps=1e-6, part=part.to(dev()), nsplit=3,
                  part_split=16 * Hd, ldp=Hd, resid_in=resid.to(dev()), h_out=hout, dyn=dyn, dyn_word=ops.DYN_BS)
    h_ref = resid + part.sum(0).to(BF16)
    n_ref = rms_norm(h_ref, nw, 1e-6)
    assert torch.equal(hout[:12].cpu(), h_ref[:12])
    got = unfrag(frag, Hd).cpu()
    assert torch.count_nonzero(got[12:]) == 0
    d = (got[:12].float() - n_ref[:12].float()).abs()
    assert d.max() <= 2 ** -6 * n_ref.float().abs().max() and (d > 0).float().mean() < 0.02
    # partial only (fc -> hidden_norm), 5 valid rows
    ops.norm_pack(norm_w=nw.to(dev()), frag=frag, H=Hd, eps=1e-6, part=part.to(dev()), nsplit=3,
                  part_split=16 * Hd, ldp=Hd, dyn=dyn, dyn_word=ops.DYN_TAU)
    n_ref = rms_norm(part.sum(0).to(BF16), nw, 1e-6)
    got = unfrag(frag, Hd).cpu()
    assert torch.count_nonzero(got[5:]) == 0
    assert (got[:5].float() - n_ref[:5].float()).abs().max() <= 2 ** -6 * n_ref.float().abs().max()
    # embedding gather
    emb = torch.randn(100, Hd, generator=g).to(BF16)
    ids = torch.randint(0, 100, (16,), generator=g)
    ops.norm_pack(norm_w=nw.to(dev()), frag=frag, H=Hd, eps=1e-6, embed=emb.to(dev()), ids=ids.to(dev()),
                  h_out=hout, dyn=dyn, dyn_word=ops.DYN_BS)
    assert torch.equal(hout[:12].cpu(), emb[ids][:12])
    n_ref = rms_norm(emb[ids], nw, 1e-6)
    assert (unfrag(frag, Hd).cpu()[:12].float() - n_ref[:12].float()).abs().max() <= 2 ** -6 * n_ref.float().abs().max()


def test_qknorm_rope_append(ops):
    from oracle import dflash_oracle as O
    from dflash_amd.model import _rope_tables
    n_q, n_kv, S, tau, bs = 8, 2, 37, 5, 12
    ld = (n_q + 2 * n_kv) * 128
    g = gen(9)
    part = torch.randn(2, 32, ld, generator=g)
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16)
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16)
    cos, sin = _rope_tables(128, 1e6, 256, dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, S, tau, bs, S)
    kc = torch.zeros(n_kv, 128, 128, dtype=BF16, device=dev())
    vc = torch.zeros_like(kc)
    q_out = torch.zeros(n_q, 16, 128, dtype=BF16, device=dev())
    ops.qknorm_rope_append(qkv=part.to(dev()), nsplit=2, split_stride=32 * ld, ld=ld, q_col=0, k_col=n_q * 128,
                           v_col=(n_q + n_kv) * 128, ctx_row0=0, blk_row0=16, n_q=n_q, n_kv=n_kv,
                           q_norm_w=qw.to(dev()), k_norm_w=kw.to(dev()), eps=1e-6, cos_tab=cos, sin_tab=sin,
                           q_out=q_out, kcache=kc, vcache=vc, dyn=dyn)
    lin = part.sum(0).to(BF16)                      # the Linear outputs
    rows = torch.cat([lin[:tau], lin[16:16 + bs]])  # [tau+bs, ld]
    q = lin[16:16 + bs, :n_q * 128].view(1, bs, n_q, 128)
    k = rows[:, n_q * 128:(n_q + n_kv) * 128].view(1, tau + bs, n_kv, 128)
    v = rows[:, (n_q + n_kv) * 128:].view(1, tau + bs, n_kv, 128).transpose(1, 2)
    q = O.rms_norm(q, qw, 1e-6).transpose(1, 2)
    k = O.rms_norm(k, kw, 1e-6).transpose(1, 2)
    pos = torch.arange(S, S + tau + bs)[None]
    c, s = O.rope_cos_sin(pos, O.rope_inv_freq(128, 1e6), BF16)
    q, k = O.apply_rotary_dflash(q, k, c, s)
    assert torch.equal(vc[:, S:S + tau + bs].cpu(), v[0])
    assert torch.count_nonzero(vc[:, :S]) == 0 and torch.count_nonzero(vc[:, S + tau + bs:]) == 0
    for got, ref in ((kc[:, S:S + tau + bs].cpu(), k[0]), (q_out[:, :bs].cpu(), q[0])):
        d = (got.float() - ref.float()).abs()
        assert d.max() <= 2 ** -6 * ref.float().abs().max() and (d > 0).float().mean() < 0.02


@pytest.mark.parametrize("n_q,n_kv,S,tau,bs", [(4, 2, 0, 3, 16), (8, 2, 100, 16, 16), (32, 8, 1024, 7, 16),
                                               (32, 4, 517, 1, 8), (8, 8, 31, 1, 12), (32, 8, 4000, 16, 16)])
def test_block_attn(ops, n_q, n_kv, S, tau, bs):
    g = gen(S + n_q)
    kv_len = S + tau + bs
    rows = kv_len + 40
    q = torch.randn(n_q, 16, 128, generator=g).to(BF16)
    k = torch.randn(n_kv, rows, 128, generator=g).to(BF16)
    v = torch.randn(n_kv, rows, 128, generator=g).to(BF16)
    k[0, 3] *= 4.0  # a spike, so one tile's max dominates and the rescale path matters
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, S, tau, bs, S)
    out = torch.empty(16 * n_q * 128, dtype=BF16, device=dev())
    ms = 32
    ops.block_attn(q=q.to(dev()), kcache=k.to(dev()), vcache=v.to(dev()), n_q=n_q, n_kv=n_kv, scale=128 ** -0.5,
                   dyn=dyn, kv_len_max=kv_len, ws=ops.attn_ws(n_q, ms, dev()), max_splits=ms, out_frag=out)
    got = unfrag(out, n_q * 128).cpu().float().view(16, n_q, 128)
    G = n_q // n_kv
    kk = k[:, :kv_len].float().repeat_interleave(G, dim=0)
    vv = v[:, :kv_len].float().repeat_interleave(G, dim=0)
    p = torch.softmax(torch.einsum("hqd,hkd->hqk", q.float(), kk) * 128 ** -0.5, dim=-1)
    ref = torch.einsum("hqk,hkd->qhd", p, vv)
    # bf16 P and bf16 output: 2^-7 relative of the output scale
    assert (got - ref).abs().max() <= 2 ** -6 * ref.abs().max()
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("n_q,n_kv,S,bs", [(4, 2, 0, 16), (32, 8, 1024, 16), (8, 2, 131, 12), (32, 8, 3, 5)])
def test_block_attn_causal(ops, n_q, n_kv, S, bs):
    """Target-verify form: query row j sees cache rows <= S + j."""
    g = gen(S + n_q + 1)
    kv_len = S + bs
    q = torch.randn(n_q, 16, 128, generator=g).to(BF16)
    k = torch.randn(n_kv, kv_len + 8, 128, generator=g).to(BF16)
    v = torch.randn(n_kv, kv_len + 8, 128, generator=g).to(BF16)
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, S, 0, bs, S)
    out = torch.empty(16 * n_q * 128, dtype=BF16, device=dev())
    ops.block_attn(q=q.to(dev()), kcache=k.to(dev()), vcache=v.to(dev()), n_q=n_q, n_kv=n_kv, scale=128 ** -0.5,
                   dyn=dyn, kv_len_max=kv_len, ws=ops.attn_ws(n_q, 32, dev()), max_splits=32, out_frag=out,
                   causal=True)
    got = unfrag(out, n_q * 128).cpu().float().view(16, n_q, 128)[:bs]
    G = n_q // n_kv
    kk = k[:, :kv_len].float().repeat_interleave(G, dim=0)
    vv = v[:, :kv_len].float().repeat_interleave(G, dim=0)
    sc = torch.einsum("hqd,hkd->hqk", q[:, :bs].float(), kk) * 128 ** -0.5
    mask = torch.arange(kv_len)[None, :] > (S + torch.arange(bs))[:, None]
    sc = sc.masked_fill(mask[None], float("-inf"))
    ref = torch.einsum("hqk,hkd->qhd", torch.softmax(sc, dim=-1), vv)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2 ** -6 * ref.abs().max()


def test_qknorm_rope_without_norm(ops):
    """Llama-style attention: no per-head q/k norm, RoPE only."""
    from oracle import dflash_oracle as O
    from dflash_amd.model import _rope_tables
    n_q, n_kv, S, bs = 4, 2, 9, 7
    ld = (n_q + 2 * n_kv) * 128
    part = torch.randn(1, 16, ld, generator=gen(3))
    cos, sin = _rope_tables(128, 5e5, 128, dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, S, 0, bs, S)
    kc = torch.zeros(n_kv, 64, 128, dtype=BF16, device=dev())
    vc = torch.zeros_like(kc)
    q_out = torch.zeros(n_q, 16, 128, dtype=BF16, device=dev())
    ops.qknorm_rope_append(qkv=part.to(dev()), nsplit=1, split_stride=16 * ld, ld=ld, q_col=0, k_col=n_q * 128,
                           v_col=(n_q + n_kv) * 128, ctx_row0=0, blk_row0=0, n_q=n_q, n_kv=n_kv, q_norm_w=None,
                           k_norm_w=None, eps=1e-6, cos_tab=cos, sin_tab=sin, q_out=q_out, kcache=kc, vcache=vc,
                           dyn=dyn)
    lin = part[0].to(BF16)[:bs]
    q = lin[:, :n_q * 128].view(1, bs, n_q, 128).transpose(1, 2)
    k = lin[:, n_q * 128:(n_q + n_kv) * 128].view(1, bs, n_kv, 128).transpose(1, 2)
    c, s = O.rope_cos_sin(torch.arange(S, S + bs)[None], O.rope_inv_freq(128, 5e5), BF16)
    qr, kr = O.apply_rotary_std(q, k, c, s)
    assert torch.equal(q_out[:, :bs].cpu(), qr[0])
    assert torch.equal(kc[:, S:S + bs].cpu(), kr[0])


@pytest.mark.parametrize("n_q,n_kv,S,tau,bs,causal", [(4, 2, 0, 3, 16, False), (8, 2, 37, 5, 12, False),
                                                      (32, 8, 1024, 7, 16, False), (32, 4, 200, 16, 16, False),
                                                      (32, 8, 1024, 0, 16, True), (8, 8, 45, 0, 9, True),
                                                      (4, 2, 0, 0, 16, True), (32, 8, 1439, 16, 16, False),
                                                      (32, 8, 3000, 3, 16, True), (8, 2, 127, 1, 2, False)])
def test_attn_fused_equals_three_launch_path(ops, n_q, n_kv, S, tau, bs, causal):
    """dfl_attn_fused (1 launch) vs dfl_qknorm_rope_append + dfl_block_attn (3 launches):
    the appended V must be bit-identical and K equal up to rare 1-ulp flips, the attention
    output equal within the rounding of a different key-split order, and both within
    tolerance of fp32 torch."""
    from dflash_amd.model import _rope_tables
    g = gen(S + n_q + bs)
    ld = (n_q + 2 * n_kv) * 128
    part = torch.randn(2, 32, ld, generator=g).to(dev())
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    cos, sin = _rope_tables(128, 1e6, 2048, dev())
    rows = S + tau + bs + 8
    k0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    v0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, S, tau, bs, S)
    common = dict(qkv=part, nsplit=2, split_stride=32 * ld, ld=ld, q_col=0, k_col=n_q * 128, v_col=(n_q + n_kv) * 128,
                  ctx_row0=0, blk_row0=16, n_q=n_q, n_kv=n_kv, q_norm_w=qw, k_norm_w=kw, eps=1e-6, cos_tab=cos,
                  sin_tab=sin, dyn=dyn)
    # reference path
    k1, v1 = k0.clone(), v0.clone()
    q1 = torch.zeros(n_q, 16, 128, dtype=BF16, device=dev())
    out1 = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev())
    ops.qknorm_rope_append(**common, q_out=q1, kcache=k1, vcache=v1)
    ops.block_attn(q=q1, kcache=k1, vcache=v1, n_q=n_q, n_kv=n_kv, scale=128 ** -0.5, dyn=dyn,
                   kv_len_max=S + tau + bs, ws=ops.attn_ws(n_q, 32, dev()), max_splits=32, out_frag=out1,
                   causal=causal)
    # fused
    k2, v2 = k0.clone(), v0.clone()
    out2 = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev())
    fws = ops.attn_fused_ws(n_q, n_kv, 32, dev())
    for _ in range(2):   # twice: the arrival tickets must be back at zero after a launch
        out2.zero_()
        ops.attn_fused(**common, kcache=k2, vcache=v2, scale=128 ** -0.5, kv_len_max=S + tau + bs, ws=fws,
                       max_splits=32, out_frag=out2, causal=causal)
    # V is a rounded copy: identical.  K differs only where the per-head RMS (summed in a
    # different fp32 order by the two kernels) moves a value across a bf16 rounding edge.
    assert torch.equal(v1, v2)
    dk = (k1.float() - k2.float()).abs()
    assert (dk > 0).float().mean() < 1e-3 and dk.max() <= 2 ** -7 * k1.float().abs().max()
    a = unfrag(out1, n_q * 128).float()[:bs]
    b = unfrag(out2, n_q * 128).float()[:bs]
    assert torch.isfinite(b).all()
    assert (a - b).abs().max() <= 2 ** -6 * a.abs().max()
    # fp32 reference from the appended cache
    kv_len = S + tau + bs
    G = n_q // n_kv
    kk = k2[:, :kv_len].float().repeat_interleave(G, dim=0)
    vv = v2[:, :kv_len].float().repeat_interleave(G, dim=0)
    sc = torch.einsum("hqd,hkd->hqk", q1[:, :bs].float(), kk) * 128 ** -0.5
    if causal:
        mask = torch.arange(kv_len, device=dev())[None, :] > (S + tau + torch.arange(bs, device=dev()))[:, None]
        sc = sc.masked_fill(mask[None], float("-inf"))
    ref = torch.einsum("hqk,hkd->qhd", torch.softmax(sc, dim=-1), vv).reshape(bs, n_q * 128)
    assert (b - ref).abs().max() <= 2 ** -6 * ref.abs().max()


@pytest.mark.parametrize("n_q,n_kv,S,tau,bs,causal,use_dyn", [
    (4, 2, 0, 3, 16, False, False), (8, 2, 37, 5, 12, False, True), (32, 8, 1024, 7, 16, False, False),
    (32, 8, 1041, 0, 16, True, False), (32, 4, 300, 16, 16, False, False), (4, 4, 70, 2, 9, False, True),
    (8, 2, 3000, 16, 16, True, True), (32, 8, 9001, 11, 16, False, False), (4, 2, 0, 0, 16, True, False),
    (8, 8, 255, 16, 5, True, False), (32, 8, 31, 1, 1, True, True),
    # >= ~5k cached keys: two query heads per workgroup share every K/V tile (k_attn_head_pair), causal and not,
    # lengths as immediates and from the device record, a ragged last tile
    (8, 2, 6000, 0, 16, True, False), (16, 8, 7777, 9, 7, False, True), (4, 1, 5300, 16, 16, True, True)])
def test_attn_head_equals_three_launch_path(ops, n_q, n_kv, S, tau, bs, causal, use_dyn):
    """dfl_attn_head (finished bf16 q/k/v rows in, one launch, one query head per workgroup, wave results
    merged in LDS, sc1 partials) against dfl_qknorm_rope_append + dfl_block_attn fed with the SAME Linear
    outputs, and against fp32 torch: appended V bit-identical, K up to rare 1-ulp flips, output within the
    rounding of a different key-split order.  Lengths as immediates and from the device record; GQA groups
    1, 2, 4, 8; zero cached keys; more splits than max_splits allows (S = 9001)."""
    from dflash_amd.model import _rope_tables
    g = gen(S + n_q + bs + 7)
    ld = (n_q + 2 * n_kv) * 128
    x = torch.randn(32, ld, generator=g).to(BF16).to(dev())           # rows 0..15 context, 16..31 block
    part = x.float()[None].contiguous()                                  # the same values as one fp32 "partial"
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    cos, sin = _rope_tables(128, 1e6, 16384, dev())
    rows = S + tau + bs + 8
    k0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    v0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, S, tau, bs, S)
    k1, v1 = k0.clone(), v0.clone()
    q1 = torch.zeros(n_q, 16, 128, dtype=BF16, device=dev())
    out1 = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev())
    ops.qknorm_rope_append(qkv=part, nsplit=1, split_stride=32 * ld, ld=ld, q_col=0, k_col=n_q * 128,
                           v_col=(n_q + n_kv) * 128, ctx_row0=0, blk_row0=16, n_q=n_q, n_kv=n_kv, q_norm_w=qw,
                           k_norm_w=kw, eps=1e-6, cos_tab=cos, sin_tab=sin, dyn=dyn, q_out=q1, kcache=k1, vcache=v1)
    ops.block_attn(q=q1, kcache=k1, vcache=v1, n_q=n_q, n_kv=n_kv, scale=128 ** -0.5, dyn=dyn,
                   kv_len_max=S + tau + bs, ws=ops.attn_ws(n_q, 32, dev()), max_splits=32, out_frag=out1,
                   causal=causal)
    k2, v2 = k0.clone(), v0.clone()
    out2 = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev())
    hws = ops.attn_head_ws(n_q, 8, 1, dev())
    for _ in range(2):   # twice: the arrival tickets must be back at zero after a launch
        out2.fill_(float("nan"))
        ops.attn_head(xq=x[16:], q_col=0, k_col=n_q * 128, v_col=(n_q + n_kv) * 128, xc=x[:16], ck_col=n_q * 128,
                      cv_col=(n_q + n_kv) * 128, n_q=n_q, n_kv=n_kv, q_norm_w=qw, k_norm_w=kw, eps=1e-6, cos_tab=cos,
                      sin_tab=sin, kcache=k2, vcache=v2, scale=128 ** -0.5, causal=causal, S=S, tau=tau, bs=bs, pos0=S,
                      dyn=dyn if use_dyn else None, ws=hws, max_splits=8, out_frag=out2)
    assert torch.equal(v1, v2)
    dk = (k1.float() - k2.float()).abs()
    assert (dk > 0).float().mean() < 1e-3 and dk.max() <= 2 ** -7 * k1.float().abs().max()
    a = unfrag(out1, n_q * 128).float()[:bs]
    b = unfrag(out2, n_q * 128).float()[:bs]
    assert torch.isfinite(b).all()
    assert (a - b).abs().max() <= 2 ** -6 * a.abs().max()
    kv_len = S + tau + bs
    G = n_q // n_kv
    kk = k2[:, :kv_len].float().repeat_interleave(G, dim=0)
    vv = v2[:, :kv_len].float().repeat_interleave(G, dim=0)
    sc = torch.einsum("hqd,hkd->hqk", q1[:, :bs].float(), kk) * 128 ** -0.5
    if causal:
        mask = torch.arange(kv_len, device=dev())[None, :] > (S + tau + torch.arange(bs, device=dev()))[:, None]
        sc = sc.masked_fill(mask[None], float("-inf"))
    ref = torch.einsum("hqk,hkd->qhd", torch.softmax(sc, dim=-1), vv).reshape(bs, n_q * 128)
    assert (b - ref).abs().max() <= 2 ** -6 * ref.abs().max()


@pytest.mark.parametrize("n_q,n_kv,S,tau,bs,causal", [(8, 2, 90, 0, 29, True), (32, 8, 1030, 20, 24, False),
                                                      (4, 2, 0, 32, 32, False), (32, 4, 70, 5, 17, True)])
def test_attn_head_two_query_tiles(ops, n_q, n_kv, S, tau, bs, causal):
    """Blocks of 17..32 rows (results.md:11-16 sweeps 20 and 24): two 16-row query tiles per workgroup, up to
    64 new rows (two LDS tiles), against fp32 torch computed from the kernel's own q-norm/RoPE reference
    (dfl_qknorm_rope_append run per 16-row tile)."""
    from dflash_amd.model import _rope_tables
    g = gen(S + n_q + bs + 3)
    ld = (n_q + 2 * n_kv) * 128
    xc = torch.randn(32, ld, generator=g).to(BF16).to(dev())
    xq = torch.randn(32, ld, generator=g).to(BF16).to(dev())
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    cos, sin = _rope_tables(128, 1e6, 4096, dev())
    rows = S + tau + bs + 8
    k0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    v0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    # reference K/V/q rows from the row-wise kernel, 16 rows at a time (context tiles, then block tiles)
    k1, v1 = k0.clone(), v0.clone()
    q1 = torch.zeros(2, n_q, 16, 128, dtype=BF16, device=dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    done = 0
    for src, n, is_blk in ((xc, tau, False), (xq, bs, True)):
        for r0 in range(0, n, 16):
            nr = min(16, n - r0)
            buf = torch.zeros(32, ld, dtype=torch.float32, device=dev())
            buf[16 if is_blk else 0:][:nr] = src[r0:r0 + nr].float()
            ops.set_dyn(dyn, S + done, 0 if is_blk else nr, nr if is_blk else 0, S + done)
            ops.qknorm_rope_append(qkv=buf[None].contiguous(), nsplit=1, split_stride=32 * ld, ld=ld, q_col=0,
                                   k_col=n_q * 128, v_col=(n_q + n_kv) * 128, ctx_row0=0, blk_row0=16, n_q=n_q, n_kv=n_kv,
                                   q_norm_w=qw, k_norm_w=kw, eps=1e-6, cos_tab=cos, sin_tab=sin, dyn=dyn,
                                   q_out=q1[r0 // 16] if is_blk else q1[0].clone(), kcache=k1, vcache=v1)
            done += nr
    k2, v2 = k0.clone(), v0.clone()
    out2 = torch.full((2, 16 * n_q * 128), float("nan"), dtype=BF16, device=dev())
    hws = ops.attn_head_ws(n_q, 8, 2, dev())
    ops.attn_head(xq=xq, q_col=0, k_col=n_q * 128, v_col=(n_q + n_kv) * 128, xc=xc, ck_col=n_q * 128,
                  cv_col=(n_q + n_kv) * 128, n_q=n_q, n_kv=n_kv, q_norm_w=qw, k_norm_w=kw, eps=1e-6, cos_tab=cos,
                  sin_tab=sin, kcache=k2, vcache=v2, scale=128 ** -0.5, causal=causal, S=S, tau=tau, bs=bs, pos0=S,
                  ws=hws, max_splits=8, out_frag=out2, q_tiles=2, out_tile_stride=out2.stride(0))
    kv_len = S + tau + bs
    assert torch.equal(v1[:, :kv_len], v2[:, :kv_len])
    dk = (k1[:, :kv_len].float() - k2[:, :kv_len].float()).abs()
    assert (dk > 0).float().mean() < 1e-3 and dk.max() <= 2 ** -7 * k1.float().abs().max()
    G = n_q // n_kv
    kk = k2[:, :kv_len].float().repeat_interleave(G, dim=0)
    vv = v2[:, :kv_len].float().repeat_interleave(G, dim=0)
    qq = torch.cat([q1[0], q1[1]], dim=1)[:, :bs].float()               # [n_q, bs, 128]
    sc = torch.einsum("hqd,hkd->hqk", qq, kk) * 128 ** -0.5
    if causal:
        mask = torch.arange(kv_len, device=dev())[None, :] > (S + tau + torch.arange(bs, device=dev()))[:, None]
        sc = sc.masked_fill(mask[None], float("-inf"))
    ref = torch.einsum("hqk,hkd->qhd", torch.softmax(sc, dim=-1), vv).reshape(bs, n_q * 128)
    got = torch.cat([unfrag(out2[0], n_q * 128), unfrag(out2[1], n_q * 128)]).float()[:bs]
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2 ** -6 * ref.abs().max()


@pytest.mark.parametrize("n_q,n_kv,H,S,tau,bs,causal", [(32, 8, 4096, 1024, 0, 16, True), (32, 8, 4096, 1031, 7, 16, False),
                                                        (16, 4, 1024, 300, 0, 5, True), (8, 8, 512, 0, 16, 16, False),
                                                        (32, 4, 2560, 4100, 3, 9, True), (4, 1, 96, 40, 0, 16, True)])
def test_attn_head_oproj_equals_two_launches(ops, n_q, n_kv, H, S, tau, bs, causal):
    """dfl_attn_head_oproj (attention stage + o_proj/residual GEMM in one launch: o_proj workgroups hold their weight
    slice in registers and wait for the heads) against dfl_attn_head followed by dfl_gemm_resid on the same inputs:
    appended K/V bit-identical, attention rows < bs within split-order rounding, rows >= bs zero, h within bf16
    rounding of a different K-summation order, sums of squares consistent with h; the launch leaves its counters
    re-armed (run twice) and never raises its failure flag."""
    from dflash_amd.model import _rope_tables
    g = gen(S + n_q + bs + H)
    ld = (n_q + 2 * n_kv) * 128
    x = torch.randn(32, ld, generator=g).to(BF16).to(dev())
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    cos, sin = _rope_tables(128, 1e6, 16384, dev())
    rows = S + tau + bs + 8
    k0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    v0 = torch.randn(n_kv, rows, 128, generator=g).to(BF16).to(dev())
    wo = (torch.randn(H, n_q * 128, generator=g) * (n_q * 128) ** -0.5).to(BF16).to(dev())
    wop = ops.pack_weight(wo)
    h0 = torch.randn(16, H, generator=g).to(BF16).to(dev())
    kw_args = dict(xq=x[16:], q_col=0, k_col=n_q * 128, v_col=(n_q + n_kv) * 128, xc=x[:16], ck_col=n_q * 128,
                   cv_col=(n_q + n_kv) * 128, n_q=n_q, n_kv=n_kv, q_norm_w=qw, k_norm_w=kw, eps=1e-6, cos_tab=cos,
                   sin_tab=sin, scale=128 ** -0.5, causal=causal, S=S, tau=tau, bs=bs, pos0=S, max_splits=16)
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, S, tau, bs, S)
    # two launches
    k1, v1, h1 = k0.clone(), v0.clone(), h0.clone()
    out1 = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev())
    ss1 = torch.zeros(H, dtype=torch.float32, device=dev())
    ops.attn_head(**kw_args, kcache=k1, vcache=v1, ws=ops.attn_head_ws(n_q, 16, 1, dev()), out_frag=out1)
    ops.gemm_resid(wop, ops.rows_frag(out1), H, n_q * 128, h1, add_residual=True, ss_out=ss1, dyn=dyn)
    # one launch, twice
    hws = ops.attn_head_ws(n_q, 16, 1, dev())
    sync = torch.zeros(ops.ATTN_OPROJ_SYNC_WORDS, dtype=torch.int32, device=dev())
    for _ in range(2):
        k2, v2, h2 = k0.clone(), v0.clone(), h0.clone()
        out2 = torch.full((16 * n_q * 128,), float("nan"), dtype=BF16, device=dev())
        ss2 = torch.full((H,), float("nan"), dtype=torch.float32, device=dev())
        ops.attn_head_oproj(**kw_args, kcache=k2, vcache=v2, ws=hws, attn_frag=out2, wo=wop, H=H, h_io=h2, ss_out=ss2,
                            sync=sync)
        assert int(sync.abs().sum()) == 0
        assert torch.equal(k1, k2) and torch.equal(v1, v2)
        a, b = unfrag(out1, n_q * 128).float(), unfrag(out2, n_q * 128).float()
        assert torch.isfinite(b).all() and (b[bs:] == 0).all()
        assert (a[:bs] - b[:bs]).abs().max() <= 2 ** -6 * a[:bs].abs().max()
        # o_proj of the launch's OWN attention rows, fp32
        want = (h0.float() + (b @ wo.float().T).to(BF16).float()).to(BF16).float()
        d = (h2.float() - want).abs()
        assert d.max() <= 2 ** -6 * want.abs().max() and (d > 0).float().mean() < 0.05
        assert (h2.float() - h1.float())[:bs].abs().max() <= 2 ** -5 * h1.float().abs().max()
        assert torch.equal(h2[bs:], h0[bs:])
        got_ss = ss2.view(H // 16, 16).sum(0)
        assert torch.allclose(got_ss, h2.float().pow(2).sum(-1), rtol=1e-4)


@pytest.mark.parametrize("S0", [200, 6100])   # 6100: two query heads per workgroup (k_attn_head_pair) in both forms
def test_attn_head_batch_and_candidates_equal_single_launches(ops, S0):
    """dfl_attn_head_batch (R requests of different lengths, lengths from their device records, one cache each) and
    dfl_attn_head_cand (C candidate blocks on ONE cached prefix, new rows to a staging area) against one
    dfl_attn_head launch per request / candidate: outputs and appended rows bit-identical (same kernel body, the
    request index only moves the base pointers)."""
    from dflash_amd.model import _rope_tables
    n_q, n_kv, R, L = 8, 2, 3, 2
    g = gen(S0)
    ld = (n_q + 2 * n_kv) * 128
    rows = S0 + 200
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16).to(dev())
    cos, sin = _rope_tables(128, 1e6, rows + 64, dev())
    xq = torch.randn(4, 16, ld, generator=g).to(BF16).to(dev())
    kc = torch.randn(4, L, n_kv, rows, 128, generator=g).to(BF16).to(dev())
    vc = torch.randn(4, L, n_kv, rows, 128, generator=g).to(BF16).to(dev())
    lens = [S0, S0 + 77, S0 - 150]
    bss = [16, 9, 13]
    common = dict(q_col=0, k_col=n_q * 128, v_col=(n_q + n_kv) * 128, n_q=n_q, n_kv=n_kv, q_norm_w=qw, k_norm_w=kw, eps=1e-6,
                  cos_tab=cos, sin_tab=sin, scale=128 ** -0.5)
    # ---- ragged batch
    dyn = torch.zeros(4, 8, dtype=torch.int32, device=dev())
    for r in range(R):
        ops.set_dyn(dyn[r], lens[r], 0, bss[r], lens[r])
    kb, vb = kc.clone(), vc.clone()
    outb = torch.zeros(4, 16 * n_q * 128, dtype=BF16, device=dev())
    ops.attn_head_batch(xq=xq, R=R, kcache=kb, vcache=vb, layer=1, causal=True, dyn=dyn, kv_len_max=max(lens) + 16,
                        ws=ops.attn_head_batch_ws(4, n_q, 16, dev()), max_splits=16, out_frag=outb, **common)
    for r in range(R):
        k1, v1 = kc[r, 1].clone(), vc[r, 1].clone()
        out1 = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev())
        ops.attn_head(xq=xq[r], kcache=k1, vcache=v1, causal=True, S=max(lens), tau=0, bs=16, pos0=0, dyn=dyn[r],
                      ws=ops.attn_head_ws(n_q, 16, 1, dev()), max_splits=16, out_frag=out1, **common)
        n = lens[r] + bss[r]
        assert torch.equal(kb[r, 1][:, :n], k1[:, :n]) and torch.equal(vb[r, 1][:, :n], v1[:, :n]), r
        assert torch.equal(unfrag(outb[r], n_q * 128)[:bss[r]], unfrag(out1, n_q * 128)[:bss[r]]), r
    assert torch.equal(kb[:, 0], kc[:, 0])            # the other layer's cache is untouched
    # ---- candidates on one cached prefix
    C, S, bs = 4, lens[0], 16
    k_out = torch.zeros(C, n_kv, 16, 128, dtype=BF16, device=dev())
    v_out = torch.zeros(C, n_kv, 16, 128, dtype=BF16, device=dev())
    outc = torch.zeros(C, 16 * n_q * 128, dtype=BF16, device=dev())
    k0, v0 = kc[0, 0].clone(), vc[0, 0].clone()
    ops.attn_head_cand(xq=xq, kcache=k0, vcache=v0, S=S, bs=bs, ws=ops.attn_head_batch_ws(C, n_q, 16, dev()), max_splits=16,
                       out_frag=outc, k_out=k_out, v_out=v_out, **common)
    assert torch.equal(k0, kc[0, 0]) and torch.equal(v0, vc[0, 0])      # the shared cache is read-only here
    for c in range(C):
        k1, v1 = kc[0, 0].clone(), vc[0, 0].clone()
        out1 = torch.zeros(16 * n_q * 128, dtype=BF16, device=dev())
        ops.attn_head(xq=xq[c], kcache=k1, vcache=v1, causal=True, S=S, tau=0, bs=bs, pos0=S,
                      ws=ops.attn_head_ws(n_q, 16, 1, dev()), max_splits=16, out_frag=out1, **common)
        assert torch.equal(k_out[c], k1[:, S:S + 16]) and torch.equal(v_out[c], v1[:, S:S + 16]), c
        assert torch.equal(outc[c], out1), c


# ------------------------------------------------------------------ integer side, golden
def test_argmax_golden(ops):
    z = np.load(os.path.join(H.GOLDEN, "argmax.npz"))
    lb = torch.from_numpy(z["logits_bf16"]).to(BF16).to(dev())
    assert np.array_equal(ops.argmax(lb).cpu().numpy(), z["ids_bf16"])
    lf = torch.from_numpy(z["logits_f32"]).to(dev())
    assert np.array_equal(ops.argmax(lf).cpu().numpy(), z["ids_f32"])
    big = torch.randn(3, 151936, generator=gen(1)).to(BF16).to(dev())
    assert torch.equal(ops.argmax(big), torch.argmax(big, dim=-1))


def test_accept_commit_golden(ops):
    for c in json.load(open(os.path.join(H.GOLDEN, "accept.json"))):
        out = torch.full((len(c["out"]),), 9999, dtype=torch.long, device=dev())
        dyn = torch.zeros(8, dtype=torch.int32, device=dev())
        res = torch.zeros(4, dtype=torch.int32, device=dev())
        ops.set_dyn(dyn, 0, 0, c["bs"], c["start"])
        ops.accept_commit(torch.tensor(c["block"], device=dev()), torch.tensor(c["posterior"], device=dev()),
                          c["bs"], out, dyn, None, res)
        assert res.tolist()[:3] == [c["acc"], c["new_start"], 0]
        assert out.tolist() == c["out"]
        d = dyn.tolist()
        assert (d[0], d[1], d[3], d[4]) == (c["start"], c["acc"] + 1, c["start"], c["new_start"])


def test_accept_commit_rearm_and_pinned_result_golden(ops):
    """The reference's accept cases (G3) through dfl_accept_commit_rearm with the result in PINNED host memory: same
    acceptance, commit and bookkeeping; the block buffer then holds the next cycle's block (the token committed at the
    new start followed by mask ids, model/dflash.py:235) — re-armed in place, wider than the block just accepted."""
    MASK = 151669
    for c in json.load(open(os.path.join(H.GOLDEN, "accept.json"))):
        out = torch.full((len(c["out"]),), 9999, dtype=torch.long, device=dev())
        dyn = torch.zeros(8, dtype=torch.int32, device=dev())
        res = torch.full((4,), -7, dtype=torch.int32).pin_memory()
        ops.set_dyn(dyn, 0, 0, c["bs"], c["start"])
        blk = torch.full((32,), 7777, dtype=torch.long, device=dev())
        blk[:c["bs"]] = torch.tensor(c["block"], device=dev())
        ops.accept_commit(blk, torch.tensor(c["posterior"], device=dev()), c["bs"], out, dyn, None, res,
                          rearm=(blk, 32, MASK))
        torch.cuda.synchronize()
        assert res.tolist()[:3] == [c["acc"], c["new_start"], 0]
        assert out.tolist() == c["out"]
        assert blk.tolist() == [c["posterior"][c["acc"]]] + [MASK] * 31
        if c["new_start"] < len(c["out"]):
            assert blk[0] == out[c["new_start"]]
    with pytest.raises(RuntimeError):      # a host-side result buffer must be pinned
        ops.accept_commit(blk, blk, 4, out, dyn, None, torch.zeros(4, dtype=torch.int32))


def test_accept_commit_stop_flag(ops):
    block = torch.tensor([5, 6, 7, 8], device=dev())
    post = torch.tensor([6, 7, 99, 3], device=dev())   # acc = 2, bonus token 99
    for stops, want in (([99], 1), ([8], 0), ([7, 1000], 1), ([3], 0)):
        out = torch.zeros(20, dtype=torch.long, device=dev())
        dyn = torch.zeros(8, dtype=torch.int32, device=dev())
        res = torch.zeros(4, dtype=torch.int32, device=dev())
        ops.set_dyn(dyn, 0, 0, 4, 10)
        ops.accept_commit(block, post, 4, out, dyn, torch.tensor(stops, device=dev()), res)
        assert res.tolist()[:3] == [2, 13, want], stops
        assert out.tolist()[10:14] == [5, 6, 7, 99]


def test_sample_temperature_posterior(ops):
    """T>0 (BASELINE config 4): the posterior draw stays torch.multinomial; given the
    reference's sampled ids the acceptance kernel reproduces its length."""
    z = np.load(os.path.join(H.GOLDEN, "sample_t.npz"))
    out = torch.zeros(64, dtype=torch.long, device=dev())
    dyn = torch.zeros(8, dtype=torch.int32, device=dev())
    res = torch.zeros(4, dtype=torch.int32, device=dev())
    ops.set_dyn(dyn, 0, 0, 16, 3)
    ops.accept_commit(torch.from_numpy(z["block"][0]).to(dev()), torch.from_numpy(z["ids"][0]).to(dev()), 16, out,
                      dyn, None, res)
    assert res.tolist()[0] == int(z["acc"])


def test_rejects_bad_arguments(ops):
    from dflash_amd._lib import DFlashHipError
    w = torch.zeros(20, 512, dtype=BF16, device=dev())
    with pytest.raises(DFlashHipError):
        ops.pack_weight(w)                       # N % 16 != 0
    with pytest.raises(RuntimeError):
        ops.argmax(torch.zeros(2, 8))            # CPU tensor: no host path
