"""Target prefill on the kernels (csrc/prefill.hip, NativeTarget.prefill; reference call site model/dflash.py:218-225:
target(input_ids, position_ids, past_key_values, use_cache=True, logits_to_keep=1, output_hidden_states=True)):
the prompt-length MFMA GEMM and its epilogues against torch, the row stages against the oracle's leaf ops, and the
whole prefill — K/V of every row, the last row's logits, the tapped hidden states — against the HF forward at tiny and
at BASELINE's Qwen3-8B shapes; keep_hf = False (one copy of the target in memory) end to end."""
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def dev():
    return torch.device("cuda", 0)


def _unpack_tiles(xf, P, K):
    """frag16 row tiles [Pp/16][K/32][64][8] -> rows [P][K]: lane = (k % 32 // 8) * 16 + m."""
    Pp = xf.numel() // K
    t = xf.view(Pp // 16, K // 32, 4, 16, 8)            # [tile][k-step][kq][m][j]
    return t.permute(0, 3, 1, 2, 4).reshape(Pp, K)[:P]


@pytest.mark.parametrize("P,N,K", [(200, 256, 128), (128, 384, 320), (1, 128, 64), (300, 1280, 512)])
def test_prefill_gemm_epilogues_match_torch(P, N, K):
    """dfl_prefill_gemm_rows / _resid / _silu vs fp32 torch on the same bf16 operands; an exact small-integer case
    first (any wrong fragment, tile or k-step index shows as a wrong integer; asymmetric operands)."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(P + N + K)
    Pp = ops.prefill_rows_padded(P)
    for exact in (True, False):
        if exact:
            W = torch.randint(-3, 4, (N, K), generator=g).float()
            X = torch.randint(-2, 3, (P, K), generator=g).float()
        else:
            W = torch.randn(N, K, generator=g) * 0.05
            X = torch.randn(P, K, generator=g)
        Wb, Xb = W.to(BF16).to(dev()), X.to(BF16).to(dev())
        wp = ops.pack_weight(Wb)
        xf = torch.full((Pp * K,), float("nan"), dtype=BF16, device=dev())
        ops.prefill_norm_pack(Xb, P, K, None, 1e-6, xf)
        assert torch.equal(_unpack_tiles(xf, P, K), Xb)                       # pack only: the rows themselves
        assert not torch.isnan(xf.float()).any()                              # padded tiles are zero fragments
        ref = Xb.float() @ Wb.float().T
        out = torch.full((Pp, N), 7.0, dtype=BF16, device=dev())
        ops.prefill_gemm_rows(wp, xf, P, N, K, out)
        assert torch.equal(out[P:], torch.full((Pp - P, N), 7.0, dtype=BF16, device=dev()))   # rows >= P untouched
        if exact:
            assert torch.equal(out[:P], ref.to(BF16)), "integer GEMM differs"    # (exact fp32 sums, one bf16 rounding)
        else:
            H.assert_close(f"prefill gemm rows {P}x{N}x{K}", out[:P], ref.to(BF16), max_rel=1e-2, mean_rel=1e-3)
        # residual add + tap copy (bf16 + bf16 -> bf16, model/dflash.py:140,144)
        h0 = (torch.randn(Pp, N, generator=g)).to(BF16).to(dev()) if not exact else torch.randint(-5, 6, (Pp, N), generator=g).to(BF16).to(dev())
        h = h0.clone()
        tap = torch.zeros(P, N, dtype=BF16, device=dev())
        ops.prefill_gemm_resid(wp, xf, P, N, K, h, tap=tap)
        want = (h0[:P].float() + ref.to(BF16).float()).to(BF16)
        if exact:
            assert torch.equal(h[:P], want) and torch.equal(tap, want)
        else:
            H.assert_close(f"prefill gemm resid {P}x{N}x{K}", h[:P], want, max_rel=1e-2, mean_rel=1e-3)
            assert torch.equal(tap, h[:P])
        assert torch.equal(h[P:], h0[P:])
        # SiLU(gate) * up over the interleaved gate/up weight -> frag16 tiles of I columns
        I = N // 2
        if I % 64 == 0:
            gp = ops.pack_weight_gateup(Wb[:I].contiguous(), Wb[I:].contiguous())
            act = torch.full((Pp * I,), float("nan"), dtype=BF16, device=dev())
            ops.prefill_gemm_silu(gp, xf, P, I, K, act)
            gate, up = ref[:, :I].to(BF16), ref[:, I:].to(BF16)
            a_ref = (torch.nn.functional.silu(gate.float()).to(BF16).float() * up.float()).to(BF16)
            got = _unpack_tiles(act, P, I)
            if exact:
                assert torch.equal(got, a_ref)
            else:
                H.assert_close(f"prefill gemm silu {P}x{I}x{K}", got, a_ref, max_rel=2e-2, mean_rel=2e-3)


def test_prefill_norm_and_rope_match_the_oracle():
    """dfl_prefill_norm_pack vs the oracle's Qwen3RMSNorm; dfl_prefill_qk_rope vs the oracle's per-head norm + rotary
    embedding (HF's apply_rotary_pos_emb on q and k alike): same rounding points, bit-identical but for rare last-bit flips."""
    from dflash_amd import ops
    from dflash_amd.model import _rope_tables
    from oracle import dflash_oracle as O
    g = torch.Generator().manual_seed(3)
    P, Hd = 77, 512
    h = torch.randn(P, Hd, generator=g).to(BF16)
    w = (1 + 0.1 * torch.randn(Hd, generator=g)).to(BF16)
    Pp = ops.prefill_rows_padded(P)
    xf = torch.empty(Pp * Hd, dtype=BF16, device=dev())
    hp = torch.zeros(Pp, Hd, dtype=BF16, device=dev())
    hp[:P] = h.to(dev())
    ops.prefill_norm_pack(hp, P, Hd, w.to(dev()), 1e-6, xf)
    assert torch.equal(_unpack_tiles(xf, P, Hd).cpu(), O.rms_norm(h, w, 1e-6))
    n_q, n_kv = 4, 2
    nqkv = (n_q + 2 * n_kv) * 128
    qkv = torch.randn(P, nqkv, generator=g).to(BF16)
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16)
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16)
    cos, sin = _rope_tables(128, 1e6, 256, dev())
    buf = torch.zeros(Pp, nqkv, dtype=BF16, device=dev())
    buf[:P] = qkv.to(dev())
    kc = torch.zeros(n_kv, 128, 128, dtype=BF16, device=dev())
    vc = torch.zeros_like(kc)
    pos0, row0 = 5, 9
    ops.prefill_qk_rope(buf, P, 0, n_q * 128, (n_q + n_kv) * 128, n_q, n_kv, qw.to(dev()), kw.to(dev()), 1e-6, cos, sin,
                        pos0, kc, vc, row0)
    q = O.rms_norm(qkv[:, :n_q * 128].view(P, n_q, 128), qw, 1e-6).transpose(0, 1)[None]          # [1, n_q, P, 128]
    k = O.rms_norm(qkv[:, n_q * 128:(n_q + n_kv) * 128].view(P, n_kv, 128), kw, 1e-6).transpose(0, 1)[None]
    c = torch.cat([cos, cos], -1).cpu()[pos0:pos0 + P][None]
    s = torch.cat([sin, sin], -1).cpu()[pos0:pos0 + P][None]
    qe = (q * c[:, None]) + (O.rotate_half(q) * s[:, None])
    ke = (k * c[:, None]) + (O.rotate_half(k) * s[:, None])

    def same_but_for_rare_ulps(name, got, ref):     # the wave's sum of squares adds in another order than torch's mean:
        bad = got != ref                            # rstd may differ in its last fp32 bit and flip a bf16 rounding
        assert float(bad.float().mean()) < 2e-3, (name, int(bad.sum()))
        H.assert_close(name, got, ref, max_rel=4e-3, mean_rel=1e-5)

    same_but_for_rare_ulps("prefill q rope", buf[:P, :n_q * 128].view(P, n_q, 128).cpu(), qe[0].transpose(0, 1))
    same_but_for_rare_ulps("prefill k rope", kc[:, row0:row0 + P].cpu(), ke[0])
    assert torch.equal(vc[:, row0:row0 + P].cpu(), qkv[:, (n_q + n_kv) * 128:].view(P, n_kv, 128).transpose(0, 1))
    assert int(kc[:, :row0].abs().sum()) == 0 and int(kc[:, row0 + P:].abs().sum()) == 0


def _prefill_vs_hf(name, hf, V, Hd, P, taps, seed=4):
    from transformers import DynamicCache
    from dflash_amd import NativeTarget
    nt = NativeTarget(hf)
    assert nt.native_prefill
    g = torch.Generator().manual_seed(seed)
    prompt = torch.randint(0, V - 100, (1, P), generator=g).to(dev())
    cache = nt.new_cache(P + 64)
    out = nt.prefill(prompt, cache, output_hidden_states=True, tap_layers=taps)
    rc = DynamicCache()
    with torch.inference_mode():
        ref = hf(prompt, position_ids=torch.arange(P, device=dev())[None], past_key_values=rc, use_cache=True,
                 logits_to_keep=1, output_hidden_states=True)
    assert out.logits.shape == ref.logits.shape == (1, 1, V) and cache.get_seq_length() == P
    H.assert_close(f"{name} prefill logits (last row)", out.logits[0], ref.logits[0])
    for l in taps:
        H.assert_close(f"{name} prefill tap layer {l}", out.hidden_states[l + 1][0], ref.hidden_states[l + 1][0])
    L = hf.config.num_hidden_layers
    for li in (0, L - 1):
        H.assert_close(f"{name} prefill K layer {li}", cache.k[li][:, :P], rc.layers[li].keys[0], max_rel=H.KV_MAX_REL)
        H.assert_close(f"{name} prefill V layer {li}", cache.v[li][:, :P], rc.layers[li].values[0], max_rel=H.KV_MAX_REL)
    missing = [i for i in range(1, L) if (i - 1) not in taps]
    if missing:                           # states that were not asked for are not kept
        with pytest.raises(KeyError):
            out.hidden_states[missing[0]]
    return nt, prompt, cache, out, ref


@pytest.mark.parametrize("P", [45, 128, 300])
def test_native_prefill_matches_hf_forward_tiny(P):
    from dflash_amd.synthetic import make_hf_qwen3
    torch.manual_seed(11)
    hf = make_hf_qwen3({**H.TINY_TARGET, "num_layers": 6}, dev())
    _prefill_vs_hf(f"tiny P={P}", hf, 2048, 512, P, taps=[1, 3])


def test_native_prefill_full_size_qwen3_8b_and_verify_on_its_cache():
    """BASELINE configs[1] target shapes at 3 layers, a 1024-row prompt (the bench's prefix): prefill vs the HF forward,
    then a verify block ON THE CACHE THE NATIVE PREFILL WROTE vs the HF forward continuing its own cache."""
    from dflash_amd.config import QWEN3_8B_TARGET
    from dflash_amd.synthetic import make_hf_qwen3
    torch.manual_seed(21)
    hf = make_hf_qwen3({**QWEN3_8B_TARGET, "num_layers": 3}, dev())
    nt, prompt, cache, out, _ = _prefill_vs_hf("Qwen3-8B shapes P=1024", hf, 151936, 4096, 1024, taps=[0, 1])
    from transformers import DynamicCache
    g = torch.Generator().manual_seed(9)
    block = torch.randint(0, 151000, (1, 16), generator=g).to(dev())
    logits = torch.zeros(16, 151936, dtype=BF16, device=dev())
    nt.verify(block[0], 1024, cache, tap_layers=[0, 1], logits_out=logits)
    rc = DynamicCache()
    with torch.inference_mode():
        hf(prompt, past_key_values=rc, use_cache=True)
        ref = hf(block, position_ids=torch.arange(1024, 1040, device=dev())[None], past_key_values=rc, use_cache=True)
    H.assert_close("verify after native prefill: logits", logits, ref.logits[0])


def test_llama_style_target_prefill():
    """No q/k norm, llama3 RoPE scaling (BASELINE configs[3] class): the rotary tables come from the wrapped model's own
    rotary module."""
    tf = pytest.importorskip("transformers")
    cfg = tf.LlamaConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
                         num_key_value_heads=2, head_dim=128, max_position_embeddings=8192, rms_norm_eps=1e-5,
                         tie_word_embeddings=False,
                         rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0, "low_freq_factor": 1.0,
                                          "high_freq_factor": 4.0, "original_max_position_embeddings": 2048})
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(5)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(BF16)
    try:
        with torch.device(dev()):
            hf = tf.LlamaForCausalLM(cfg).eval()
    finally:
        torch.set_default_dtype(prev)
    _prefill_vs_hf("Llama-style", hf, 2048, 512, 90, taps=[0])


def test_keep_hf_false_one_copy_end_to_end():
    """NativeTarget(keep_hf=False): the wrapped model is gone after packing — prefill and verify run on the packed
    weights alone, dflash_generate commits the target's greedy walk, calling the object like the HF model raises."""
    from dflash_amd import NativeTarget, dflash_generate
    from dflash_amd.synthetic import greedy_walk, impose_greedy_walk, make_hf_qwen3
    import test_hip_model as TM
    cfg = H.tiny_cfg()
    m = TM.make_model(cfg)
    torch.manual_seed(11)
    hf = make_hf_qwen3({**H.TINY_TARGET, "num_layers": 6}, dev())
    perm = impose_greedy_walk(hf, seed=5)
    nt = NativeTarget(hf, keep_hf=False)
    del hf
    assert nt.hf is None
    with pytest.raises(RuntimeError):
        nt(torch.zeros(1, 4, dtype=torch.long, device=dev()))
    prompt = torch.randint(0, 2000, (1, 150), generator=torch.Generator().manual_seed(4)).to(dev())
    n_new = 70
    G = greedy_walk(perm, prompt, n_new + 40).to(dev())
    plan = H.make_plan(64, 16, 17)

    def hook(blk, start, call):
        k = min(plan[call], blk.shape[1] - 1)
        blk[0, 1:k + 1] = G[start + 1:start + k + 1]
        if k + 1 < blk.shape[1]:
            w = G[start + k + 1]
            blk[0, k + 1] = torch.where(blk[0, k + 1] == w, (w + 1) % 2000, blk[0, k + 1])

    r = dflash_generate(m, nt, prompt, cfg.mask_token_id, n_new, 16, None, 0.0, draft_token_hook=hook)
    assert r.output_ids[0].tolist() == G[:150 + n_new].tolist()
    assert max(r.acceptance_lengths) > 8


def test_hf_prefill_is_still_selectable():
    """prefill="hf" (and MoE targets): the wrapped model's forward fills the cache, as in rounds 1-2."""
    from dflash_amd import NativeTarget
    from dflash_amd.synthetic import make_hf_qwen3
    torch.manual_seed(11)
    hf = make_hf_qwen3({**H.TINY_TARGET, "num_layers": 3}, dev())
    a, b = NativeTarget(hf, prefill="hf"), NativeTarget(hf)
    assert not a.native_prefill and b.native_prefill
    prompt = torch.randint(0, 2000, (1, 60), generator=torch.Generator().manual_seed(1)).to(dev())
    ca, cb = a.new_cache(128), b.new_cache(128)
    oa, ob = a.prefill(prompt, ca), b.prefill(prompt, cb)
    H.assert_close("hf vs native prefill logits", ob.logits[0], oa.logits[0])
    H.assert_close("hf vs native prefill K", cb.k[2][:, :60], ca.k[2][:, :60], max_rel=H.KV_MAX_REL)
    with pytest.raises(NotImplementedError):
        NativeTarget(hf, prefill="hf", keep_hf=False)


@pytest.mark.parametrize("P,n_q,n_kv", [(45, 4, 2), (300, 4, 4), (1024, 8, 2), (17, 2, 1)])
def test_prefill_attention_matches_torch(P, n_q, n_kv):
    """dfl_prefill_attn vs fp32 torch attention (causal, GQA) on random bf16 q / K / V rows: every query row, every head;
    stale cache rows beyond P (NaN-poisoned here) must not leak in."""
    from dflash_amd import ops
    g = torch.Generator().manual_seed(P + n_q)
    Pp = ops.prefill_rows_padded(P)
    nqkv = (n_q + 2 * n_kv) * 128
    qkv = torch.zeros(Pp, nqkv, dtype=BF16, device=dev())
    qkv[:P, :n_q * 128] = torch.randn(P, n_q * 128, generator=g).to(BF16).to(dev())
    kc = torch.full((n_kv, P + 40, 128), float("nan"), dtype=BF16, device=dev())
    vc = torch.full_like(kc, float("nan"))
    kc[:, :P] = torch.randn(n_kv, P, 128, generator=g).to(BF16).to(dev())
    vc[:, :P] = torch.randn(n_kv, P, 128, generator=g).to(BF16).to(dev())
    xf = torch.zeros(Pp * n_q * 128, dtype=BF16, device=dev())
    ops.prefill_attn(qkv, P, 0, kc, vc, n_q, n_kv, 128 ** -0.5, xf)
    got = _unpack_tiles(xf, P, n_q * 128).float().view(P, n_q, 128)
    q = qkv[:P, :n_q * 128].float().view(P, n_q, 128).transpose(0, 1)                       # [n_q, P, 128]
    k = kc[:, :P].float().repeat_interleave(n_q // n_kv, dim=0)
    v = vc[:, :P].float().repeat_interleave(n_q // n_kv, dim=0)
    sc = (q @ k.transpose(1, 2)) * 128 ** -0.5
    sc = sc.masked_fill(torch.triu(torch.ones(P, P, dtype=torch.bool, device=dev()), 1), float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ v).transpose(0, 1)                                  # [P, n_q, 128]
    assert not torch.isnan(got).any()
    H.assert_close(f"prefill attention P={P} heads {n_q}/{n_kv}", got, ref, max_rel=1e-2, mean_rel=1e-3)


def test_prefill_sdpa_core_is_the_same_prefill():
    """prefill_attn = "sdpa" (torch's attention on the rows the kernels produced) against the default k_pattn: logits
    and K/V of the whole prefill agree."""
    from dflash_amd import NativeTarget
    from dflash_amd.synthetic import make_hf_qwen3
    torch.manual_seed(11)
    hf = make_hf_qwen3({**H.TINY_TARGET, "num_layers": 4}, dev())
    a, b = NativeTarget(hf), NativeTarget(hf)
    b.prefill_attn = "sdpa"
    prompt = torch.randint(0, 2000, (1, 200), generator=torch.Generator().manual_seed(1)).to(dev())
    ca, cb = a.new_cache(256), b.new_cache(256)
    oa, ob = a.prefill(prompt, ca), b.prefill(prompt, cb)
    H.assert_close("k_pattn vs sdpa prefill logits", oa.logits[0], ob.logits[0])
    H.assert_close("k_pattn vs sdpa prefill K (last layer)", ca.k[3][:, :200], cb.k[3][:, :200], max_rel=H.KV_MAX_REL)
