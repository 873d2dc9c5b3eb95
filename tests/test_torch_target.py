"""The pure-torch test target agrees with HF Qwen3ForCausalLM (third-party, in the
image) on logits and tapped hidden states, so fixtures made with one are valid for
the other."""
import pytest
import torch

import helpers as H


def test_matches_hf_qwen3():
    tf = pytest.importorskip("transformers")
    t = H.TINY_TARGET
    hf_cfg = tf.Qwen3Config(vocab_size=t["vocab_size"], hidden_size=t["hidden_size"],
                            num_hidden_layers=t["num_layers"], num_attention_heads=t["num_heads"],
                            num_key_value_heads=t["num_kv_heads"], head_dim=t["head_dim"],
                            intermediate_size=t["intermediate_size"], rms_norm_eps=1e-6,
                            rope_parameters={"rope_type": "default", "rope_theta": t["rope_theta"]},
                            tie_word_embeddings=False)
    hf_cfg._attn_implementation = "eager"
    hf = tf.Qwen3ForCausalLM(hf_cfg).eval()
    mine = H.tiny_target(dtype=torch.float32)
    mine.load_hf_state_dict(hf.state_dict())
    ids = torch.randint(0, 2000, (1, 23), generator=torch.Generator().manual_seed(0))
    with torch.inference_mode():
        cache = tf.DynamicCache()
        a = hf(ids[:, :15], past_key_values=cache, use_cache=True, output_hidden_states=True)
        a2 = hf(ids[:, 15:], past_key_values=cache, use_cache=True, output_hidden_states=True,
                position_ids=torch.arange(15, 23)[None])
        c2 = mine.new_cache()
        b = mine(ids[:, :15], past_key_values=c2, output_hidden_states=True)
        b2 = mine(ids[:, 15:], past_key_values=c2, output_hidden_states=True,
                  position_ids=torch.arange(15, 23)[None])
    for x, y in ((a, b), (a2, b2)):
        assert torch.allclose(x.logits, y.logits, atol=2e-5, rtol=1e-4)
        assert len(x.hidden_states) == len(y.hidden_states)
        for hx, hy in zip(x.hidden_states, y.hidden_states):
            assert torch.allclose(hx, hy, atol=2e-5, rtol=1e-4)
