"""Draft config parsing: HF Qwen3Config object, parsed config.json dict, defaults."""
import pytest

from dflash_amd.config import DFlashConfig, QWEN3_8B_DRAFT


def test_from_dict_and_defaults():
    d = {"hidden_size": 4096, "num_hidden_layers": 5, "num_attention_heads": 32, "num_key_value_heads": 8,
         "head_dim": 128, "intermediate_size": 12288, "vocab_size": 151936, "num_target_layers": 36,
         "block_size": 16, "rope_parameters": {"rope_type": "default", "rope_theta": 1e6},
         "dflash_config": {"mask_token_id": 151669}}
    c = DFlashConfig.from_any(d)
    assert c.target_layer_ids == [1, 9, 17, 25, 33]            # model/utils.py:4-14 default
    assert (c.fc_in, c.q_dim, c.kv_dim, c.rope_theta, c.mask_token_id) == (20480, 4096, 1024, 1e6, 151669)
    n = sum(a * b if len(s) == 2 else a for s in c.state_dict_shapes().values() for a, b in [(s + (1,))[:2]])
    assert n == 1_048_626_432                                  # SURVEY.md §8 probe of the 8B-shaped draft
    assert DFlashConfig.from_any(c) is c


def test_from_hf_config_object():
    tf = pytest.importorskip("transformers")
    hf = tf.Qwen3Config(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                        head_dim=128, intermediate_size=1024, vocab_size=2048)
    hf.block_size, hf.num_target_layers = 12, 6
    hf.dflash_config = {"mask_token_id": 7, "target_layer_ids": [0, 4]}
    c = DFlashConfig.from_any(hf)
    assert (c.block_size, c.mask_token_id, c.target_layer_ids, c.head_dim) == (12, 7, [0, 4], 128)


def test_rejects_bias():
    with pytest.raises(NotImplementedError):
        DFlashConfig(**{**QWEN3_8B_DRAFT, "attention_bias": True})


def test_rejects_unsupported_architectures_before_any_launch():
    """VERDICT r3 "missing" #3: the reference takes attention_bias, sliding_window and head_dim from the config
    (model/dflash.py:36,42-56,97); the gfx950 path covers head_dim 128, unbiased projections and full attention only.
    Everything else must raise at construction — on this CPU-only box: before any device work."""
    import types
    from dflash_amd import DFlashDraftModel
    from dflash_amd.target import NativeTarget
    d = {"hidden_size": 512, "num_hidden_layers": 2, "num_attention_heads": 4, "num_key_value_heads": 2, "head_dim": 128,
         "intermediate_size": 1024, "vocab_size": 2048, "num_target_layers": 6, "block_size": 16,
         "dflash_config": {"mask_token_id": 7}}
    with pytest.raises(NotImplementedError, match="attention_bias"):
        DFlashConfig.from_any({**d, "attention_bias": True})
    with pytest.raises(NotImplementedError, match="sliding"):
        DFlashConfig.from_any({**d, "sliding_window": 4096, "layer_types": ["full_attention", "sliding_attention"]})
    DFlashConfig.from_any({**d, "sliding_window": None, "layer_types": ["full_attention"] * 2})      # Qwen3 defaults: fine
    DFlashConfig.from_any({**d, "sliding_window": 4096, "layer_types": ["full_attention"] * 2})      # window unused: fine
    with pytest.raises(NotImplementedError, match="head_dim"):
        DFlashDraftModel({**d, "head_dim": 64}, device="cuda")
    # the target side: checked on the config alone, before the wrapped model's device is even looked at
    base = dict(num_attention_heads=4, hidden_size=512, head_dim=128, attention_bias=False, mlp_bias=False,
                layer_types=["full_attention"] * 2, sliding_window=None)
    NativeTarget.check_config(types.SimpleNamespace(**base))
    for bad, pat in (({"head_dim": 64}, "head_dim"), ({"attention_bias": True}, "bias"), ({"mlp_bias": True}, "bias"),
                     ({"layer_types": ["sliding_attention", "full_attention"], "sliding_window": 1024}, "sliding")):
        with pytest.raises(NotImplementedError, match=pat):
            NativeTarget.check_config(types.SimpleNamespace(**{**base, **bad}))
    hf = types.SimpleNamespace(config=types.SimpleNamespace(**{**base, "attention_bias": True}))
    with pytest.raises(NotImplementedError, match="bias"):
        NativeTarget(hf)        # never reaches hf.lm_head / the device
