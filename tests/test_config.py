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
