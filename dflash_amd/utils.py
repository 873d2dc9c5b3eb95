"""Host-side helpers of the hot path with the reference's names and meaning
(model/utils.py:4-34).  `sample` at temperature 0 runs the wavefront argmax
kernel through the C-ABI; there is no CPU fallback for device tensors."""
from __future__ import annotations

from typing import Optional, Sequence

import torch


def build_target_layer_ids(num_target_layers: int, num_draft_layers: int) -> list[int]:
    """Evenly spaced target layers to tap, in [1, N-3] (model/utils.py:4-14).
    Python `round` (half-to-even) on purpose: that is what the checkpoints were
    trained with."""
    if num_draft_layers == 1:
        return [num_target_layers // 2]
    first, last = 1, num_target_layers - 3
    step = (last - first) / (num_draft_layers - 1)
    return [int(round(first + i * step)) for i in range(num_draft_layers)]


def extract_context_feature(hidden_states: Sequence[torch.Tensor], layer_ids: Optional[Sequence[int]]) -> torch.Tensor:
    """Concatenate the tapped target layers' outputs on the feature axis
    (model/utils.py:16-25): hidden_states[0] is the embedding output, so layer l's
    output is entry l + 1."""
    return torch.cat([hidden_states[l + 1] for l in layer_ids], dim=-1)


def sample(logits: torch.Tensor, temperature: float = 0.0) -> torch.Tensor:
    """model/utils.py:27-34.  logits [B, T, V] -> ids int64 [B, T].

    T < 1e-5: first-max-index argmax, on the device through `dfl_argmax`.
    Otherwise the reference's own softmax + `torch.multinomial` draw is kept so
    the RNG stream is the caller's torch generator."""
    if temperature < 1e-5:
        if not logits.is_cuda:
            raise RuntimeError("dflash_amd.sample: logits must live on the GPU (no CPU path in the product)")
        from . import ops
        return ops.argmax(logits)
    b, t, v = logits.shape
    probs = torch.softmax(logits.view(-1, v) / temperature, dim=-1)
    return torch.multinomial(probs, num_samples=1).view(b, t)
