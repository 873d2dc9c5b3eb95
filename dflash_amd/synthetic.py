"""Seeded random-init weights of a given architecture.

No checkpoints exist offline, so tests, the golden-vector
generator and bench.py all build weights from a seed.  Generation is on the CPU
generator in fp32 then cast, so the same seed yields bit-identical weights in
the build container and on the GPU box.
"""
from __future__ import annotations

import torch

from .config import DFlashConfig


def make_draft_state_dict(cfg: DFlashConfig, seed: int = 0, dtype=torch.bfloat16, std: float = 0.02,
                          device="cpu", norm_jitter: float = 0.1) -> dict:
    """State dict with the reference's key names.  Linear weights
    ~ N(0, std) (HF default init), norm weights ~ 1 + norm_jitter * N(0,1) so a
    norm-weight bug cannot hide behind all-ones."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in cfg.state_dict_shapes().items():
        if len(shape) == 1:
            t = 1.0 + norm_jitter * torch.randn(shape, generator=g)
        else:
            # big tensors row-chunked: keeps peak host memory low for 8B-shaped drafts
            t = torch.empty(shape, dtype=dtype)
            rows = max(1, (1 << 24) // shape[1])
            for r in range(0, shape[0], rows):
                n = min(rows, shape[0] - r)
                t[r:r + n] = (torch.randn((n, shape[1]), generator=g) * std).to(dtype)
        sd[name] = t.to(dtype).to(device)
    return sd
