"""dflash_amd — MI355X-native (gfx950) hot path of DFlash block-diffusion speculative
decoding: draft block forward, greedy unmask, acceptance scan with KV rollback.

Drop-in for the reference's `model` package on that path:

    from dflash_amd import DFlashDraftModel, sample, extract_context_feature
    from dflash_amd import dflash_generate, dflash_generate_policy, EWMAPerformanceScheduler

The kernels live in `lib/libdflash_hip.so` (C ABI: include/dflash_hip.h), built by
`python -m dflash_amd.build`.  Nothing here falls back to PyTorch or the CPU.
"""
from .config import DFlashConfig
from .utils import build_target_layer_ids, extract_context_feature, sample
from .scheduler import EWMAPerformanceScheduler
from .model import DFlashDraftModel, DFlashKVCache
from .generate import dflash_generate, dflash_generate_policy
from .target import NativeTarget
from .candidates import dflash_generate_candidate_solutions

__all__ = ["DFlashConfig", "DFlashDraftModel", "DFlashKVCache", "EWMAPerformanceScheduler",
           "build_target_layer_ids", "extract_context_feature", "sample", "dflash_generate",
           "dflash_generate_policy", "NativeTarget", "dflash_generate_candidate_solutions"]
