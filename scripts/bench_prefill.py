"""Target prefill at BASELINE's prefix (P = 1024 prompt rows, Qwen3-8B shapes, 36 layers): NativeTarget.prefill on the
kernels against the wrapped HF model's forward, same box, same weights.  Run on the GPU box: python scripts/bench_prefill.py [P] [layers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dflash_amd import NativeTarget
from dflash_amd.config import QWEN3_8B_TARGET
from dflash_amd.synthetic import make_hf_qwen3

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 36
dev = torch.device("cuda", 0)
torch.manual_seed(0)
hf = make_hf_qwen3({**QWEN3_8B_TARGET, "num_layers": L}, dev)
prompt = torch.randint(0, 151000, (1, P), generator=torch.Generator().manual_seed(1)).to(dev)
taps = [1, 9, 17, 25, 33] if L == 36 else [0]


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for mode in ("native", "hf"):
    nt = NativeTarget(hf, prefill=mode)
    cache = nt.new_cache(P + 64)
    ms = timed(lambda: nt.prefill(prompt, cache, output_hidden_states=True, tap_layers=taps))
    flops = 2.0 * P * sum(p.numel() for n, p in hf.named_parameters() if "layers" in n and p.dim() == 2)
    print(f"prefill P={P} layers={L} {mode:6s}: {ms:7.2f} ms  ({flops / ms / 1e9:.0f} TFLOP/s over the layers' GEMMs)")
    del nt, cache
