"""Micro-benchmark of the batched (ragged multi-request) skinny GEMMs: device time per launch
for R = 2 and 4 request tiles, weights rotated through > 256 MiB so no launch is served from
the Infinity Cache; the single-request kernel on the same shape beside it.
Run on the GPU box: python scripts/bench_gemm_batch.py [names...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dflash_amd import ops
from bench_gemm import time_launches

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16


def weights(N, K):
    nbytes = N * K * 2
    n_buf = max(4, int(600e6 // nbytes) + 1)
    return [torch.randn(N * K // 2, device=dev, dtype=torch.float32).view(BF16)[:N * K].contiguous() for _ in range(n_buf)]


def dyn_for(MT, R):
    d = torch.zeros(MT, 8, dtype=torch.int32)
    d[:R, 1], d[:R, 2] = 16, 16
    return d.to(dev)


def run(name, kind, N, K):
    wps = weights(N if kind != "silu" else 2 * N, K)
    nbytes = wps[0].numel() * 2
    line = f"{name:8s} {kind:6s} N={N:6d} K={K:6d} {nbytes / 1e6:7.1f} MB:"
    # single-request reference
    x1 = torch.randn(16, K, device=dev).to(BF16)
    d1 = torch.zeros(8, dtype=torch.int32, device=dev)
    ops.set_dyn(d1, 0, 16, 16, 0)
    s1 = ops.rows_plain(x1, ops.DYN_BS)
    if os.environ.get("SRC") == "frag" and kind in ("resid", "rows"):   # o_proj / down_proj read fragments in the cycle
        s1 = ops.rows_frag(x1.reshape(-1).contiguous())
    if kind == "f32":
        ks = ops.pick_ksplit(N, K, 1)
        out = torch.empty(ks * 16 * N, device=dev, dtype=torch.float32)
        us = time_launches(lambda i: ops.gemm_f32(wps[i], s1, None, 1, N, K, ks, out, d1), len(wps))
    elif kind == "resid":
        h = torch.zeros(16, N, device=dev, dtype=BF16)
        ss = torch.zeros(N, device=dev)
        us = time_launches(lambda i: ops.gemm_resid(wps[i], s1, N, K, h, add_residual=True, ss_out=ss, dyn=d1), len(wps))
    elif kind == "rows":     # finished bf16 Linear outputs (the q/k/v rows the attention stage reads)
        h = torch.zeros(16, N, device=dev, dtype=BF16)
        us = time_launches(lambda i: ops.gemm_resid(wps[i], s1, N, K, h, add_residual=False, dyn=d1), len(wps))
    elif kind == "silu":
        act = torch.empty(16 * N, device=dev, dtype=BF16)
        us = time_launches(lambda i: ops.gemm_silu_mul(wps[i], s1, N, K, act, d1), len(wps))
    else:
        ws1 = ops.argmax_ws(dev)
        ids = torch.zeros(16, dtype=torch.int64, device=dev)
        us = time_launches(lambda i: ops.gemm_argmax(wps[i], s1, N, K, 0, 16, ws1, ids, 0, dyn=d1), len(wps))
    line += f"  single {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s |"
    for R in (2, 4):
        MT = ops.batch_tiles(R)
        x = torch.randn(MT, 16, K, device=dev).to(BF16)
        dyn = dyn_for(MT, R)
        src = ops.brows_plain(x, ops.DYN_BS)
        if os.environ.get("SRC") == "frag":
            src = ops.brows_frag(x.reshape(MT, 16 * K).contiguous())
        ws = ops.gemm_batch_ws(2 * N if kind == "silu" else N, K, dev)
        if kind == "f32":
            out = torch.empty(ops.batch_ksplit(K) * MT * 16 * N, device=dev, dtype=torch.float32)
            us = time_launches(lambda i: ops.gemm_f32_batch(wps[i], src, R, N, K, out, dyn), len(wps))
        elif kind == "resid":
            h = torch.zeros(MT, 16, N, device=dev, dtype=BF16)
            ss = torch.zeros(MT, N, device=dev)
            us = time_launches(lambda i: ops.gemm_resid_batch(wps[i], src, R, N, K, h, add_residual=True, ws=ws, dyn=dyn,
                                                              ss_out=ss), len(wps))
        elif kind == "rows":
            h = torch.zeros(MT, 16, N, device=dev, dtype=BF16)
            us = time_launches(lambda i: ops.gemm_resid_batch(wps[i], src, R, N, K, h, add_residual=False, ws=ws, dyn=dyn),
                               len(wps))
        elif kind == "silu":
            act = torch.empty(MT, 16 * N, device=dev, dtype=BF16)
            us = time_launches(lambda i: ops.gemm_silu_mul_batch(wps[i], src, R, N, K, act, ws, dyn), len(wps))
        else:
            ids = torch.zeros(MT, 16, dtype=torch.int64, device=dev)
            us = time_launches(lambda i: ops.gemm_argmax_batch(wps[i], src, R, N, K, 0, 16, ws, ids, 0, dyn,
                                                               nrows_dyn_word=ops.DYN_BS), len(wps))
        line += f"  R={R} {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s"
    print(line, flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    shapes = [("qkv", "f32", 6144, 4096), ("qkvr", "rows", 6144, 4096), ("or", "rows", 4096, 4096), ("o", "resid", 4096, 4096), ("gateup", "silu", 12288, 4096),
              ("down", "resid", 4096, 12288), ("downf", "f32", 4096, 12288), ("of", "f32", 4096, 4096), ("fc", "resid", 4096, 20480), ("kv_all", "f32", 10240, 4096),
              ("lm_head", "argmax", 151936, 4096)]
    for s in shapes:
        if "all" in which or s[0] in which:
            run(*s)
