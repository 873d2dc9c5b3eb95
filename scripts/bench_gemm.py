"""Micro-benchmark of the skinny GEMM over (shape, ksplit): device time per launch with
the weights rotated through > 256 MiB of distinct buffers so no launch is served from
the Infinity Cache.  Run on the GPU box: python scripts/bench_gemm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dflash_amd import ops

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16


def time_launches(fn, n_buf, iters=5):
    torch.cuda.synchronize()
    for i in range(n_buf):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(n_buf):
            fn(i)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n_buf)
    return best * 1e3  # us


def run(name, N, K, mt, ksplits):
    nbytes = N * K * 2
    n_buf = max(4, int(600e6 // nbytes) + 1)
    wps = [torch.randn(N * K // 2, device=dev, dtype=torch.float32).view(BF16)[:N * K].contiguous() for _ in range(n_buf)]
    x0 = torch.randn(16 * K, device=dev).to(BF16)
    x1 = torch.randn(16 * K, device=dev).to(BF16)
    for ks in ksplits:
        if ks < ops.min_ksplit(K, mt):
            continue
        out = torch.empty(ks * mt * 16 * N, device=dev, dtype=torch.float32)
        us = time_launches(lambda i: ops.gemm_f32(wps[i], x0, x1 if mt == 2 else None, mt, N, K, ks, out), n_buf)
        print(f"{name:8s} N={N:6d} K={K:6d} mt={mt} ksplit={ks:2d}: {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s"
              f"{'  <- picked' if ks == ops.pick_ksplit(N, K, mt) else ''}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    shapes = [("fc", 4096, 20480, 1, [5, 6, 8, 10, 16]), ("qkv_d", 6144, 4096, 2, [2, 3, 4, 6, 8]),
              ("qkv_t", 6144, 4096, 1, [1, 2, 3, 4, 6]), ("o", 4096, 4096, 1, [1, 2, 4, 8]),
              ("down", 4096, 12288, 1, [3, 4, 6, 8, 12]), ("kv", 2048, 4096, 1, [1, 2, 4, 8])]
    for s in shapes:
        if "all" in which or s[0] in which:
            run(*s)
    if "all" in which or "silu" in which:
        I, K = 12288, 4096
        n_buf = 4
        wps = [torch.randn(I * K, device=dev, dtype=torch.float32).view(BF16)[:2 * I * K].contiguous() for _ in range(n_buf)]
        x = torch.randn(16 * K, device=dev).to(BF16)
        act = torch.empty(16 * I, device=dev, dtype=BF16)
        us = time_launches(lambda i: ops.gemm_silu_mul(wps[i], x, I, K, act), n_buf)
        print(f"silu     I={I} K={K} frag source  : {us:8.1f} us  {2 * I * K * 2 / us / 1e6:7.2f} TB/s")
        h = torch.randn(16, K, device=dev).to(BF16)
        nw = torch.ones(K, device=dev, dtype=BF16)
        dyn = torch.zeros(8, dtype=torch.int32, device=dev)
        ops.set_dyn(dyn, 0, 0, 16, 0)
        for nss in (1, 256):
            ss = torch.rand(nss * 16, device=dev) * (K / nss)
            src = ops.rows_normed(h, ss, nss, nw, 1e-6, ops.DYN_BS)
            us = time_launches(lambda i: ops.gemm_silu_mul(wps[i], src, I, K, act, dyn), n_buf)
            print(f"silu     I={I} K={K} normed nss={nss:3d}: {us:8.1f} us  {2 * I * K * 2 / us / 1e6:7.2f} TB/s")
        srcp = ops.rows_plain(h, ops.DYN_BS)
        us = time_launches(lambda i: ops.gemm_silu_mul(wps[i], srcp, I, K, act, dyn), n_buf)
        print(f"silu     I={I} K={K} plain rows   : {us:8.1f} us  {2 * I * K * 2 / us / 1e6:7.2f} TB/s")
