"""Probe for the "+4 % on every kernel once a second stream / a captured graph exists in the process"
(profiles/r3_graph_ab.txt).

One process, one box: a fixed train of weight-streaming launches (the gate/up GEMM of an 8B layer, weights rotated
through > 600 MB so that no launch is served from the Infinity Cache, plus the lm_head GEMM) is timed with HIP events
on the current stream in a sequence of process states:

  A  fresh process, only torch's current (null) stream
  B  a second torch.cuda.Stream() object created, nothing launched on it
  C  one tiny kernel launched on the second stream, synchronised
  D  the second stream deleted (torch keeps pool streams alive: the HSA queue stays)
  E  a one-kernel hipGraph captured (not replayed)
  F  the graph replayed once

Each state: `reps` trains, best and median per launch.  Run it several times under different environments
(GPU_MAX_HW_QUEUES=1/2/4, HIP_FORCE_DEV_KERNARG, ...) — scripts/probes/stream_slowdown.sh does.
usage: python scripts/probes/stream_slowdown.py [order]   order = letters to run, default ABCDEF
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dflash_amd import ops

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
I, K, V = 12288, 4096, 151936


def build():
    g = torch.Generator(device=dev).manual_seed(0)
    n_buf = 4
    wps = [torch.randn(I * K, device=dev, dtype=torch.float32, generator=g).view(BF16)[:2 * I * K].contiguous()
           for _ in range(n_buf)]
    lm = torch.randn(V * K // 2, device=dev, dtype=torch.float32, generator=g).view(BF16)[:V * K].contiguous()
    x = torch.randn(16 * K, device=dev, generator=g).to(BF16)
    act = torch.empty(16 * I, device=dev, dtype=BF16)
    ws = ops.argmax_ws(dev)
    ids = torch.zeros(16, dtype=torch.int64, device=dev)

    def train():
        for i in range(n_buf):
            ops.gemm_silu_mul(wps[i], x, I, K, act)
        ops.gemm_argmax(lm, ops.rows_frag(x), V, K, 0, 16, ws, ids, 0)

    return train, n_buf


def measure(train, reps=30):
    for _ in range(3):
        train()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        train()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return min(ts), statistics.median(ts)


def main():
    order = sys.argv[1] if len(sys.argv) > 1 else "ABCDEF"
    train, _ = build()
    keep = {}
    env = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_SDMA",
                                          "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "ROC_ACTIVE_WAIT_TIMEOUT") if os.environ.get(k)}
    print(f"# env {env}  order {order}", flush=True)
    for st in order:
        if st == "B":
            keep["s2"] = torch.cuda.Stream()
        elif st == "C":
            if "s2" not in keep:
                keep["s2"] = torch.cuda.Stream()
            with torch.cuda.stream(keep["s2"]):
                keep["t"] = torch.zeros(64, device=dev) + 1
            keep["s2"].synchronize()
        elif st == "D":
            keep.pop("s2", None)
            torch.cuda.synchronize()
        elif st == "E":
            g = torch.cuda.CUDAGraph()
            y = torch.zeros(64, device=dev)
            with torch.cuda.graph(g):
                y += 1
            keep["g"] = g
        elif st == "F":
            keep["g"].replay()
            torch.cuda.synchronize()
        elif st == "G":   # drop the graph object again
            keep.pop("g", None)
            torch.cuda.synchronize()
        elif st == "P":   # a high-priority stream instead of a default-priority one
            keep["sp"] = torch.cuda.Stream(priority=-1)
            with torch.cuda.stream(keep["sp"]):
                keep["t2"] = torch.zeros(64, device=dev) + 1
            keep["sp"].synchronize()
        elif st == "N":   # a non-blocking memcpy on the null stream (SDMA queue?)
            h = torch.zeros(1 << 20, dtype=torch.uint8).pin_memory()
            keep["h2d"] = h.to(dev, non_blocking=True)
            torch.cuda.synchronize()
        best, med = measure(train)
        print(f"state {st}: train best {best:8.1f} us  median {med:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
