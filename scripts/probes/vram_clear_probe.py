"""What torch.cuda.empty_cache() costs the kernels that run AFTER it (the "+3-4 % after a graph capture").

torch.cuda.graph() calls torch.cuda.empty_cache() before it begins a capture.  With ~17 GB of dropped weights sitting in
torch's caching allocator (bench.py: the HF model behind NativeTarget(keep_hf=False)) that hands 17 GB back to the
driver, which clears released VRAM in the background.  This probe frees GB gigabytes the same way and times a fixed
train of weight-streaming launches (4 x gate/up + lm_head of an 8B layer, ~1.8 GB per train) every few milliseconds
before and after the empty_cache(): the series shows how long and by how much the clear competes for HBM.
usage: python scripts/probes/vram_clear_probe.py [GB=17]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from stream_slowdown import build

dev = torch.device("cuda", 0)
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 17.0


def series(train, seconds, tag):
    out, t_end = [], time.perf_counter() + seconds
    t0 = time.perf_counter()
    while time.perf_counter() < t_end:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        train()
        e.record()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0, s.elapsed_time(e) * 1e3))
    best = min(v for _, v in out)
    print(f"{tag}: {len(out)} trains, best {best:.1f} us; per 100 ms window (mean us): " +
          " ".join(f"{sum(v for t, v in out if w <= t < w + 0.1) / max(1, sum(1 for t, v in out if w <= t < w + 0.1)):.0f}"
                   for w in [i / 10 for i in range(int(seconds * 10))]), flush=True)


def main():
    train, _ = build()
    for _ in range(5):
        train()
    torch.cuda.synchronize()
    blocks = [torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1) for _ in range(int(GB))]
    torch.cuda.synchronize()
    series(train, 0.5, f"before (holding {GB:.0f} GB)")
    del blocks
    torch.cuda.synchronize()
    series(train, 0.5, "freed into torch's cache ")
    t0 = time.perf_counter()
    torch.cuda.empty_cache()
    print(f"empty_cache() returned after {1e3 * (time.perf_counter() - t0):.1f} ms (host)", flush=True)
    series(train, 2.0, "after empty_cache()       ")


if __name__ == "__main__":
    main()
