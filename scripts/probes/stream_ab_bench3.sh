#!/bin/bash
# Is the "+3-4 % after a torch graph capture" the driver clearing the VRAM that torch.cuda.empty_cache() hands back
# (torch.cuda.graph() calls it; ~16 GB of dropped HF weights sit in torch's cache)?  One box, interleaved.
cd "${GRAFT_REPO_ROOT:-$PWD}"
one() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline $BARGS 2>gpurun_out/sab3_$tag.err | tail -1 > gpurun_out/sab3_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/sab3_{sys.argv[1]}.json")); h = d.get("host_side") or {}; r = d["roofline"]
lm = r["also"][0]["avg_ms"] if "also" in r else r["avg_ms"]
print(f"{sys.argv[1]:22s} {d['ms_per_step']:.4f} ms/cycle  lm_head ev {1e3*lm:.1f} us  enqueue {h.get('enqueue_ms_per_cycle', 0):.3f}  wait {h.get('poll_wait_ms_per_cycle', 0):.3f}  lossless {d['lossless_fraction']}", flush=True)
PY
}
for rep in 1 2; do
  BARGS="" one eager_$rep DUMMY=1
  BARGS="" one empty_$rep DFL_BENCH_EXTRA_STREAM=emptycache
  BARGS="" one empty_sleep3_$rep DFL_BENCH_EXTRA_STREAM=emptycache DFL_BENCH_EXTRA_SLEEP=3
  BARGS="--graph" one graph_$rep DUMMY=1
  BARGS="--graph" one graph_early_$rep DFL_BENCH_EARLY_EMPTY=1 DFL_BENCH_EXTRA_SLEEP=3
  BARGS="--hf-prefill" one keephf_empty_$rep DFL_BENCH_EXTRA_STREAM=emptycache
done
BARGS="--steps 400" one eager_400 DUMMY=1
BARGS="--steps 400" one empty_400 DFL_BENCH_EXTRA_STREAM=emptycache
BARGS="--steps 400 --graph" one graph_early_400 DFL_BENCH_EARLY_EMPTY=1 DFL_BENCH_EXTRA_SLEEP=3
BARGS="--requests-per-gpu 4" one b4_eager DUMMY=1
BARGS="--requests-per-gpu 4 --graph" one b4_graph_early DFL_BENCH_EARLY_EMPTY=1 DFL_BENCH_EXTRA_SLEEP=3
