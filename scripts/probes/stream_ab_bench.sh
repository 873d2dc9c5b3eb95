#!/bin/bash
# bench.py in the process states blamed for the "+4 % on every kernel" (profiles/r3_graph_ab.txt), one box, interleaved.
cd "${GRAFT_REPO_ROOT:-$PWD}"
one() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline $EXTRA 2>gpurun_out/sab_$tag.err | tail -1 > gpurun_out/sab_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/sab_{sys.argv[1]}.json")); h = d["host_side"]; r = d["roofline"]
print(f"{sys.argv[1]:10s} {d['ms_per_step']:.4f} ms/cycle  gate/up ev {1e3*r['avg_ms']:.2f} us  lm_head ev {1e3*r['also'][0]['avg_ms']:.1f} us  "
      f"enqueue {h['enqueue_ms_per_cycle']:.3f}  wait {h['poll_wait_ms_per_cycle']:.3f}  lossless {d['lossless_fraction']}", flush=True)
PY
}
for rep in 1 2; do
  EXTRA="" one eager$rep DUMMY=1
  EXTRA="" one used$rep DFL_BENCH_EXTRA_STREAM=used
  EXTRA="" one graph1_$rep DFL_BENCH_EXTRA_STREAM=graph1
  EXTRA="--graph" one graph$rep DUMMY=1
done
