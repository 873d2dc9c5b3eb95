#!/bin/bash
# bench.py in the process states blamed for the "+4 % on every kernel" (profiles/r3_graph_ab.txt), one box, interleaved.
#   usage: stream_ab_bench.sh "<state> <state> ..." [reps] [prof]     states: eager used graph1 graph1_del emptycache hipgraph graph
cd "${GRAFT_REPO_ROOT:-$PWD}"
R=$PWD
STATES=${1:-"eager used graph1 graph"}; REPS=${2:-2}; PROF=${3:-}
one() { tag=$1; st=$2; extra=""; [ "$st" = graph ] && extra="--graph"
  env DFL_BENCH_EXTRA_STREAM=$([ "$st" = eager -o "$st" = graph ] && echo "" || echo $st) python3 bench.py --no-cpu-baseline $extra 2>gpurun_out/sab_$tag.err | tail -1 > gpurun_out/sab_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/sab_{sys.argv[1]}.json")); h = d["host_side"]; r = d["roofline"]
print(f"{sys.argv[1]:14s} {d['ms_per_step']:.4f} ms/cycle  gate/up ev {1e3*r['avg_ms']:.2f} us  lm_head ev {1e3*r['also'][0]['avg_ms']:.1f} us  "
      f"enqueue {h['enqueue_ms_per_cycle']:.3f}  wait {h['poll_wait_ms_per_cycle']:.3f}  lossless {d['lossless_fraction']}", flush=True)
PY
}
for rep in $(seq $REPS); do for st in $STATES; do one ${st}_$rep $st; done; done
if [ -n "$PROF" ]; then   # kernel trace of two states: which kernels get longer, and do the gaps change?
  cd /tmp && export TMPDIR=/tmp
  for st in $PROF; do
    rm -rf "$R/gpurun_out/sabprof_$st"
    DFL_BENCH_EXTRA_STREAM=$([ "$st" = eager ] && echo "" || echo $st) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/sabprof_$st" -- \
      python3 "$R/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/sabprof_$st.json" 2> "$R/gpurun_out/sabprof_$st.err"
    echo "== profiled $st rc=$?"
    python3 "$R/scripts/kstats_trace.py" "$R/gpurun_out/sabprof_$st" 3000 | tee "$R/gpurun_out/sabprof_$st.txt" | head -12
    python3 - "$R/gpurun_out/sabprof_$st" <<'PY'
import csv, glob, os, sys, collections
p = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
q = collections.Counter((r.get("Queue_Id"), r.get("Stream_Id")) for r in csv.DictReader(open(p)))
print("queues/streams:", dict(q))
PY
    rm -rf "$R/gpurun_out/sabprof_$st"
  done
fi
