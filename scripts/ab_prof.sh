#!/bin/bash
# In-situ per-kernel A/B on one box: rocprofv3 kernel trace of the bench command with each library.
#   scripts/ab_prof.sh <tag>=<lib.so> [<tag>=<lib.so> ...]      -> gpurun_out/abprof_<tag>.txt
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%=*}; lib=${spec#*=}
  rm -rf "$R/gpurun_out/abprof_$tag"
  DFL_LIB_PATH="$R/$lib" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/abprof_$tag" -- \
    python3 "$R/bench.py" --steps 24 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > "$R/gpurun_out/abprof_$tag.json" 2> "$R/gpurun_out/abprof_$tag.err"
  echo "== $tag rc=$?"
  python3 "$R/scripts/kstats_trace.py" "$R/gpurun_out/abprof_$tag" 3000 | tee "$R/gpurun_out/abprof_$tag.txt" | head -14
  rm -rf "$R/gpurun_out/abprof_$tag"
done
