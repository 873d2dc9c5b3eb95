#!/bin/bash
# Round-3 profile set, run on the GPU box (gpurun): kernel stats of the bench command (N = 1 and 4 requests per GPU),
# FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel trace only), summaries into gpurun_out/r3prof/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O="$R/gpurun_out/r3prof"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_stats() {  # tag, bench args...
  tag=$1; shift
  rm -rf "$O/stats_$tag"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$tag" -- \
    python3 "$R/bench.py" --no-cpu-baseline "$@" > "$O/bench_${tag}_profiled.json" 2> "$O/stats_$tag.err" || return 1
  python3 "$R/scripts/kstats.py" "$O/stats_$tag" > "$O/bench_${tag}_kernels_short.txt"
  python3 "$R/scripts/kstats_trace.py" "$O/stats_$tag" 3000 > "$O/bench_${tag}_kernels_by_grid.txt"
  cp "$(ls "$O/stats_$tag"/*/*kernel_stats.csv | head -1)" "$O/bench_${tag}_kernel_stats.csv"
  rm -rf "$O/stats_$tag"
  head -8 "$O/bench_${tag}_kernels_short.txt"
}
run_pmc() {  # tag, suffix for pmc_summary, bench args...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$O/pmc_${c}$tag"
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_${c}$tag" -- \
      python3 "$R/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --target-layers 8 "$@" > /dev/null 2> "$O/pmc_${c}$tag.err" || return 1
  done
  python3 "$R/scripts/pmc_summary.py" "$O" "$tag" > "$O/pmc${tag}_summary.csv"
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf "$O/pmc_${c}$tag"; done
  cat "$O/pmc${tag}_summary.csv"
}
case "${1:-all}" in
  stats1) run_stats n1 --steps 24 --warmup 2 ;;
  stats4) run_stats batch4 --steps 24 --warmup 2 --requests-per-gpu 4 ;;
  pmc1) run_pmc "" ;;
  pmc4) run_pmc "_b4" --requests-per-gpu 4 ;;
  all) run_stats n1 --steps 24 --warmup 2 && run_stats batch4 --steps 24 --warmup 2 --requests-per-gpu 4 && run_pmc "" && run_pmc "_b4" --requests-per-gpu 4 ;;
esac
