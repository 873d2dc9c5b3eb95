#!/bin/bash
# Round-4 profile set, run on the GPU box (gpurun): rocprofv3 kernel stats of the bench command per workload, FETCH_SIZE /
# WRITE_SIZE PMC passes (separate runs, kernel trace only), summaries into gpurun_out/r4prof/.  Profiled runs launch
# eagerly (--eager): same kernels as the replayed cycle, one dispatch record per launch.
#   profile_r4.sh stats1 | stats4 | statsl | statsm | pmc1 | pmc4 | pmcm | lines | linem
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O="$R/gpurun_out/r4prof"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_stats() {  # tag, bench args...
  tag=$1; shift
  rm -rf "$O/stats_$tag"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$tag" -- \
    python3 "$R/bench.py" --no-cpu-baseline --eager "$@" > "$O/bench_${tag}_profiled.json" 2> "$O/stats_$tag.err" || { tail -5 "$O/stats_$tag.err"; return 1; }
  python3 "$R/scripts/kstats.py" "$O/stats_$tag" > "$O/bench_${tag}_kernels_short.txt"
  python3 "$R/scripts/kstats_trace.py" "$O/stats_$tag" 3000 > "$O/bench_${tag}_kernels_by_grid.txt"
  cp "$(ls "$O/stats_$tag"/*/*kernel_stats.csv | head -1)" "$O/bench_${tag}_kernel_stats.csv"
  rm -rf "$O/stats_$tag"
  head -10 "$O/bench_${tag}_kernels_by_grid.txt"
}
run_pmc() {  # tag (suffix for pmc_summary), bench args...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$O/pmc_${c}$tag"
    timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_${c}$tag" -- \
      python3 "$R/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --eager "$@" > /dev/null 2> "$O/pmc_${c}$tag.err" || { tail -5 "$O/pmc_${c}$tag.err"; return 1; }
  done
  python3 "$R/scripts/pmc_summary.py" "$O" "$tag" > "$O/pmc${tag}_summary.csv"
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf "$O/pmc_${c}$tag"; done
  cat "$O/pmc${tag}_summary.csv"
}
case "${1:-all}" in
  stats1) run_stats n1 --steps 24 --warmup 2 --no-batch4 ;;
  stats4) run_stats batch4 --steps 24 --warmup 2 --requests-per-gpu 4 ;;
  statsl) run_stats llama31_8b --steps 24 --warmup 2 --workload llama31-8b ;;
  statsm) run_stats qwen3_30b_a3b --steps 24 --warmup 4 --workload qwen3-30b-a3b ;;
  pmc1) run_pmc "" --target-layers 8 --no-batch4 ;;
  pmc4) run_pmc "_b4" --target-layers 8 --requests-per-gpu 4 ;;
  pmcm) run_pmc "_moe" --target-layers 6 --workload qwen3-30b-a3b ;;
  attn)    # the attention stage's split knobs in the N = 1 cycle: one old-key split per head = 64 workgroups
    cd "$R"
    for rep in 1 2; do for w in 224 64 96; do
      DFL_ATTN_HEAD_WGS=$w python3 bench.py --no-cpu-baseline --no-batch4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N=1 attn_head_wgs=$w', round(d['ms_per_step'],4), d['lossless_fraction'])"
    done; done ;;
  linem)   # BASELINE configs[4]'s line alone
    cd "$R"
    python3 bench.py --no-cpu-baseline --workload qwen3-30b-a3b > "$O/r4_bench_qwen3_30b_a3b.json" 2> "$O/r4_bench_moe.err"
    python3 -c "import json; d=json.loads(open('$O/r4_bench_qwen3_30b_a3b.json').read().strip().splitlines()[-1]); print(round(d['value'],1),'tok/s',round(d['ms_per_step'],4),'ms',d['lossless_fraction'],'tau',round(d['mean_acceptance_length'],2),'roofline frac',round(d['roofline']['frac'],3), d['used_block_sizes'])" ;;
  lines)   # the bench lines kept under profiles/ (un-profiled, replay where the workload allows it)
    cd "$R"
    python3 bench.py > "$O/r4_bench_n1.json" 2> "$O/r4_bench_n1.err"; tail -c 300 "$O/r4_bench_n1.err"
    python3 bench.py --no-cpu-baseline --workload llama31-8b > "$O/r4_bench_llama31_8b.json" 2> "$O/r4_bench_llama.err"
    python3 bench.py --no-cpu-baseline --workload qwen3-30b-a3b > "$O/r4_bench_qwen3_30b_a3b.json" 2> "$O/r4_bench_moe.err"
    python3 bench.py --no-cpu-baseline --steps 200 --warmup 8 > "$O/r4_bench_n1_200cycles.json" 2>/dev/null
    python3 - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r4prof")
for f in ("r4_bench_n1.json", "r4_bench_llama31_8b.json", "r4_bench_qwen3_30b_a3b.json", "r4_bench_n1_200cycles.json"):
    try:
        d = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
        b = d.get("batch4") or {}
        print(f, round(d["value"], 1), "tok/s", round(d["ms_per_step"], 4), "ms", d["lossless_fraction"], "roofline frac", round(d["roofline"]["frac"], 3),
              "| batch4", round(b.get("value", 0), 1), round(b.get("ms_per_step", 0), 4), b.get("lossless_fraction"))
    except Exception as e:
        print(f, "FAILED", e)
PY
    ;;
esac
