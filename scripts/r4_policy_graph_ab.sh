#!/bin/bash
# BASELINE configs[4] (EWMA schedule over 8,12,16): the policy loop by graph replay (one pair of graphs per size) against eager launches
cd "${GRAFT_REPO_ROOT:-$PWD}"
one() { timeout -k 10 500 python bench.py --workload qwen3-30b-a3b --no-cpu-baseline $1 2>gpurun_out/pg_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('policy $1', round(d['ms_per_step'],4), 'ms/cycle', round(d['value'],1), 'tok/s lossless', d['lossless_fraction'], 'tau', round(d['mean_acceptance_length'],2), d['used_block_sizes'], 'enqueue', round(d['host_side']['enqueue_ms_per_cycle'],3), d['host_side']['mode'][:24])" || tail -5 gpurun_out/pg_err.txt; }
one --eager && one "" && one --eager && one ""
