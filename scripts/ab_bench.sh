#!/bin/bash
# Same-box A/B of two builds of libdflash_hip.so through bench.py (boxes of the pool differ by a few per cent, so a
# comparison across gpurun calls says little):  scripts/ab_bench.sh <libA.so> <libB.so> [rounds] [bench args...]
# Interleaved rounds; prints ms per cycle, draft+lm_head ms, verify ms per run.
A=$1; B=$2; N=${3:-2}; shift 3 || true
R=${GRAFT_REPO_ROOT:-$PWD}
for i in $(seq $N); do
  for L in "$A" "$B"; do
    DFL_LIB_PATH=$L python3 "$R/bench.py" --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); h=d['hot_path']; r=d['roofline']
print('$L'.split('/')[-2], 'ms/cycle %.3f  draft+lm %.3f  verify %.3f  lossless %.2f  roofline kernel %.1f us' % (d['ms_per_step'], h['draft_plus_lm_head_ms_per_cycle'], h['target_verify_ms_per_cycle'], d['lossless_fraction'], 1e3*r['avg_ms']))"
  done
done
