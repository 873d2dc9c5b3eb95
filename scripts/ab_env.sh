#!/bin/bash
# A/B of an environment switch on one box, interleaved: scripts/ab_env.sh VAR "<bench args>" [reps=2]
# prints ms_per_step and the lossless fraction of `python bench.py --no-cpu-baseline <bench args>` with VAR=0 and VAR=1.
var=$1; args=$2; reps=${3:-2}
for i in $(seq $reps); do
  for v in 0 1; do
    env $var=$v python bench.py --no-cpu-baseline $args 2>gpurun_out/ab_env.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'], 4), d['lossless_fraction'])" || tail -3 gpurun_out/ab_env.err
  done
done
