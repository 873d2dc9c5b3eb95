#!/bin/bash
# same-box interleaved A/B of two builds of the library: scripts/r4_lib_ab.sh <variant .so> "<bench args>" [reps]
cd "${GRAFT_REPO_ROOT:-$PWD}"
lib=$1; args=$2; reps=${3:-2}
one() { env $1 timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$2', round(d['ms_per_step'],4), 'ms/cycle lossless', d['lossless_fraction'])"; }
for i in $(seq $reps); do one A=1 "default lib" && one DFL_LIB_PATH=$PWD/$lib "variant $lib" || exit 1; done
