#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
b4() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --requests-per-gpu 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch4 $tag', round(d['ms_per_step'],4), d['lossless_fraction'])"; }
DFL_RING_OPROJ=1 python -m pytest tests/test_hip_batch.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1; do echo "== DFL_RING_OPROJ=$v"; DFL_RING_OPROJ=$v SRC=frag python scripts/bench_gemm_batch.py of 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
for rep in 1 2 3; do
  b4 "default" DUMMY=1
  b4 "oproj_ring4" DFL_RING_OPROJ=1
done
