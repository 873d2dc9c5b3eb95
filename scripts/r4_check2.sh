#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$PWD}"
python -m pytest tests/test_hip_batch.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -4
DFL_RING_RESID=1 python -m pytest tests/test_hip_batch.py -x -q -m gpu 2>&1 | tail -3
for v in "DFL_RING_VARIANT=0" "DFL_RING_VARIANT=1"; do echo "== $v"; env $v SRC=frag python scripts/bench_gemm_batch.py gateup lm_head 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
for v in 0 1 2; do echo "== DFL_RING_RESID=$v"; DFL_RING_RESID=$v SRC=frag python scripts/bench_gemm_batch.py qkvr or qkv 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
for v in 0 1 2; do DFL_RING_RESID=$v python bench.py --no-cpu-baseline --requests-per-gpu 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch4 resid-ring=$v', round(d['ms_per_step'],4), d['lossless_fraction'])"; done
