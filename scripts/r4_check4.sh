#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$PWD}"
python -m pytest tests/test_hip_batch.py tests/test_hip_moe.py -x -q -m gpu 2>&1 | tail -4
DFL_QKV_PARTS=0 python -m pytest tests/test_hip_batch.py -x -q -m gpu 2>&1 | tail -2
for v in 0 2 4; do echo "== DFL_RING_VARIANT=$v"; DFL_RING_VARIANT=$v SRC=frag python scripts/bench_gemm_batch.py gateup 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
for rep in 1 2; do for q in 0 1; do for v in 0 2; do DFL_RING_VARIANT=$v DFL_QKV_PARTS=$q python bench.py --no-cpu-baseline --requests-per-gpu 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch4 qkv_parts=$q ring_variant=$v', round(d['ms_per_step'],4), d['lossless_fraction'])"; done; done; done
for v in 0 1; do echo "== DFL_RING_N1=$v"; DFL_RING_N1=$v SRC=frag python scripts/bench_gemm_batch.py o down or 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
DFL_RING_N1=1 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do DFL_RING_N1=$v python bench.py --no-cpu-baseline --no-batch4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N=1 ring_n1=$v', round(d['ms_per_step'],4), d['lossless_fraction'])"; done; done
