#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2; do for v in 0 1; do DFL_GEMM_NODYN=$v python bench.py --no-cpu-baseline --no-batch4 --eager 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N=1 eager nodyn=$v', round(d['ms_per_step'],4), d['lossless_fraction'])"; done; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
