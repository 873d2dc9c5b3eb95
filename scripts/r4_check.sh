#!/bin/bash
# round-4 GPU check: ring-kernel tests, rotation sweep of the micro-benchmark, the default bench line (headline + batch4)
cd "${GRAFT_REPO_ROOT:-$PWD}"
O=gpurun_out/r4check; mkdir -p $O
python -m pytest tests/test_hip_batch.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -4
for m in 0 1 3 5; do echo "== ring rot=$m"; DFL_RING_ROT=$m SRC=frag python scripts/bench_gemm_batch.py gateup lm_head 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
echo "== slab"; DFL_BATCH_GEMM=slab SRC=frag python scripts/bench_gemm_batch.py gateup lm_head 2>&1 | grep -v -i "warn\|amdgpu.ids"
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4check/bench_default.json"))
b = d["batch4"]
print(f"N=1 {d['ms_per_step']:.4f} ms/cycle {d['value']:.0f} tok/s lossless {d['lossless_fraction']} mode {d['host_side']['mode'][:20]} enqueue {d['host_side']['enqueue_ms_per_cycle']:.3f}")
print(f"batch4 {b['ms_per_step']:.4f} ms/cycle {b['value']:.0f} tok/s lossless {b['lossless_fraction']} lm_head {1e3*b['roofline']['avg_ms']:.1f} us frac cycle {b['cycle_frac_of_8TBps']:.3f}")
PY
DFL_BATCH_GEMM=slab python bench.py --no-cpu-baseline --requests-per-gpu 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch4 slab headline', round(d['ms_per_step'],4), d['lossless_fraction'])"
python bench.py --no-cpu-baseline --requests-per-gpu 4 --eager 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch4 ring eager', round(d['ms_per_step'],4), d['lossless_fraction'])"
