python -m pytest tests -m gpu -q > gpurun_out/r3_final_gpu_tests.log 2>&1; tail -1 gpurun_out/r3_final_gpu_tests.log
python bench.py > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err
python bench.py --requests-per-gpu 4 --no-cpu-baseline > gpurun_out/r3_bench_batch4.json 2> gpurun_out/r3_bench_batch4.err
python bench.py --requests-per-gpu 4 --no-cpu-baseline --graph > gpurun_out/r3_bench_batch4_graph.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/r3_bench_n1.json","gpurun_out/r3_bench_batch4.json","gpurun_out/r3_bench_batch4_graph.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4), d["lossless_fraction"])
PY
