set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
for v in 1 0 1 0; do
DFL_HOST_RESULT=$v timeout -k 10 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline > gpurun_out/b_ab.log 2>&1 || { tail -30 gpurun_out/b_ab.log; exit 1; }
echo "HOST_RESULT=$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_ab.log) $(grep -o '"lossless[^,]*' gpurun_out/b_ab.log | head -1)"
done
