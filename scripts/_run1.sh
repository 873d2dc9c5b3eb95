set -e
cd /root/repo
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline > gpurun_out/b_n1.log 2>&1 || { tail -30 gpurun_out/b_n1.log; exit 1; }
echo "N1 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_n1.log) $(grep -o '"lossless[^,]*' gpurun_out/b_n1.log | head -1) $(grep -o '"mean_acceptance_length": [0-9.]*' gpurun_out/b_n1.log)"
done
timeout -k 10 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --requests-per-gpu 4 > gpurun_out/b_b4.log 2>&1 || { tail -30 gpurun_out/b_b4.log; exit 1; }
echo "B4 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_b4.log) $(grep -o '"lossless[^,]*' gpurun_out/b_b4.log | head -1) $(grep -o '"value": [0-9.]*' gpurun_out/b_b4.log | head -1)"
timeout -k 10 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --requests-per-gpu 4 --graph > gpurun_out/b_b4g.log 2>&1 || { tail -30 gpurun_out/b_b4g.log; exit 1; }
echo "B4graph $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_b4g.log) $(grep -o '"lossless[^,]*' gpurun_out/b_b4g.log | head -1) $(grep -o '"value": [0-9.]*' gpurun_out/b_b4g.log | head -1)"
