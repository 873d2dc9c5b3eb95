set -e
cd /root/repo
for p in 1024 1024 4096 8192; do
timeout -k 10 400 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --prefix $p > gpurun_out/b_p$p.log 2>&1 || { tail -30 gpurun_out/b_p$p.log; exit 1; }
echo "prefix $p $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_p$p.log) $(grep -o '"lossless[^,]*' gpurun_out/b_p$p.log | head -1)"
done
