set -e
cd /root/repo
for t in 8 16 12 6; do
DFL_ATTN_HEAD_TILES=$t timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/b_n1.log 2>&1 || { tail -30 gpurun_out/b_n1.log; exit 1; }
echo "TILES=$t $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_n1.log)"
done
DFL_ATTN_HEAD_WGS=288 timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/b_n1.log 2>&1; echo "WGS=288 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_n1.log)"
