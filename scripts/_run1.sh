set -e
cd /root/repo
for m in 256 448 256 448; do
DFL_ATTN_HEAD_WGS_MULTI=$m timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --requests-per-gpu 4 > gpurun_out/b_b4.log 2>&1 || { tail -30 gpurun_out/b_b4.log; exit 1; }
echo "R=4 MULTI=$m $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_b4.log)"
done
for m in 256 448 256 448; do
DFL_ATTN_HEAD_WGS_MULTI=$m timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --requests-per-gpu 2 > gpurun_out/b_b2.log 2>&1 || { tail -30 gpurun_out/b_b2.log; exit 1; }
echo "R=2 MULTI=$m $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_b2.log)"
done
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/b_n1.log 2>&1; echo "N1 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b_n1.log)"
