#!/bin/bash
# round-4: norm + gate Linear + routing of an MoE layer as one launch (dfl_moe_router; DFL_MOE_ROUTER=split: three launches) —
# tests, a kernel trace of both arms, then a same-box interleaved A/B of BASELINE configs[4]'s line
cd "${GRAFT_REPO_ROOT:-$PWD}"
R=$PWD; O=$R/gpurun_out/r4router; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_moe.py tests/test_hip_candidates.py -x -q -m gpu 2>&1 | tail -4 || exit 1
cd /tmp && export TMPDIR=/tmp
for v in fused split; do
DFL_MOE_ROUTER=$v timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -o t -- python3 $R/bench.py --workload qwen3-30b-a3b --schedule 8,12,16 --no-cpu-baseline --steps 24 --warmup 4 > $O/trace_$v.json 2> $O/trace_$v.err || exit 1
python3 $R/scripts/kstats_trace.py $O/trace_$v > $O/kernels_$v.txt 2>&1; echo "== $v"; grep -v "k_p\|pack_weight" $O/kernels_$v.txt | head -9
done
cd $R
one() { DFL_MOE_ROUTER=$1 timeout -k 10 500 python bench.py --workload qwen3-30b-a3b --schedule 8,12,16 --no-cpu-baseline 2>$O/err_$1.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('router=$1', round(d['ms_per_step'],4), 'ms/cycle', round(d['value'],1), 'tok/s lossless', d['lossless_fraction'], 'verify', round(d['hot_path']['target_verify_ms_per_cycle'],3))"; }
one split && one fused && one split && one fused
