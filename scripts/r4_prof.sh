#!/bin/bash
# round-4: kernel trace of the 4-request leg (ring GEMMs) and an A/B of the N = 1 modes, one box
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4prof; mkdir -p $O
one() { tag=$1; shift; python3 bench.py --no-cpu-baseline --no-batch4 "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python3 -c "
import json; d=json.load(open('$O/$tag.json')); h=d['host_side']
print('$tag', round(d['ms_per_step'],4), 'ms/cycle  enqueue', round(h['enqueue_ms_per_cycle'],3), 'wait', round(h['poll_wait_ms_per_cycle'],3), 'lossless', d['lossless_fraction'])"; }
for rep in 1 2; do
  one eager_$rep --eager
  one graph_e4_$rep
  one graph_e48_$rep --event-every 48
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_b4
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_b4 -- python3 $R/bench.py --steps 24 --warmup 2 --no-cpu-baseline --requests-per-gpu 4 --eager > $O/bench_b4_profiled.json 2> $O/trace_b4.err
python3 $R/scripts/kstats_trace.py $O/trace_b4 3000 | tee $O/b4_kernels_by_grid.txt | head -16
rm -rf $O/trace_b4
